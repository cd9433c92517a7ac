#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3k
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 300 python tools/cd_bench.py > $OUT/cd_bench_wide.log 2>&1
cat $OUT/cd_bench_wide.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cd_ or golden" > $OUT/pytest_wide.log 2>&1
tail -3 $OUT/pytest_wide.log
Q="--steps 5 --warmup 2 --no-cpu-baseline --no-gather --no-pcie-f64 --no-block"
timeout 300 python bench.py $Q > $OUT/b_vgg16_wide.json 2> $OUT/b_vgg16_wide.err
python - <<PY
import json
d=json.load(open("$OUT/b_vgg16_wide.json"))
print("vgg16 wide", d["job_ms"], d["value"], d["mask_parity_vs_reference_golden"])
pl=d["per_layer_rank0"]
for k in list(pl)[:7]: print("  ",k,pl[k])
PY
