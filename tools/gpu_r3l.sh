#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3l
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
CD_BENCH_C=512,1024,2048 timeout 300 python tools/cd_bench.py > $OUT/cd_bench.log 2>&1
cat $OUT/cd_bench.log
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cd_ or golden or tie" > $OUT/pytest.log 2>&1
tail -3 $OUT/pytest.log
Q="--steps 5 --warmup 2 --no-cpu-baseline --no-gather --no-pcie-f64 --no-block"
timeout 300 python bench.py --workload resnet50 $Q > $OUT/b_resnet50.json 2> $OUT/b_resnet50.err
python - <<PY
import json
d=json.load(open("$OUT/b_resnet50.json"))
print("resnet50", d["job_ms"], d["value"], d["mask_parity_vs_reference_golden"])
pl=d["per_layer_rank0"]
for k in list(pl)[:8]: print("  ",k,pl[k])
PY
