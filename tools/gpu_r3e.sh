#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3e
R=$GRAFT_REPO_ROOT
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
CD_BENCH_C=512,1024,2048 timeout 300 python tools/cd_bench.py > $OUT/cd_bench.log 2>&1
cat $OUT/cd_bench.log
Q="--steps 5 --warmup 2 --no-cpu-baseline --no-gather --no-pcie-f64 --no-block"
timeout 300 python bench.py --workload resnet50 $Q > $OUT/b_resnet50.json 2> $OUT/b_resnet50.err
CP_CD_EXCLUSIVE=0 timeout 300 python bench.py --workload resnet50 $Q > $OUT/b_resnet50_noexcl.json 2> $OUT/b_resnet50_noexcl.err
for f in resnet50 resnet50_noexcl; do python - <<PY
import json
try:
    d=json.load(open("$OUT/b_$f.json"))
    print("$f", d["job_ms"], d["value"], d["mask_parity_vs_reference_golden"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
    pl=d["per_layer_rank0"]
    for k in list(pl)[:4]: print("  ",k,pl[k])
except Exception as e:
    print("$f ERR", e)
PY
done
