#!/bin/bash
# round 3, fourth GPU pass: tie sentinels, prune_resnet vs the reference-driven golden, exclusive-CU default, stream priority
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3d
R=$GRAFT_REPO_ROOT
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -6 $OUT/pytest.log
Q="--steps 5 --warmup 2 --no-cpu-baseline --no-gather --no-pcie-f64 --no-block"
timeout 300 python bench.py $Q > $OUT/b_default.json 2> $OUT/b_default.err
CP_JOB_PRIORITY=1 timeout 300 python bench.py $Q > $OUT/b_prio.json 2> $OUT/b_prio.err
CP_JOB_PRECOMPUTE=0 timeout 300 python bench.py $Q > $OUT/b_pre0.json 2> $OUT/b_pre0.err
CP_JOB_PRECOMPUTE=5 timeout 300 python bench.py $Q > $OUT/b_pre5.json 2> $OUT/b_pre5.err
timeout 300 python bench.py --per-stream 2 $Q > $OUT/b_ps2.json 2> $OUT/b_ps2.err
timeout 300 python bench.py --workload resnet50 $Q > $OUT/b_resnet50.json 2> $OUT/b_resnet50.err
timeout 300 python bench.py --workload resnet50 --per-stream 1 $Q > $OUT/b_resnet50_ps1.json 2> $OUT/b_resnet50_ps1.err
timeout 300 python bench.py --workload vgg16_5x $Q > $OUT/b_vgg16_5x.json 2> $OUT/b_vgg16_5x.err
for f in default prio pre0 pre5 ps2 resnet50 resnet50_ps1 vgg16_5x; do python - <<PY
import json
try:
    d=json.load(open("$OUT/b_$f.json"))
    print("$f", d["job_ms"], d["value"], d["mask_parity_vs_reference_golden"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
except Exception as e:
    print("$f ERR", e)
PY
done
