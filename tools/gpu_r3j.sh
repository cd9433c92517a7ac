#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3j
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 300 python tools/cd_bench.py > $OUT/cd_bench.log 2>&1
cat $OUT/cd_bench.log
Q="--steps 5 --warmup 2 --no-cpu-baseline --no-gather --no-pcie-f64 --no-block"
for w in vgg16 resnet50 vgg16_5x; do
timeout 300 python bench.py --workload $w $Q > $OUT/b_$w.json 2> $OUT/b_$w.err
python - <<PY
import json
try:
    d=json.load(open("$OUT/b_$w.json"))
    print("$w", d["job_ms"], d["value"], d["mask_parity_vs_reference_golden"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
    pl=d["per_layer_rank0"]
    for k in list(pl)[:3]: print("  ",k,pl[k])
except Exception as e:
    print("$w ERR", e)
PY
done
