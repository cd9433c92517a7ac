#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3h
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
Q="--steps 5 --warmup 2 --no-cpu-baseline --no-gather --no-pcie-f64 --no-block"
for v in "512:5" "512:3" "512:5,256:3" "512:2,256:3" "256:3" "512:5,256:3,128:2,64:2"; do
  tag=$(echo $v | tr ':,' '__')
  CP_BENCH_PER_STREAM_BY_WIDTH=$v timeout 300 python bench.py $Q > $OUT/b_$tag.json 2> $OUT/b_$tag.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/b_$tag.json"))
    print("$v", d["job_ms"], d["value"], d["mask_parity_vs_reference_golden"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], [(c["layers"][0][:3], len(c["layers"]), c["ms"]) for c in d["chunks_rank0_last_job"]])
except Exception as e:
    print("$v ERR", e)
PY
done
