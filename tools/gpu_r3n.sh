#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3n
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
Q="--steps 5 --warmup 2 --no-cpu-baseline --no-gather --no-pcie-f64 --no-block"
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --workload resnet50 $Q > $OUT/b_$tag.json 2> $OUT/b_$tag.err; python - <<PY
import json
try:
    d=json.load(open("$OUT/b_$tag.json"))
    print("$tag", d["job_ms"], d["value"], d["mask_parity_vs_reference_golden"], [(c["layers"][0][:3], len(c["layers"]), c["ms"]) for c in d["chunks_rank0_last_job"][:6]])
except Exception as e:
    print("$tag ERR", e)
PY
}
run base X=1
run w2048_1 CP_BENCH_PER_STREAM_BY_WIDTH=2048:1
run w2048_1_1024_1 CP_BENCH_PER_STREAM_BY_WIDTH=2048:1,1024:1
run all3 CP_BENCH_PER_STREAM_BY_WIDTH=2048:1,1024:1,512:3,256:3,128:3,64:3
run spread CP_CD_SPREAD=1 CP_CD_EXCLUSIVE=0
