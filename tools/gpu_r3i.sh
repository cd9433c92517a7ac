#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3i
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
Q="--steps 5 --warmup 2 --no-cpu-baseline --no-gather --no-pcie-f64 --no-block"
run() { tag=$1; shift; env "$@" timeout 300 python bench.py $Q > $OUT/b_$tag.json 2> $OUT/b_$tag.err; python - <<PY
import json
try:
    d=json.load(open("$OUT/b_$tag.json"))
    print("$tag", d["job_ms"], d["value"], d["mask_parity_vs_reference_golden"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"], [(c["layers"][0][:3], c["ms"]) for c in d["chunks_rank0_last_job"][:5]])
except Exception as e:
    print("$tag ERR", e)
PY
}
run full2 CP_JOB_LATENCY_KIND=full CP_JOB_PRECOMPUTE=2
run full3 CP_JOB_LATENCY_KIND=full CP_JOB_PRECOMPUTE=3
run full5 CP_JOB_LATENCY_KIND=full CP_JOB_PRECOMPUTE=5
run gram3 CP_JOB_PRECOMPUTE=3
# two ranks on this one GPU through gloo: the N > 1 flow of bench.py (weak and strong), self-launched
CP_BENCH_DIST_BACKEND=gloo timeout 400 python bench.py --gpus 2 --steps 3 --warmup 1 --no-gather > $OUT/b_2ranks_weak.json 2> $OUT/b_2ranks_weak.err
echo "2ranks weak rc=$?"
CP_BENCH_DIST_BACKEND=gloo timeout 400 python bench.py --gpus 2 --scaling strong --steps 3 --warmup 1 --no-gather > $OUT/b_2ranks_strong.json 2> $OUT/b_2ranks_strong.err
echo "2ranks strong rc=$?"
for f in 2ranks_weak 2ranks_strong; do python - <<PY
import json
try:
    d=json.load(open("$OUT/b_$f.json"))
    print("$f", d["n_gpus"], d["scaling"], d["job_ms"], d["value"], d["mask_parity_vs_reference_golden"], d.get("masks_identical_on_every_rank"), d["exchange_rank0"], d["config"]["parallelism"])
except Exception as e:
    print("$f ERR", e); print(open("$OUT/b_$f.err").read()[-1500:])
PY
done
