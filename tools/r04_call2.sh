#!/bin/bash
# Round-4 call 2 (GPU box): the one-launch-per-step Cholesky (chol_step.hip) and the CD team fallback.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_call2
mkdir -p $OUT
cd $R
timeout -k 5 900 python -m pytest tests -m gpu -q -x < /dev/null > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-gather --no-block --no-pcie-f64"
job() {  # name, env...
  local name=$1; shift
  env "$@" timeout -k 5 200 python $R/bench.py $Q --profile-mode --steps 3 --warmup 2 --jobs-per-step 12 > $OUT/job_$name.json 2> $OUT/job_$name.err
  python - $OUT/job_$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-28s job_ms %8.3f  layers/s %8.1f  parity %s  gram_ms %s" % (sys.argv[2], d.get("job_ms", -1), d["value"], d.get("mask_parity_vs_reference_golden"), (d.get("roofline") or {}).get("avg_launch_ms")))
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
job steps CP_NOP=1
job old CP_CHOL_STEPS=0
job steps2 CP_NOP=1
job resnet_steps CP_BENCH_WORKLOAD=resnet50
job resnet_old CP_BENCH_WORKLOAD=resnet50 CP_CHOL_STEPS=0
job v5x_steps CP_BENCH_WORKLOAD=vgg16_5x
job v5x_old CP_BENCH_WORKLOAD=vgg16_5x CP_CHOL_STEPS=0
timeout -k 5 300 python $R/bench.py $Q --steps 10 --warmup 3 > $OUT/bench_steps.json 2> $OUT/bench_steps.err; echo "bench rc=$?"
CP_CHOL_STEPS=0 timeout -k 5 300 python $R/bench.py $Q --steps 10 --warmup 3 > $OUT/bench_old.json 2> $OUT/bench_old.err; echo "bench old rc=$?"
python - $OUT <<'PY'
import json, sys
for n in ("bench_steps", "bench_old"):
    try:
        d = json.loads(open(sys.argv[1] + "/" + n + ".json").read().strip().splitlines()[-1])
        print(n, "job_ms", d["job_ms"])
        for k, v in d["per_layer_rank0"].items():
            print("   ", k, v["ms_alone"], "search", v["alpha_search_ms"], "refit", v["refit_ms"])
        for k, v in d["stage_ms_alone_by_shape_rank0"].items():
            print("   ", k, {a: b for a, b in v.items() if "chol" in a or "solve" in a or "factor" in a or "subst" in a})
    except Exception as e:
        print(n, "unreadable", e)
PY
rm -rf /tmp/kt
timeout -k 5 200 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py --profile-mode --steps 2 --warmup 1 --jobs-per-step 4 > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
DB=$(find /tmp/kt -name '*.db' | head -1)
[ -n "$DB" ] && python $R/tools/rocpd_kernels.py $DB 10 > $OUT/kernels_vgg16.md 2>&1
head -20 $OUT/kernels_vgg16.md
