#!/bin/bash
# Round-4 call 4 (GPU box): what stretches a Cholesky step inside the job -- timeline of one job; variants.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_call4
mkdir -p $OUT
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
job base CP_NOP=1
job panel_last CP_CHOL_PANEL_LAST=1
job v128 CP_LIB_PATH=$R/build_variants/v128/libcpmi355.so
job v128_panel_last CP_LIB_PATH=$R/build_variants/v128/libcpmi355.so CP_CHOL_PANEL_LAST=1
job base2 CP_NOP=1
rm -rf /tmp/kt
timeout -k 5 200 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py --profile-mode --steps 1 --warmup 1 --jobs-per-step 3 > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
DB=$(find /tmp/kt -name '*.db' | head -1)
if [ -n "$DB" ]; then
  python $R/tools/rocpd_timeline.py $DB --anchor=k_lasso_prep:12 --streams=2 > $OUT/timeline_last_job.md 2>&1
fi
head -60 $OUT/timeline_last_job.md
