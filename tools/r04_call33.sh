#!/bin/bash
# how many layers' full normal equations fit under the searches, with the GEMM of this round
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${CALL_NAME:-r04_call33}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-gather --no-block --no-pcie-f64 --no-pipelined --profile-mode --steps 3 --warmup 2 --jobs-per-step 12"
for n in 2 0 1 3 4 5 2; do
  timeout -k 5 120 python $R/bench.py $Q --precompute-heaviest $n > $OUT/job_pre$n.json 2> $OUT/job_pre$n.err
  python - $OUT/job_pre$n.json $n <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("precompute_heaviest %s  job_ms %8.3f  layers/s %8.1f  parity %s" % (sys.argv[2], d.get("job_ms", -1), d["value"], d.get("mask_parity_vs_reference_golden")))
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
done
