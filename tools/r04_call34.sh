#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${CALL_NAME:-r04_call34}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-gather --no-block --no-pcie-f64 --no-pipelined --profile-mode --steps 3 --warmup 2 --jobs-per-step 12"
for v in 0 1 0 1; do
  CP_WAIT_SPIN=$v timeout -k 5 120 python $R/bench.py $Q > $OUT/job_spin$v.json 2> $OUT/job_spin$v.err
  python - $OUT/job_spin$v.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("CP_WAIT_SPIN %s  job_ms %8.3f  layers/s %8.1f  parity %s" % (sys.argv[2], d.get("job_ms", -1), d["value"], d.get("mask_parity_vs_reference_golden")))
PY
done
