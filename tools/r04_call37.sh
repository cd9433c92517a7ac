#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${CALL_NAME:-r04_call37}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-gather --no-block --no-pcie-f64 --no-pipelined --profile-mode --steps 3 --warmup 2 --jobs-per-step 12"
for v in 0 1500 3000 5000 0 3000; do
  CP_PRECOMPUTE_DELAY_US=$v timeout -k 5 120 python $R/bench.py $Q > $OUT/job_delay$v.json 2> $OUT/job_delay$v.err
  python - $OUT/job_delay$v.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("CP_PRECOMPUTE_DELAY_US %s  job_ms %8.3f  layers/s %8.1f  parity %s" % (sys.argv[2], d.get("job_ms", -1), d["value"], d.get("mask_parity_vs_reference_golden")))
PY
done
