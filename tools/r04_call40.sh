#!/bin/bash
# the multi-CU coordinate-descent team for the 512-channel layers too (its rows live in LDS / registers of two CUs):
# does the search then tolerate more normal equations computed under it?
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${CALL_NAME:-r04_call40}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-gather --no-block --no-pcie-f64 --no-pipelined --profile-mode --steps 3 --warmup 2 --jobs-per-step 12"
for cfg in "513 2" "512 2" "512 3" "512 5" "513 2"; do
  set -- $cfg
  CP_CD_MULTI_MIN_C=$1 timeout -k 5 120 python $R/bench.py $Q --precompute-heaviest $2 > $OUT/job_$1_$2.json 2> $OUT/job_$1_$2.err
  python - $OUT/job_$1_$2.json "$cfg" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("min_c / precompute %s  job_ms %8.3f  layers/s %8.1f  parity %s" % (sys.argv[2], d.get("job_ms", -1), d["value"], d.get("mask_parity_vs_reference_golden")))
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
done
