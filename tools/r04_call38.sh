#!/bin/bash
# column means computed under the search: golden / refit tests, then the vgg16 line twice
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${CALL_NAME:-r04_call38}
mkdir -p $OUT
cd $R
timeout -k 5 400 python -m pytest tests/test_gpu_parity.py -k "refit or golden or run_to_run or concurrent or resident" -x -q > $OUT/pytest_sel.log 2>&1; echo "tests rc $?"; tail -4 $OUT/pytest_sel.log
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-gather --no-block --no-pcie-f64 --no-pipelined --profile-mode --steps 3 --warmup 2 --jobs-per-step 12"
for v in 1 2; do
  timeout -k 5 120 python $R/bench.py $Q > $OUT/job_$v.json 2> $OUT/job_$v.err
  python - $OUT/job_$v.json $v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("run %s  job_ms %8.3f  layers/s %8.1f  parity %s" % (sys.argv[2], d.get("job_ms", -1), d["value"], d.get("mask_parity_vs_reference_golden")), [c['ms'] for c in d['chunks_rank0_last_job']][:5])
PY
done
