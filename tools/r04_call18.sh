#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_call18
mkdir -p $OUT
cd $R
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sign or itq or vh" < /dev/null > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest.log
echo "--- itq profile (sign route)"; timeout -k 5 300 python tests/tools/itq_profile.py 2>&1 | tail -5
echo "--- itq profile (CP_ITQ_SIGN=0)"; CP_ITQ_SIGN=0 timeout -k 5 300 python tests/tools/itq_profile.py 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
timeout -k 5 600 python $R/bench.py --workload r3 --no-cpu-baseline --steps 2 --warmup 1 > $OUT/bench_r3.json 2> $OUT/bench_r3.err; echo "r3 rc=$?"
python - $OUT/bench_r3.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("r3 job_ms", d["job_ms"], d["stage_ms_per_job"])
print({k: v["itq_ms"] for k, v in d["per_conv"].items()})
PY

