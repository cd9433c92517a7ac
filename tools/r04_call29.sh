#!/bin/bash
# run-wise k = 3 gather: its bit-exact tests, then the gather figures of the bench line
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${CALL_NAME:-r04_call29}
mkdir -p $OUT
cd $R
timeout -k 5 300 python -m pytest tests/test_gpu_parity.py tests/test_net_gpu.py -k "patch_gather or extract or net or R3 or provider" -x -q > $OUT/pytest_gather.log 2>&1; echo "gather tests rc $?"; tail -4 $OUT/pytest_gather.log
cd /tmp && export TMPDIR=/tmp
timeout -k 5 200 python $R/bench.py --no-cpu-baseline --no-block --no-pcie-f64 --no-pipelined > $OUT/bench_quick.json 2> $OUT/bench_quick.err; echo "bench rc $?"
python - $OUT/bench_quick.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("job_ms", d.get("job_ms"), "layers/s", d["value"], "parity", d.get("mask_parity_vs_reference_golden"))
print(json.dumps(d.get("patch_gather"))[:1500])
PY
