#!/bin/bash
# last rebuild of the round (unused helpers removed): GEMM / refit / golden tests and one vgg16 line
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${CALL_NAME:-r04_call42}
mkdir -p $OUT
cd $R
timeout -k 5 300 python -m pytest tests/test_gpu_parity.py -k "gemm_tn or refit or full_size or run_to_run" -x -q > $OUT/pytest_sel.log 2>&1; echo "tests rc $?"; tail -3 $OUT/pytest_sel.log
timeout -k 5 120 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
cd /tmp && export TMPDIR=/tmp
timeout -k 5 200 python $R/bench.py --no-cpu-baseline --no-gather --no-block --no-pcie-f64 --no-pipelined > $OUT/bench_quick.json 2> $OUT/bench_quick.err; echo "bench rc $?"
python - $OUT/bench_quick.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("job_ms", d.get("job_ms"), "layers/s", d["value"], "parity", d.get("mask_parity_vs_reference_golden"), r["peak_measured"], r["cycles_per_mfma_measured"])
PY
