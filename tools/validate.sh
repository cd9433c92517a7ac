#!/bin/bash
# One GPU-box call that re-validates a build: the factorisation micro-benchmark (the persistent form against the launch-per-step
# form, bit for bit), the GPU suite, smoke(), and the bench line exactly as the driver runs it.  Outputs under gpurun_out/validate/.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/validate
mkdir -p $OUT
cd $R
timeout -k 5 120 tools/ubench/chol_chain quick > $OUT/chol_chain_quick.md 2>&1
timeout -k 5 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
tail -3 $OUT/pytest_gpu.log
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1
tail -1 $OUT/smoke.log
timeout -k 5 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/validate/bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "job_ms", "value_conv3_block", "chol_form_ab_job_ms") if k in d}, d.get("mask_parity_vs_reference_golden"), len(open("gpurun_out/validate/bench.json").read()), "bytes")
PY
grep -E "identical|MISMATCH" $OUT/chol_chain_quick.md
