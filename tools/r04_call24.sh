#!/bin/bash
# GEMM tests + one vgg16 line (quick)
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${CALL_NAME:-r04_call24}
mkdir -p $OUT
cd $R
timeout -k 5 200 python -m pytest tests/test_gpu_parity.py -k "gemm_tn or refit_matches or run_to_run" -x -q > $OUT/pytest_gemm.log 2>&1; echo "gemm tests rc $?"; tail -5 $OUT/pytest_gemm.log
cd /tmp && export TMPDIR=/tmp
timeout -k 5 200 python $R/bench.py --no-cpu-baseline --no-gather --no-block --no-pcie-f64 > $OUT/bench_quick.json 2> $OUT/bench_quick.err; echo "bench rc $?"
python - $OUT/bench_quick.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("job_ms", d.get("job_ms"), "layers/s", d["value"], "parity", d.get("mask_parity_vs_reference_golden"))
r = d.get("roofline", {})
for k in r.get("kernels", []):
    print({kk: k.get(kk) for kk in ("kernel", "achieved", "avg_launch_ms", "sum_ms_per_job")})
for k, v in d['per_layer_rank0'].items():
    print(' ', k, {kk: v.get(kk) for kk in ('ms_alone', 'kept', 'refit_ms')})
for k, v in d['stage_ms_alone_by_shape_rank0'].items():
    print(' ', k, {a: b for a, b in v.items() if 'gemm' in a or 'reduce' in a or 'chol' in a})
print([c['ms'] for c in d['chunks_rank0_last_job']])
PY
