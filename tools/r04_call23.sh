#!/bin/bash
# the GEMM with tail split / in-kernel reduction / fused mirror: its own test first, then the suite, then one vgg16 line
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_call23
mkdir -p $OUT
cd $R
timeout -k 5 200 python -m pytest tests/test_gpu_parity.py -k "gemm_tn or refit_matches or run_to_run" -x -q > $OUT/pytest_gemm.log 2>&1; echo "gemm tests rc $?"; tail -15 $OUT/pytest_gemm.log
timeout -k 5 400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -8 $OUT/pytest_gpu.log
cd /tmp && export TMPDIR=/tmp
timeout -k 5 200 python $R/bench.py --no-cpu-baseline --no-gather --no-block --no-pcie-f64 > $OUT/bench_quick.json 2> $OUT/bench_quick.err; echo "bench rc $?"
python - $OUT/bench_quick.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("job_ms", d.get("job_ms"), "layers/s", d["value"], "parity", d.get("mask_parity_vs_reference_golden"))
r = d.get("roofline", {})
print({k: r.get(k) for k in ("kernel", "achieved", "frac", "peak_measured", "cycles_per_mfma_measured", "effective_ghz")})
for k in r.get("kernels", []):
    print({kk: k.get(kk) for kk in ("kernel", "achieved", "avg_launch_ms", "sum_ms_per_job", "alone")})
pl = d.get("per_layer", {})
for name in list(pl)[:12]:
    print(name, {kk: pl[name].get(kk) for kk in ("ms_alone", "search_ms", "refit_ms")})
PY
