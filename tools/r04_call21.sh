#!/bin/bash
# matrix pipe + vector pipe probe, GPU suite of the restored tree, one quick vgg16 line
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_call21
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 120 $R/tools/ubench/mfma_valu_mix > $OUT/mfma_valu_mix.md 2>&1
cat $OUT/mfma_valu_mix.md
cd $R
timeout -k 5 400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -5 $OUT/pytest_gpu.log
cd /tmp
timeout -k 5 200 python $R/bench.py --no-cpu-baseline --no-gather --no-block --no-pcie-f64 > $OUT/bench_quick.json 2> $OUT/bench_quick.err; echo "bench rc $?"
python - $OUT/bench_quick.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("job_ms", d.get("job_ms"), "layers/s", d["value"], "parity", d.get("mask_parity_vs_reference_golden"))
print(json.dumps(d.get("roofline"))[:1500])
PY
