#!/bin/bash
# round 6, call 10: X^T Y on the context's auxiliary stream next to the Gram product -- refit tests, the three jobs A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_call10
mkdir -p $OUT
cd $R
timeout -k 5 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rowshard.py -m gpu -x -q -k "refit or dictionary or fc_kernel or streamed or chol or reproducible or concurrent or batch or resident or sharded" > $OUT/pytest_subset.log 2>&1
tail -3 $OUT/pytest_subset.log
summ() {
python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print(sys.argv[2], "job_ms", d["job_ms"], "value", d["value"], "parity", d["mask_parity_vs_reference_golden"], "chol", r.get("sum_ms_per_job"), "gram", (r.get("gram") or {}).get("sum_ms_per_job"),
          "block", (d.get("value_conv3_block") or {}).get("ms_per_pass"), "2jobs", d.get("two_jobs_in_flight_layers_per_s"), "seq", d.get("pcie_inclusive_job_ms"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for A in 1 0 1 0; do
  CP_XTY_ASIDE=$A timeout -k 5 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gather --no-extras --no-pcie-f64 --detail $OUT/vgg16_a$A.detail.json > $OUT/vgg16_a$A.json 2> $OUT/vgg16_a$A.err
  summ $OUT/vgg16_a$A.json vgg16_aside$A
done
for W in resnet50 vgg16_5x; do
  for A in 1 0; do
    CP_XTY_ASIDE=$A timeout -k 5 300 python3 bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline --no-gather --no-pcie-f64 --no-pipelined --detail '' > $OUT/${W}_a$A.json 2> $OUT/${W}_a$A.err
    summ $OUT/${W}_a$A.json ${W}_aside$A
  done
done
