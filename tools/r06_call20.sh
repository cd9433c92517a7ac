#!/bin/bash
# round 6, call 20: the new defaults of the persistent factorisation (4 launches, 6 workgroups per block row) on every workload
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_call20
mkdir -p $OUT
cd $R
timeout -k 5 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "refit or fc_kernel or persistent or chol or dictionary or resident or streamed" > $OUT/pytest.log 2>&1; tail -2 $OUT/pytest.log
summ() {
python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print(sys.argv[2], "value", d["value"], "job_ms", d["job_ms"], "parity", d["mask_parity_vs_reference_golden"], "ab", d.get("chol_form_ab_job_ms"), "chol", r.get("sum_ms_per_job"),
          "block", (d.get("value_conv3_block") or {}).get("ms_per_pass"), "2jobs", d.get("two_jobs_in_flight_layers_per_s"), "seq", d.get("pcie_inclusive_job_ms"), "bound", d.get("strong_scaling_bound_ms"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for W in vgg16 resnet50 vgg16_5x; do
  timeout -k 5 300 python3 bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline --no-gather --no-extras --no-pcie-f64 --detail $OUT/$W.detail.json > $OUT/$W.json 2> $OUT/$W.err
  summ $OUT/$W.json $W
done
