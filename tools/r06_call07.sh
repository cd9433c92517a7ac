#!/bin/bash
# round 6, call 7: persistent factorisation against launch-per-step on the other workloads: resnet50, vgg16_5x, the conv3_x block
# alone, the sequential (PCIe-inclusive) pass, r3
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_call07
mkdir -p $OUT
cd $R
summ() {
    python3 - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    keys = ("value", "job_ms", "mask_parity_vs_reference_golden", "pcie_inclusive_job_ms", "value_conv3_block", "two_jobs_in_flight_layers_per_s", "r3")
    print(sys.argv[2], {k: d.get(k) for k in keys if d.get(k) is not None})
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
for FORM in steps chain; do
  for W in resnet50 vgg16_5x; do
    CP_CHOL_FORM=$FORM timeout -k 5 300 python3 bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline --no-gather --no-pcie-f64 --detail $OUT/${W}_$FORM.detail.json > $OUT/${W}_$FORM.json 2> $OUT/${W}_$FORM.err
    summ $OUT/${W}_$FORM.json ${W}_$FORM
  done
  CP_CHOL_FORM=$FORM timeout -k 5 300 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gather --no-extras --no-pcie-f64 --detail $OUT/vgg16_$FORM.detail.json > $OUT/vgg16_$FORM.json 2> $OUT/vgg16_$FORM.err
  summ $OUT/vgg16_$FORM.json vgg16_$FORM
  CP_CHOL_FORM=$FORM timeout -k 5 300 python3 bench.py --workload r3 --steps 2 --warmup 1 --no-cpu-baseline --detail '' > $OUT/r3_$FORM.json 2> $OUT/r3_$FORM.err
  summ $OUT/r3_$FORM.json r3_$FORM
done
