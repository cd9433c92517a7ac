#!/bin/bash
# round 6, call 14: persistent factorisation against launch-per-step once more, with 16 hardware queues
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_call14
mkdir -p $OUT
cd $R
run() {
    name=$1; shift
    env "$@" timeout -k 5 300 python3 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-gather --no-pcie-f64 --detail $OUT/$name.detail.json > $OUT/$name.json 2> $OUT/$name.err
    python3 - "$OUT/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(sys.argv[2], "job_ms", d["job_ms"], "value", d["value"], "parity", d["mask_parity_vs_reference_golden"], "chol", r.get("sum_ms_per_job"), "gram", r["gram"]["sum_ms_per_job"],
          "block", d["value_conv3_block"]["ms_per_pass"], "2jobs", d.get("two_jobs_in_flight_layers_per_s"), "seq", d.get("pcie_inclusive_job_ms"), "bound", d.get("strong_scaling_bound_ms"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run steps CP_CHOL_FORM=steps
run chain X=1
run steps2 CP_CHOL_FORM=steps
run chain2 X=1
run chain_w2 CP_CHOL_WG_PER_BLK=2
run chain_w5 CP_CHOL_WG_PER_BLK=5
run chain_L2 CP_CHOL_LAZY=2
