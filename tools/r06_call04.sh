#!/bin/bash
# round 6, call 4: the vgg16 job with the persistent factorisation (grid size / lazy period sweep) against the launch-per-step form
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_call04
mkdir -p $OUT
cd $R
run() {   # name, env...
    name=$1; shift
    env "$@" timeout -k 5 300 python3 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-block --no-gather \
        --no-pcie-f64 --no-pipelined --detail $OUT/$name.detail.json > $OUT/$name.json 2> $OUT/$name.err
    python3 - "$OUT/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    print(sys.argv[2], "job_ms", d["job_ms"], "value", d["value"], "parity", d["mask_parity_vs_reference_golden"], "werr", d.get("weights_rel_frobenius_max"),
          "chol sum", r.get("sum_ms_per_job"), "chip", r.get("chip_level_frac"), "gram sum", r["gram"]["sum_ms_per_job"], "cd ns", r.get("alpha_search_ns_per_step"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run steps CP_CHOL_FORM=steps
run chain_L4_w3 CP_CHOL_LAZY=4 CP_CHOL_WG_PER_BLK=3
run chain_L2_w3 CP_CHOL_LAZY=2 CP_CHOL_WG_PER_BLK=3
run chain_L4_w2 CP_CHOL_LAZY=4 CP_CHOL_WG_PER_BLK=2
run chain_L4_w5 CP_CHOL_LAZY=4 CP_CHOL_WG_PER_BLK=5
run chain_L4_w8 CP_CHOL_LAZY=4 CP_CHOL_WG_PER_BLK=8
run steps_again CP_CHOL_FORM=steps
