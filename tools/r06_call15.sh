#!/bin/bash
# round 6, call 15: workgroups of the persistent factorisation leave as the trailing matrix shrinks -- ubench (bit for bit, alone,
# five side by side) and the vgg16 job by the number kept per remaining block row
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_call15
mkdir -p $OUT
cd $R
timeout -k 5 120 tools/ubench/chol_chain quick keep=3 > $OUT/quick.md 2>&1; echo "quick rc=$?"; grep -E "identical|MISMATCH" $OUT/quick.md
timeout -k 5 300 tools/ubench/chol_chain keep=3 > $OUT/chol_chain_keep3.md 2>&1; tail -18 $OUT/chol_chain_keep3.md
run() {
    name=$1; shift
    env "$@" timeout -k 5 300 python3 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-gather --no-pcie-f64 --no-pipelined --no-block --detail $OUT/$name.detail.json > $OUT/$name.json 2> $OUT/$name.err
    python3 - "$OUT/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    det = json.load(open(sys.argv[1].replace(".json", ".detail.json")))
    print(sys.argv[2], "job_ms", d["job_ms"], "value", d["value"], "parity", d["mask_parity_vs_reference_golden"], "chol", r.get("sum_ms_per_job"), "gram", r["gram"]["sum_ms_per_job"],
          "backsub", det["roofline"]["latency_bound_chains_ms_per_job"]["backward_substitution (banded)"], "bound", d.get("strong_scaling_bound_ms"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run steps CP_CHOL_FORM=steps
run keep0 CP_CHOL_KEEP_PER_ROW=0
run keep3 CP_CHOL_KEEP_PER_ROW=3
run keep2 CP_CHOL_KEEP_PER_ROW=2
run keep1 CP_CHOL_KEEP_PER_ROW=1
run keep2_w5 CP_CHOL_KEEP_PER_ROW=2 CP_CHOL_WG_PER_BLK=5
run steps2 CP_CHOL_FORM=steps
run keep2b CP_CHOL_KEEP_PER_ROW=2
