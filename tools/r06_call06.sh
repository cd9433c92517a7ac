#!/bin/bash
# round 6, call 6: backward sweep in one launch -- ubench checks, the refit / dictionary GPU tests, the vgg16 job A/B
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_call06
mkdir -p $OUT
cd $R
timeout -k 5 120 tools/ubench/chol_chain quick > $OUT/chol_chain_quick.md 2>&1
echo "quick rc=$?"; grep -E "identical|sweep ok|FAILED|WRONG|NO" $OUT/chol_chain_quick.md | head
timeout -k 5 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "refit or dictionary or fc_kernel or streamed or chol or reproducible or concurrent" > $OUT/pytest_subset.log 2>&1
tail -4 $OUT/pytest_subset.log
run() {
    name=$1; shift
    env "$@" timeout -k 5 300 python3 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-block --no-gather \
        --no-pcie-f64 --no-pipelined --detail $OUT/$name.detail.json > $OUT/$name.json 2> $OUT/$name.err
    python3 - "$OUT/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    det = json.load(open(sys.argv[1].replace(".json", ".detail.json")))
    print(sys.argv[2], "job_ms", d["job_ms"], "parity", d["mask_parity_vs_reference_golden"], "werr", d.get("weights_rel_frobenius_max"),
          "chol sum", r.get("sum_ms_per_job"), "gram sum", r["gram"]["sum_ms_per_job"], det["roofline"]["latency_bound_chains_ms_per_job"])
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run steps CP_CHOL_FORM=steps
run chain_back CP_CHOL_LAZY=4
run chain_noback CP_CHOL_BACK=0
run chain_back_w2 CP_CHOL_WG_PER_BLK=2
run chain_back_w4 CP_CHOL_WG_PER_BLK=4
run chain_back_L2 CP_CHOL_LAZY=2
run steps2 CP_CHOL_FORM=steps
run chain_back2 CP_CHOL_LAZY=4
