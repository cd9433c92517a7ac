#!/bin/bash
# round 6, call 17: the persistent factorisation cut into 1 .. 4 launches at step boundaries (workgroups sized per phase)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_call17
mkdir -p $OUT
cd $R
timeout -k 5 120 tools/ubench/chol_chain quick > $OUT/quick.md 2>&1; echo "quick rc=$?"; grep -E "all identical|MISMATCH" $OUT/quick.md
CP_CHOL_PHASES=3 timeout -k 5 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "refit or fc_kernel or persistent or chol or dictionary_matches_reference_golden_full" > $OUT/pytest_p3.log 2>&1; tail -2 $OUT/pytest_p3.log
run() {
    name=$1; shift
    env "$@" timeout -k 5 300 python3 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-gather --no-pcie-f64 --no-pipelined --no-block --detail $OUT/$name.detail.json > $OUT/$name.json 2> $OUT/$name.err
    python3 - "$OUT/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    det = json.load(open(sys.argv[1].replace(".json", ".detail.json")))
    print(sys.argv[2], "job_ms", d["job_ms"], "ab", d.get("chol_form_ab_job_ms"), "parity", d["mask_parity_vs_reference_golden"], "chol", r.get("sum_ms_per_job"), "gram", r["gram"]["sum_ms_per_job"],
          "backsub", det["roofline"]["latency_bound_chains_ms_per_job"]["backward_substitution (banded)"], "bound", d.get("strong_scaling_bound_ms"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run p1 CP_CHOL_PHASES=1
run p2 CP_CHOL_PHASES=2
run p3 CP_CHOL_PHASES=3
run p4 CP_CHOL_PHASES=4
run p6 CP_CHOL_PHASES=6
run p3_w4 CP_CHOL_PHASES=3 CP_CHOL_WG_PER_BLK=4
run p1b CP_CHOL_PHASES=1
