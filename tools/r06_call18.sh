#!/bin/bash
# round 6, call 18: phases x workgroups per block row of the persistent factorisation, in the job
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_call19
mkdir -p $OUT
cd $R
run() {
    name=$1; shift
    env "$@" timeout -k 5 300 python3 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-gather --no-pcie-f64 --no-pipelined --no-block --detail $OUT/$name.detail.json > $OUT/$name.json 2> $OUT/$name.err
    python3 - "$OUT/$name.json" "$name" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d["roofline"]
    det = json.load(open(sys.argv[1].replace(".json", ".detail.json")))
    print(sys.argv[2], "job_ms", d["job_ms"], "ab", d.get("chol_form_ab_job_ms"), "chol", r.get("sum_ms_per_job"), "gram", r["gram"]["sum_ms_per_job"],
          "backsub", det["roofline"]["latency_bound_chains_ms_per_job"]["backward_substitution (banded)"], "bound", d.get("strong_scaling_bound_ms"))
except Exception as e:
    print(sys.argv[2], "FAILED", e)
PY
}
run p3_w5 CP_CHOL_PHASES=3 CP_CHOL_WG_PER_BLK=5
run p4_w5 CP_CHOL_PHASES=4 CP_CHOL_WG_PER_BLK=5
run p3_w5 CP_CHOL_PHASES=3 CP_CHOL_WG_PER_BLK=5
run p4_w5 CP_CHOL_PHASES=4 CP_CHOL_WG_PER_BLK=5
run p4_w6 CP_CHOL_PHASES=4 CP_CHOL_WG_PER_BLK=6
run p5_w6 CP_CHOL_PHASES=5 CP_CHOL_WG_PER_BLK=6
run p4_w7 CP_CHOL_PHASES=4 CP_CHOL_WG_PER_BLK=7
run p6_w7 CP_CHOL_PHASES=6 CP_CHOL_WG_PER_BLK=7
