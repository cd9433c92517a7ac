#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_call10
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 python $R/tools/seq_pass_probe.py > $OUT/seq_pass_probe.txt 2>&1; echo "probe rc=$?"; cat $OUT/seq_pass_probe.txt | cut -c1-700
CP_BENCH_DIST_BACKEND=gloo timeout -k 5 600 python $R/bench.py --gpus 2 --steps 3 --warmup 1 --no-gather > $OUT/bench_2ranks_gloo_strong.json 2> $OUT/bench_2ranks.err; echo "2 ranks rc=$?"; grep -i "error" $OUT/bench_2ranks.err | tail -3
python - $OUT <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1] + "/bench_2ranks_gloo_strong.json").read().strip().splitlines()[-1])
    print("2 ranks: value", d["value"], "job_ms", d["job_ms"], "scaling", d["scaling"], "parity", d["mask_parity_vs_reference_golden"])
    print("replica", d.get("replica_throughput"))
    b = d.get("strong_scaling_bound") or {}
    print("bound", {k: v for k, v in b.items() if k not in ("note", "row_sharding")})
    print("exchange", d.get("exchange_rank0"))
except Exception as e:
    print("2 ranks unreadable", e)
PY
