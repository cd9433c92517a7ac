#!/bin/bash
# Round 5, GPU call 12: the assisted layer on the GPU (two ranks on this one GPU, gloo), bench.py --gpus 2 flows:
# exchange in rounds, and one layer row-assisted (forced: the plan asks for a rank with slack, which two ranks do not have).
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_call12
mkdir -p $OUT
cd $R
timeout -k 5 600 python -m pytest tests/test_gpu_rowshard.py tests/test_gpu_parity.py -m gpu -q -x -k "helped or rccl or two_ranks or sharded" < /dev/null > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
CP_BENCH_DIST_BACKEND=gloo timeout -k 5 400 python bench.py --gpus 2 --steps 3 --warmup 1 --no-gather < /dev/null > $OUT/bench_2ranks_rounds.json 2> $OUT/bench_2ranks_rounds.err; echo "2 ranks rounds rc=$?"
CP_BENCH_ASSISTS=7:1 CP_BENCH_DIST_BACKEND=gloo timeout -k 5 400 python bench.py --gpus 2 --steps 3 --warmup 1 --no-gather < /dev/null > $OUT/bench_2ranks_assist.json 2> $OUT/bench_2ranks_assist.err; echo "2 ranks assist rc=$?"
for f in rounds assist; do python - $OUT/bench_2ranks_$f.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    c = d["config"]
    print(sys.argv[1].split("_")[-1], d["value"], d.get("job_ms"), "parity", d.get("mask_parity_vs_reference_golden"), "rounds", c.get("exchange_round_of_layer"),
          "assisted", c.get("row_assisted_layers"), c.get("row_assist_timings_rank0_ms"), "exch", (d.get("exchange_rank0") or {}).get("rounds"), "replica", (d.get("replica_throughput") or {}).get("value"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
tail -3 $OUT/bench_2ranks_$f.err | cut -c1-300
done
