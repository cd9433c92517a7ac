#!/bin/bash
# round 6, call 2: the driver's bench command (the compact line + bench_detail.json), wall clock around it
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_call02
mkdir -p $OUT
cd $R
T0=$(date +%s.%N)
timeout -k 5 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail $OUT/bench_detail.json > $OUT/bench.json 2> $OUT/bench.err
echo "rc=$? wall=$(echo "$(date +%s.%N) - $T0" | bc) s"
wc -c $OUT/bench.json; wc -l $OUT/bench.json
cat $OUT/bench.json
tail -5 $OUT/bench.err
