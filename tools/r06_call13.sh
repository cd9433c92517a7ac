#!/bin/bash
# round 6, call 13: with 16 hardware queues -- the default line with the extras in this process and in a child process
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_call13
mkdir -p $OUT
cd $R
for M in 1 0; do
CP_BENCH_EXTRAS_INPROCESS=$M timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --detail $OUT/bench_$M.detail.json > $OUT/bench_$M.json 2> $OUT/bench_$M.err
python3 -c "
import json; d=json.loads(open('$OUT/bench_$M.json').read().strip().splitlines()[-1]); print('inprocess=$M', d['value'], d['job_ms'], d.get('value_conv3_block'), d.get('two_jobs_in_flight_layers_per_s'), d.get('pcie_inclusive_job_ms'), d.get('other_workloads'), d.get('r3'))"
done
