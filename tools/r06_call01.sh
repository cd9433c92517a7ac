#!/bin/bash
# round 6, call 1: GPU suite with the new tests, smoke, the driver's bench command (the compact line + bench_detail.json)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_call01
mkdir -p $OUT
cd $R
timeout -k 5 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1
tail -5 $OUT/pytest_gpu.log
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.log 2>&1
tail -1 $OUT/smoke.log
/usr/bin/time -v timeout -k 5 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail $OUT/bench_detail.json > $OUT/bench.json 2> $OUT/bench.err
wc -c $OUT/bench.json; wc -l $OUT/bench.json
cat $OUT/bench.json
grep -E "Elapsed|Maximum resident" $OUT/bench.err
