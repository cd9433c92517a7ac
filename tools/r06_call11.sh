#!/bin/bash
# round 6, call 11: why the short resnet50 / vgg16_5x / R3 legs of the default run are 40 % slower than the same jobs in a process of their own
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_call11
mkdir -p $OUT
cd $R
show() { python3 -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1]); print('$2', d['job_ms'], d.get('other_workloads'), d.get('r3'))"; }
timeout 400 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gather --no-pcie-f64 --detail '' > $OUT/a.json 2> $OUT/a.err; show $OUT/a.json "no cpu leg"
timeout 400 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gather --no-pcie-f64 --no-block --no-pipelined --detail '' > $OUT/b.json 2> $OUT/b.err; show $OUT/b.json "no cpu leg, no block, no pipelined"
timeout 400 python3 - <<'PY'
import sys, time
sys.path.insert(0, '.')
import bench
from benchkit.extras import short_job
from benchkit.r3 import r3_short_pass
print("extras alone:", {j: short_job(0, j) for j in ("resnet50", "vgg16_5x")}, r3_short_pass(0))
PY
