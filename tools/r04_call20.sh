#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_call20
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-gather --no-block --no-pcie-f64"
job() {  # name, env...
  local name=$1; shift
  env "$@" timeout -k 5 200 python $R/bench.py $Q --profile-mode --steps 3 --warmup 2 --jobs-per-step 12 > $OUT/job_$name.json 2> $OUT/job_$name.err
  python - $OUT/job_$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-22s job_ms %8.3f  layers/s %8.1f  parity %s" % (sys.argv[2], d.get("job_ms", -1), d["value"], d.get("mask_parity_vs_reference_golden")))
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
job base CP_NOP=1
for us in 100 200 400 700; do job stagger$us CP_JOB_STAGGER_US=$us; done
job base_b CP_NOP=1
job resnet_200 CP_BENCH_WORKLOAD=resnet50 CP_JOB_STAGGER_US=200
