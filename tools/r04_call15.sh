#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_call15
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
    r = d.get("roofline") or {}
    ks = {k["kernel"][:12]: k["sum_ms_per_job"] for k in r.get("kernels", [])}
    print("%-22s job_ms %8.3f  layers/s %8.1f  parity %s  sums %s" % (sys.argv[2], d.get("job_ms", -1), d["value"], d.get("mask_parity_vs_reference_golden"), ks))
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
job base CP_NOP=1
for n in 2 3 4 5 8; do job full$n CP_JOB_LATENCY_KIND=full CP_JOB_PRECOMPUTE=$n; done
job base_b CP_NOP=1
job resnet_full4 CP_BENCH_WORKLOAD=resnet50 CP_JOB_LATENCY_KIND=full CP_JOB_PRECOMPUTE=4
job v5x_full4 CP_BENCH_WORKLOAD=vgg16_5x CP_JOB_LATENCY_KIND=full CP_JOB_PRECOMPUTE=4
