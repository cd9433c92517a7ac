#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_call11
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
( time timeout -k 5 900 python $R/bench.py --workload r3 --steps 2 --warmup 1 > $OUT/bench_r3.json 2> $OUT/bench_r3.err ) 2> $OUT/bench_r3.time; echo "r3 rc=$?"; tail -3 $OUT/bench_r3.time; grep -i "error\|Traceback" -A6 $OUT/bench_r3.err | tail -12
python - $OUT <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1] + "/bench_r3.json").read().strip().splitlines()[-1])
    print("r3 job_ms", d["job_ms"], "value", d["value"], d["stage_ms_per_job"])
    for k, v in d["per_conv"].items(): print("  ", k, v)
    print(d.get("cpu_baseline"))
except Exception as e:
    print("r3 unreadable", e)
PY
Q="--no-cpu-baseline --no-gather --no-block --no-pcie-f64"
job() {  # name, env...
  local name=$1; shift
  env "$@" timeout -k 5 200 python $R/bench.py $Q --profile-mode --steps 3 --warmup 2 --jobs-per-step 12 > $OUT/job_$name.json 2> $OUT/job_$name.err
  python - $OUT/job_$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-28s job_ms %8.3f  layers/s %8.1f  parity %s" % (sys.argv[2], d.get("job_ms", -1), d["value"], d.get("mask_parity_vs_reference_golden")))
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
for n in 0 1 2 3 4 5; do job precompute$n CP_JOB_PRECOMPUTE=$n; done
job prio CP_JOB_PRIORITY=1
timeout -k 5 400 python $R/bench.py --no-cpu-baseline --no-gather --no-block --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - $OUT <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + "/bench.json").read().strip().splitlines()[-1])
p = d["pcie_inclusive"]
print("job_ms", d["job_ms"], "pcie f32", p["job_ms_sequential_with_h2d"], "first", p["first_pass_ms"], "f64", p.get("x_float64", {}).get("job_ms_sequential_with_h2d"))
PY
