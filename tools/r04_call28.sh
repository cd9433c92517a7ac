#!/bin/bash
# two jobs in flight (vgg16 line), the r3 workload with the VGPR-form MFMA kernels
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${CALL_NAME:-r04_call28}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 200 python $R/bench.py --no-cpu-baseline --no-gather --no-block --no-pcie-f64 > $OUT/bench_quick.json 2> $OUT/bench_quick.err; echo "bench rc $?"; tail -3 $OUT/bench_quick.err
python - $OUT/bench_quick.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("job_ms", d.get("job_ms"), "layers/s", d["value"], "parity", d.get("mask_parity_vs_reference_golden"))
print(d.get("two_jobs_in_flight"))
PY
timeout -k 5 300 python $R/bench.py --workload r3 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_r3.json 2> $OUT/bench_r3.err; echo "r3 rc $?"; tail -3 $OUT/bench_r3.err
python - $OUT/bench_r3.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("r3 ms_per_step", d.get("ms_per_step"), d.get("stage_ms_per_job"))
print({k: (v.get("vh_ms"), v.get("itq_ms"), v.get("prune_ms")) for k, v in d.get("per_conv", {}).items()})
PY
