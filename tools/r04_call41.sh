#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${CALL_NAME:-r04_call41}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 400 python $R/bench.py < /dev/null > $OUT/bench_vgg16.json 2> $OUT/bench_vgg16.err; echo "vgg rc=$?"
python - $OUT/bench_vgg16.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline") or {}
print(d["value"], d.get("job_ms"), d.get("mask_parity_vs_reference_golden"), r.get("frac"), r.get("peak_measured"), r.get("cycles_per_mfma_measured"), r.get("traffic_source"),
      (d.get("cpu_baseline") or {}).get("job_speedup_wall_clock"), (d.get("two_jobs_in_flight") or {}).get("value"))
PY
