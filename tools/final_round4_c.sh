#!/bin/bash
# Round-4 last check of the committed tree: suite, smoke, the default bench line.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final4c
mkdir -p $OUT
cd $R
timeout -k 5 400 python -m pytest tests -m gpu -q -x < /dev/null > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
timeout -k 5 120 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
cd /tmp && export TMPDIR=/tmp
timeout -k 5 400 python $R/bench.py < /dev/null > $OUT/bench_vgg16.json 2> $OUT/bench_vgg16.err; echo "vgg rc=$?"
python - $OUT/bench_vgg16.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d.get("roofline") or {}
print(d["value"], d.get("job_ms"), d.get("mask_parity_vs_reference_golden"), r.get("frac"), r.get("peak_measured"), r.get("traffic_source"),
      (d.get("cpu_baseline") or {}).get("job_speedup_wall_clock"), (d.get("two_jobs_in_flight") or {}).get("value"))
PY
