#!/bin/bash
# Round 5, GPU call 14: last check of the committed tree -- suite, smoke, the driver's own bench command.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_call14
mkdir -p $OUT
cd $R
timeout -k 5 500 python -m pytest tests -m gpu -q -x < /dev/null > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
timeout -k 5 120 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout -k 5 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 < /dev/null > $OUT/bench_driver_cmd.json 2> $OUT/bench_driver_cmd.err; echo "bench rc=$?"
python - $OUT/bench_driver_cmd.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("value", d["value"], "job_ms", d["job_ms"], "ms_per_step", d["ms_per_step"], "steps", d["steps"], "parity", d["mask_parity_vs_reference_golden"],
      "frac", r["frac"], "traffic", r["traffic"], r["traffic_source"], "cpu", d["cpu_baseline"]["job_speedup_wall_clock"], "block", d["value_conv3_block"]["value"],
      "two", d["two_jobs_in_flight"]["value"], "sampled", r.get("jobs_with_stage_brackets"))
PY
