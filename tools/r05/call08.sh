#!/bin/bash
# Round 5, GPU call 8: stage brackets read on every 4th timed job (the read-back is ~1 ms of host time per job).
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_call08
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for i in 1 2; do
timeout -k 5 300 python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-gather --no-pcie-f64 < /dev/null > $OUT/bench_$i.json 2> $OUT/bench_$i.err
python - $OUT/bench_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("job_ms", d["job_ms"], "value", d["value"], "parity", d["mask_parity_vs_reference_golden"], "jobs", d["config"]["jobs_timed"], r.get("jobs_with_stage_brackets"),
      "kern", [(k["sum_ms_per_job"], k["frac"], (k.get("chip_level") or {}).get("achieved")) for k in r["kernels"]], "lat", r["latency_bound_chains_ms_per_job"],
      "two", (d.get("two_jobs_in_flight") or {}).get("value"), "block", (d.get("value_conv3_block") or {}).get("value"))
PY
done
