#!/bin/bash
# Round 5, GPU call 11: panel rows stored as they become final (streaming form); suite; the job.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_call11
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 120 $R/tools/ubench/chol_bulk 1 7 > $OUT/chol_bulk_form7.md 2>&1
grep -E "all 36 steps|check:|step 5, " $OUT/chol_bulk_form7.md | cut -c1-420
cd $R
timeout -k 5 900 python -m pytest tests -m gpu -q -x < /dev/null > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
cd /tmp
for i in 1 2; do
timeout -k 5 300 python $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-gather --no-pcie-f64 --no-block --no-pipelined < /dev/null > $OUT/bench_$i.json 2> $OUT/bench_$i.err
python - $OUT/bench_$i.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print("job_ms", d["job_ms"], "value", d["value"], "parity", d["mask_parity_vs_reference_golden"],
      "kern", [(k["sum_ms_per_job"], k["frac"], (k.get("chip_level") or {}).get("achieved")) for k in r["kernels"]], "lat", r["latency_bound_chains_ms_per_job"])
PY
done
