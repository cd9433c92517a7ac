#!/bin/bash
# Round 5, GPU call 3: k_chol_step's flag hand-off without agent-scope acquire / release (sc1 loads / stores instead of
# buffer_inv sc1 per poll + per wave and buffer_wbl2): suite incl. the new tests, chol_bulk alone, the vgg16 job.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_call03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
$R/tools/ubench/chol_bulk 1 > $OUT/chol_bulk.md 2>&1
tail -4 $OUT/chol_bulk.md | cut -c1-400
Q="--steps 3 --warmup 2 --jobs-per-step 8 --no-cpu-baseline --no-block --no-gather --no-pcie-f64"
run() {
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 5 240 python $R/bench.py $Q "$@" < /dev/null > $OUT/$name.json 2> $OUT/$name.err
  python - $OUT/$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    lat = r.get("latency_bound_chains_ms_per_job") or {}
    print(sys.argv[2], "job_ms", d.get("job_ms"), "parity", d.get("mask_parity_vs_reference_golden"),
          "search/back", [round(v, 1) for v in lat.values()], "gram/chol", [(k["sum_ms_per_job"], (k.get("chip_level") or {}).get("achieved")) for k in r.get("kernels", [])],
          "two", (d.get("two_jobs_in_flight") or {}).get("value"), "ns/step", (r.get("alpha_search") or {}).get("ns_per_step_in_the_job_by_channels"))
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
run job_a X=1 --
run job_pre3 X=1 -- --precompute-heaviest 3
run job_pre5 X=1 -- --precompute-heaviest 5
run job_side5 CP_SIDE_STREAM_PER_CTX=1 -- --precompute-heaviest 5
run job_b X=1 --
cd $R
timeout -k 5 900 python -m pytest tests -m gpu -q -x < /dev/null > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -5 $OUT/pytest.log
