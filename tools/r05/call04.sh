#!/bin/bash
# Round 5, GPU call 4: the look-ahead form of the diagonal role (CP_CHOL_DIAG bits) alone (chol_bulk: timings, phase stamps,
# U^T U check), then the refit / dictionary tests and the job.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_call04
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for F in 3 7; do
  timeout -k 5 120 $R/tools/ubench/chol_bulk 1 $F > $OUT/chol_bulk_form$F.md 2>&1
  echo "== form $F"; grep -E "all 36 steps|check:|step 5, diagonal" $OUT/chol_bulk_form$F.md | cut -c1-330
done
cd $R
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "refit or fc_kernel or golden_full_size or chol or concurrent_streams" < /dev/null > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
cd /tmp
Q="--steps 3 --warmup 2 --jobs-per-step 8 --no-cpu-baseline --no-block --no-gather --no-pcie-f64 --no-pipelined"
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
          "alone", {k: v.get("achieved") for k, v in (r.get("alone") or {}).items() if isinstance(v, dict)})
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
run form7 X=1 --
run form3 CP_CHOL_DIAG=3 --
run form7_c X=1 --
run form7_b X=1 --
