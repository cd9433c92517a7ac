#!/bin/bash
# Round 5, GPU call 5: publication after the barrier (k_chol_step, streaming form) + the coefficient lay-out riding in the
# backward substitution band by band (CP_REFIT_BAND_FINAL): ubench, the whole GPU suite, the job with / without.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_call05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 120 $R/tools/ubench/chol_bulk 1 7 > $OUT/chol_bulk_form7.md 2>&1
grep -E "all 36 steps|check:|step 5, " $OUT/chol_bulk_form7.md | cut -c1-420
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
          "alone", {k: v.get("achieved") for k, v in (r.get("alone") or {}).items() if isinstance(v, dict)},
          "c512 alone", [v["ms_alone"] for k, v in d["per_layer_rank0"].items() if k.startswith(("V08", "V12"))])
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
run band1 X=1 --
run band0 CP_REFIT_BAND_FINAL=0 --
run band1_b X=1 --
run band0_b CP_REFIT_BAND_FINAL=0 --
cd $R
timeout -k 5 900 python -m pytest tests -m gpu -q -x < /dev/null > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
