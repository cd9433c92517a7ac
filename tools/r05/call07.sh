#!/bin/bash
# Round 5, GPU call 7: which layers get their normal equations under the search -- the cost model's first two of the tied
# wide layers (CP_RSET_ADAPTIVE_PRECOMPUTE=0) or the two with the longest measured searches (default) -- with 2 and 3 of them.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_call07
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
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
          "chunks", [c["ms"] for c in d["chunks_rank0_last_job"]][:5], "steps", [v["cd_steps"] for k, v in d["per_layer_rank0"].items()][7:])
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
run adaptive2 X=1 --
run model2 CP_RSET_ADAPTIVE_PRECOMPUTE=0 --
run adaptive3 X=1 -- --precompute-heaviest 3
run adaptive2_b X=1 --
run model2_b CP_RSET_ADAPTIVE_PRECOMPUTE=0 --
run adaptive1 X=1 -- --precompute-heaviest 1
