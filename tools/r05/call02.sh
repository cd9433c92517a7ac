#!/bin/bash
# Round 5, GPU call 2: where the serial piece of a factorisation step goes (chol_bulk with stamps in the diagonal and panel roles),
# s_setprio for the serial roles, lower HIP priority for the streams of the narrow layers.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_call02
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
$R/tools/ubench/chol_bulk 0 > $OUT/chol_bulk_prio0.md 2>&1
$R/tools/ubench/chol_bulk 1 > $OUT/chol_bulk_prio1.md 2>&1
tail -5 $OUT/chol_bulk_prio1.md
Q="--steps 3 --warmup 2 --jobs-per-step 8 --no-cpu-baseline --no-block --no-pipelined --no-gather --no-pcie-f64"
run() {
  local name=$1; shift
  local envs=()
  while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout -k 5 200 python $R/bench.py $Q "$@" < /dev/null > $OUT/$name.json 2> $OUT/$name.err
  python - $OUT/$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    lat = r.get("latency_bound_chains_ms_per_job") or {}
    print(sys.argv[2], "job_ms", d.get("job_ms"), "parity", d.get("mask_parity_vs_reference_golden"),
          "search/back", [round(v, 1) for v in lat.values()], "gram/chol", [(k["sum_ms_per_job"]) for k in r.get("kernels", [])],
          "chunks", [c["ms"] for c in d["chunks_rank0_last_job"]])
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
run prio1 X=1 --
run prio0 CP_CHOL_PRIO=0 --
run low512 CP_RSET_LOW_PRIO_BELOW=512 --
run low256 CP_RSET_LOW_PRIO_BELOW=256 --
run low512_prio0 CP_RSET_LOW_PRIO_BELOW=512 CP_CHOL_PRIO=0 --
run low512_pre3 CP_RSET_LOW_PRIO_BELOW=512 -- --precompute-heaviest 3
run prio1_b X=1 --
