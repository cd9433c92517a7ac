#!/bin/bash
# Round 5, GPU call 13: the chip-filling products of all streams taking turns (event chain across streams, CP_GEMM_SERIALIZE=1)
# against interleaving; the driver's own command line for reference.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_call13
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
          "search/back", [round(v, 1) for v in lat.values()], "gram/chol", [(k["sum_ms_per_job"], (k.get("chip_level") or {}).get("achieved")) for k in r.get("kernels", [])])
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
run inter X=1 --
run serial CP_GEMM_SERIALIZE=1 --
run inter_b X=1 --
run serial_b CP_GEMM_SERIALIZE=1 --
run serial_5x CP_GEMM_SERIALIZE=1 -- --workload vgg16_5x
run inter_5x X=1 -- --workload vgg16_5x
