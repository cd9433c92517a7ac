#!/bin/bash
# Round 5, GPU call 10: the keepers' row ring with 4 register sets (rows requested 24-32 coordinate steps ahead instead of 8-16):
# cd_bench alone, bit-exactness tests, the job (is the in-job stretch of the search L2-miss latency?).
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_call10
mkdir -p $OUT
V=$R/build_variants/libcpmi355_ring4.so
cd /tmp && export TMPDIR=/tmp
CD_BENCH_C=256,512,1024,2048 CD_BENCH_FLAGS=0 timeout -k 5 200 python $R/tools/cd_bench.py > $OUT/cd_bench_ring2.txt 2>&1; grep "^c=" $OUT/cd_bench_ring2.txt | cut -c1-120
CD_BENCH_C=256,512,1024,2048 CD_BENCH_FLAGS=0 timeout -k 5 200 python $R/tools/cd_bench.py $V > $OUT/cd_bench_ring4.txt 2>&1; grep "^c=" $OUT/cd_bench_ring4.txt | cut -c1-120
cd $R
CP_LIB_PATH=$V timeout -k 5 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "cd_fit_bit_exact or soft_threshold or zero_diagonal or max_iter or golden_full_size or multi_cu" < /dev/null > $OUT/pytest_ring4.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest_ring4.log; tail -3 $OUT/pytest_ring4.log
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
          "search/back", [round(v, 1) for v in lat.values()], "ns/step", (r.get("alpha_search") or {}).get("ns_per_step_in_the_job_by_channels"))
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
run ring2 X=1 --
run ring4 CP_LIB_PATH=$V --
run ring2_b X=1 --
run ring4_b CP_LIB_PATH=$V --
run ring4_res CP_LIB_PATH=$V -- --workload resnet50
run ring2_res X=1 -- --workload resnet50
