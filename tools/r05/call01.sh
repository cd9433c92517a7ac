#!/bin/bash
# Round 5, GPU call 1: suite on the tree with the advisor fixes; scheduling experiments on the vgg16 job
#   (kernarg channel list on/off, per-context side streams with 2..5 layers' normal equations under the searches);
#   full per-stream rocprof timeline of one job.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_call01
mkdir -p $OUT
cd $R
timeout -k 5 420 python -m pytest tests -m gpu -q -x < /dev/null > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
Q="--steps 3 --warmup 2 --jobs-per-step 8 --no-cpu-baseline --no-block --no-pipelined --no-gather --no-pcie-f64"
run() {  # name, env..., -- extra args
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
    print(sys.argv[2], "job_ms", d.get("job_ms"), "value", d["value"], "parity", d.get("mask_parity_vs_reference_golden"),
          "search", list(lat.values())[:1], "kern", [(k["sum_ms_per_job"]) for k in r.get("kernels", [])])
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
run base_memcpy CP_REFIT_CHAN_KERNARG=0 --
run kernarg X=1 --
run kernarg_b X=1 --
run side2 CP_SIDE_STREAM_PER_CTX=1 -- --precompute-heaviest 2
run side3 CP_SIDE_STREAM_PER_CTX=1 -- --precompute-heaviest 3
run side4 CP_SIDE_STREAM_PER_CTX=1 -- --precompute-heaviest 4
run side5 CP_SIDE_STREAM_PER_CTX=1 -- --precompute-heaviest 5
run shared5 X=1 -- --precompute-heaviest 5
rm -rf /tmp/kt
timeout -k 5 240 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py --profile-mode --steps 2 --warmup 1 --jobs-per-step 4 > $OUT/under_rocprof.json 2> $OUT/kt.err
DB=$(find /tmp/kt -name '*.db' | head -1)
[ -n "$DB" ] && python $R/tools/rocpd_timeline.py $DB --anchor=k_lasso_prep:12 --streams=14 > $OUT/timeline_all_streams.md 2>&1
rm -rf /tmp/kt5
CP_SIDE_STREAM_PER_CTX=1 timeout -k 5 240 rocprofv3 --kernel-trace -d /tmp/kt5 -o r -- python $R/bench.py --profile-mode --steps 2 --warmup 1 --jobs-per-step 4 --precompute-heaviest 5 > $OUT/under_rocprof_side5.json 2> $OUT/kt5.err
DB=$(find /tmp/kt5 -name '*.db' | head -1)
[ -n "$DB" ] && python $R/tools/rocpd_timeline.py $DB --anchor=k_lasso_prep:12 --streams=18 > $OUT/timeline_side5.md 2>&1
ls -la $OUT
