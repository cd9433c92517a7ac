#!/bin/bash
# Round-4 call 6 (GPU box): block inverse inside the diagonal role; long products on a lowest-priority stream; band width.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_call6
mkdir -p $OUT
cd $R
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "refit or fc_kernel or full_size or batch or resident or prefactored" < /dev/null > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-gather --no-block --no-pcie-f64"
job() {  # name, env...
  local name=$1; shift
  env "$@" timeout -k 5 200 python $R/bench.py $Q --profile-mode --steps 3 --warmup 2 --jobs-per-step 12 > $OUT/job_$name.json 2> $OUT/job_$name.err
  python - $OUT/job_$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("%-28s job_ms %8.3f  layers/s %8.1f  parity %s  gram_ms %s" % (sys.argv[2], d.get("job_ms", -1), d["value"], d.get("mask_parity_vs_reference_golden"), (d.get("roofline") or {}).get("avg_launch_ms")))
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
job base CP_NOP=1
job lowprio CP_WIDE_LOWPRIO=1
job lowprio_hi CP_WIDE_LOWPRIO=1 CP_CTX_PRIORITY=-1
job hi_only CP_CTX_PRIORITY=-1
job ob8 CP_SOLVE_OB=8
job ob8_lowprio CP_SOLVE_OB=8 CP_WIDE_LOWPRIO=1
job base_b CP_NOP=1
job lowprio_b CP_WIDE_LOWPRIO=1
job resnet_lowprio CP_BENCH_WORKLOAD=resnet50 CP_WIDE_LOWPRIO=1
job v5x_lowprio CP_BENCH_WORKLOAD=vgg16_5x CP_WIDE_LOWPRIO=1
rm -rf /tmp/kt
CP_WIDE_LOWPRIO=1 timeout -k 5 200 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py --profile-mode --steps 1 --warmup 1 --jobs-per-step 3 > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
DB=$(find /tmp/kt -name '*.db' | head -1)
if [ -n "$DB" ]; then
  python $R/tools/rocpd_timeline.py $DB --anchor=k_lasso_prep:12 --streams=1 > $OUT/timeline_last_job_lowprio.md 2>&1
fi
sed -n 1,12p $OUT/timeline_last_job_lowprio.md; grep -n "k_chol_step" $OUT/timeline_last_job_lowprio.md | sed -n 2,40p
