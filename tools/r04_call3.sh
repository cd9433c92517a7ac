#!/bin/bash
# Round-4 call 3 (GPU box): forward substitution fused into the Cholesky step launches; timeline of one job.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_call3
mkdir -p $OUT
cd $R
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "refit or fc_kernel or full_size or batch or resident or prefactored or nonlinear or vh_ or itq or resnet50_and_vgg16_5x or sharded" < /dev/null > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
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
job fused CP_NOP=1
job nofwd CP_CHOL_FUSE_FORWARD=0
job fused2 CP_NOP=1
job resnet_fused CP_BENCH_WORKLOAD=resnet50
job v5x_fused CP_BENCH_WORKLOAD=vgg16_5x
rm -rf /tmp/kt
timeout -k 5 200 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py --profile-mode --steps 1 --warmup 1 --jobs-per-step 3 > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
DB=$(find /tmp/kt -name '*.db' | head -1)
if [ -n "$DB" ]; then
  python $R/tools/rocpd_kernels.py $DB 5 > $OUT/kernels_vgg16.md 2>&1
  python $R/tools/rocpd_timeline.py $DB --streams=2 > $OUT/timeline_last40ms.md 2>&1
fi
head -12 $OUT/kernels_vgg16.md
head -40 $OUT/timeline_last40ms.md
