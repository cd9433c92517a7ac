#!/bin/bash
# Round-4 diagnostics (GPU box): what clock the f64 MFMA pipe runs at, what a dependent chain's next kernel waits for on
# a busy chip (by resource footprint), and whether bounded GEMM grids (persistent tile loop) shorten the job.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_call1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CP_BENCH_EPOCH=1
Q="--no-cpu-baseline --no-gather --no-block --no-pcie-f64"
timeout -k 5 120 $R/tools/ubench/mfma_clock > $OUT/mfma_clock.md 2>&1; echo "mfma_clock rc=$?"
# idle baseline of the side-car, then next to the default job
timeout -k 5 30 $R/tools/ubench/sidecar 4 > $OUT/sidecar_idle.md 2>&1
( timeout -k 5 80 $R/tools/ubench/sidecar 45 > $OUT/sidecar_job.md 2>&1 ) &
SC=$!
sleep 1
timeout -k 5 300 python $R/bench.py $Q --steps 10 --warmup 3 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench default rc=$?"
wait $SC
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
job default CP_NOP=1
job default2 CP_NOP=1
for cap in 64 128 192 256 384 512; do job gridcap$cap CP_GEMM_GRID_CAP=$cap; done
job choltasks CP_CHOL_TASKS=1
job hwq8 GPU_MAX_HW_QUEUES=8
job hwq64 GPU_MAX_HW_QUEUES=64
job resnet_default CP_BENCH_WORKLOAD=resnet50
job resnet_cap128 CP_BENCH_WORKLOAD=resnet50 CP_GEMM_GRID_CAP=128
# the persistent tile loop against the goldens of the vgg16 job
cd $R
CP_GEMM_GRID_CAP=128 timeout -k 5 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden" < /dev/null > $OUT/pytest_cap128.log 2>&1; echo "pytest cap128 rc=$?"; tail -2 $OUT/pytest_cap128.log
cd /tmp
# shader clock per kernel, each alone (counter passes serialise the dispatches)
rm -rf /tmp/pm_clk
timeout -k 5 200 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d /tmp/pm_clk -o r -- python $R/bench.py --profile-mode --steps 1 --warmup 1 --jobs-per-step 1 > /dev/null 2> $OUT/pm_clk.err
DB=$(find /tmp/pm_clk -name '*.db' | head -1)
[ -n "$DB" ] && python $R/tools/rocpd_clock.py $DB > $OUT/clock_per_kernel.md 2>&1
ls -la $OUT
