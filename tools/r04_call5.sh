#!/bin/bash
# Round-4 call 5 (GPU box): 128-VGPR Cholesky step kernel (D-layout 16x16 factor) vs the 248-VGPR build; panel order; timeline.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_call5
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
job v128 CP_NOP=1
job v128_panel_last CP_CHOL_PANEL_LAST=1
job v248 CP_LIB_PATH=$R/build_variants/v248/libcpmi355.so
job v248_panel_last CP_LIB_PATH=$R/build_variants/v248/libcpmi355.so CP_CHOL_PANEL_LAST=1
job v128_b CP_NOP=1
job v128_panel_last_b CP_CHOL_PANEL_LAST=1
job resnet_v128 CP_BENCH_WORKLOAD=resnet50
job resnet_v128_pl CP_BENCH_WORKLOAD=resnet50 CP_CHOL_PANEL_LAST=1
job v5x_v128 CP_BENCH_WORKLOAD=vgg16_5x
job v5x_v128_pl CP_BENCH_WORKLOAD=vgg16_5x CP_CHOL_PANEL_LAST=1
timeout -k 5 300 python $R/bench.py $Q --steps 5 --warmup 2 > $OUT/bench_v128.json 2> $OUT/bench_v128.err; echo "bench rc=$?"
python - $OUT <<'PY'
import json, sys
for n in ("bench_v128",):
    try:
        d = json.loads(open(sys.argv[1] + "/" + n + ".json").read().strip().splitlines()[-1])
        print(n, "job_ms", d["job_ms"])
        for k, v in d["per_layer_rank0"].items():
            print("   ", k, v["ms_alone"], "search", v["alpha_search_ms"], "refit", v["refit_ms"])
        for k, v in d["stage_ms_alone_by_shape_rank0"].items():
            print("   ", k, {a: b for a, b in v.items() if "chol" in a or "solve" in a or "factor" in a or "subst" in a or "backward" in a})
    except Exception as e:
        print(n, "unreadable", e)
PY
rm -rf /tmp/kt
timeout -k 5 200 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py --profile-mode --steps 1 --warmup 1 --jobs-per-step 3 > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
DB=$(find /tmp/kt -name '*.db' | head -1)
if [ -n "$DB" ]; then
  python $R/tools/rocpd_timeline.py $DB --anchor=k_lasso_prep:12 --streams=1 > $OUT/timeline_last_job.md 2>&1
fi
sed -n 1,12p $OUT/timeline_last_job.md; grep -n "k_chol_step\|k_solve_strips\|k_gemm_tn_f64<0, 0\|finalize\|block_inverse" $OUT/timeline_last_job.md | sed -n 1,70p
