#!/bin/bash
# chol_step with global_ instead of flat_ accesses: refit / golden tests, one vgg16 line, the one-layer trace
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${CALL_NAME:-r04_call32}
mkdir -p $OUT
cd $R
timeout -k 5 300 python -m pytest tests/test_gpu_parity.py -k "refit or full_size or fc_kernel or prefactored" -x -q > $OUT/pytest_refit.log 2>&1; echo "refit tests rc $?"; tail -4 $OUT/pytest_refit.log
cd /tmp && export TMPDIR=/tmp
timeout -k 5 200 python $R/bench.py --no-cpu-baseline --no-gather --no-block --no-pcie-f64 --no-pipelined > $OUT/bench_quick.json 2> $OUT/bench_quick.err; echo "bench rc $?"
python - $OUT/bench_quick.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("job_ms", d.get("job_ms"), "layers/s", d["value"], "parity", d.get("mask_parity_vs_reference_golden"))
r = d.get("roofline", {})
for k in r.get("kernels", []):
    print({kk: k.get(kk) for kk in ("kernel", "achieved", "avg_launch_ms", "sum_ms_per_job")})
for k, v in d['per_layer_rank0'].items():
    print(' ', k, {kk: v.get(kk) for kk in ('ms_alone', 'kept', 'refit_ms')})
for k, v in d['stage_ms_alone_by_shape_rank0'].items():
    print(' ', k, {a: b for a, b in v.items() if 'gemm' in a or 'chol' in a or 'backward' in a or 'forward' in a})
print([c['ms'] for c in d['chunks_rank0_last_job']])
PY
rm -rf /tmp/kt1
timeout -k 5 200 rocprofv3 --kernel-trace -d /tmp/kt1 -o r -- python $R/tools/one_layer_trace.py 512 > $OUT/one_layer.log 2> $OUT/one_layer.err
DB=$(find /tmp/kt1 -name '*.db' | head -1)
python $R/tools/rocpd_timeline.py $DB --anchor=k_lasso_prep:1 --streams=3 > $OUT/timeline_one_layer.md 2>&1
python $R/tools/rocpd_kernels.py $DB 3 > $OUT/kernels_one_layer.md 2>&1
head -8 $OUT/kernels_one_layer.md | cut -c1-200
grep "k_chol_step" $OUT/timeline_one_layer.md | sed -n '5,22p' | cut -c1-120
