#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_call14
mkdir -p $OUT
cd $R
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "refit or fc_kernel or full_size or batch or resident or prefactored or multi_cu" < /dev/null > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
Q="--no-cpu-baseline --no-gather --no-block --no-pcie-f64"
job() {  # name, env...
  local name=$1; shift
  env "$@" timeout -k 5 200 python $R/bench.py $Q --profile-mode --steps 3 --warmup 2 --jobs-per-step 12 > $OUT/job_$name.json 2> $OUT/job_$name.err
  python - $OUT/job_$name.json $name <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    ks = {k["kernel"][:12]: k["sum_ms_per_job"] for k in r.get("kernels", [])}
    print("%-22s job_ms %8.3f  layers/s %8.1f  parity %s  sums %s" % (sys.argv[2], d.get("job_ms", -1), d["value"], d.get("mask_parity_vs_reference_golden"), ks))
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
}
job new CP_NOP=1
job pre CP_LIB_PATH=$R/build_variants/pre/libcpmi355.so
job new_b CP_NOP=1
job pre_b CP_LIB_PATH=$R/build_variants/pre/libcpmi355.so
job resnet_new CP_BENCH_WORKLOAD=resnet50
job v5x_new CP_BENCH_WORKLOAD=vgg16_5x
timeout -k 5 300 python $R/bench.py $Q --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - $OUT <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + "/bench.json").read().strip().splitlines()[-1])
print("job_ms", d["job_ms"], {k: v["ms_alone"] for k, v in d["per_layer_rank0"].items()})
for k, v in d["stage_ms_alone_by_shape_rank0"].items():
    print("   ", k, {a: b for a, b in v.items() if "chol" in a or "backward" in a})
p = d["pcie_inclusive"]; print("pcie", p["job_ms_sequential_with_h2d"], p["first_pass_ms"])
PY
cd $R && REPS=6 timeout -k 5 200 python tools/dropin_latency.py 512 256 2>&1 | tail -8
