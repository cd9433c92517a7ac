#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_call17
mkdir -p $OUT
cd $R
timeout -k 5 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "sign or itq or vh" < /dev/null > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest.log
echo "--- itq profile (sign route)"; timeout -k 5 300 python tests/tools/itq_profile.py 2>&1 | tail -5
echo "--- itq profile (CP_ITQ_SIGN=0)"; CP_ITQ_SIGN=0 timeout -k 5 300 python tests/tools/itq_profile.py 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
timeout -k 5 600 python $R/bench.py --workload r3 --no-cpu-baseline --steps 2 --warmup 1 > $OUT/bench_r3.json 2> $OUT/bench_r3.err; echo "r3 rc=$?"
python - $OUT/bench_r3.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("r3 job_ms", d["job_ms"], d["stage_ms_per_job"])
print({k: v["itq_ms"] for k, v in d["per_conv"].items()})
PY
Q="--no-cpu-baseline --no-gather --no-block --no-pcie-f64"
for V in new pre; do
  E="CP_NOP=1"; [ $V = pre ] && E="CP_LIB_PATH=$R/build_variants/pre/libcpmi355.so"
  env $E timeout -k 5 300 python $R/bench.py $Q --steps 5 --warmup 2 > $OUT/bench_$V.json 2> $OUT/bench_$V.err; echo "bench $V rc=$?"
  python - $OUT/bench_$V.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("job_ms", d["job_ms"], {k[:3]: v["ms_alone"] for k, v in d["per_layer_rank0"].items()})
for k, v in d["stage_ms_alone_by_shape_rank0"].items():
    print("   ", k, {a: b for a, b in v.items() if "chol" in a or "backward" in a})
PY
done
