#!/bin/bash
# Round-4 call 8 (GPU box): uploads streamed behind the alpha search (cp_prune_layer_h2d).
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_call8
mkdir -p $OUT
cd $R
timeout -k 5 900 python -m pytest tests/test_gpu_parity.py tests/test_net_gpu.py -m gpu -q -x -k "dictionary or golden or net or R3 or dropin or prune_layer or inputs_not_modified" < /dev/null > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 python $R/tools/dropin_latency.py 512 256 64 > $OUT/dropin_latency.txt 2>&1; echo "dropin rc=$?"; cat $OUT/dropin_latency.txt
timeout -k 5 400 python $R/bench.py --no-cpu-baseline --no-gather --no-block --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - $OUT <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + "/bench.json").read().strip().splitlines()[-1])
print("job_ms", d["job_ms"], "parity", d["mask_parity_vs_reference_golden"])
p = d["pcie_inclusive"]
print("pcie f32", p["job_ms_sequential_with_h2d"], "first", p["first_pass_ms"], "f64", p.get("x_float64"))
print({k: v["ms_alone"] for k, v in d["per_layer_rank0"].items()})
PY
