#!/bin/bash
# Round-4 call 9 (GPU box): per-layer cost of the PCIe-inclusive pass; two ranks on one GPU (gloo) in strong mode.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_call9
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 400 python $R/bench.py --no-cpu-baseline --no-gather --no-block --steps 5 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - $OUT <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + "/bench.json").read().strip().splitlines()[-1])
print("job_ms", d["job_ms"], "scaling", d["scaling"], "bound", d.get("strong_scaling_bound"))
p = d["pcie_inclusive"]
print("pcie f32", p["job_ms_sequential_with_h2d"], "first", p["first_pass_ms"], "per layer", p.get("per_layer_ms"))
print("pcie f64", p.get("x_float64"))
print({k: v["ms_alone"] for k, v in d["per_layer_rank0"].items()})
PY
CP_BENCH_DIST_BACKEND=gloo timeout -k 5 600 python $R/bench.py --gpus 2 --steps 3 --warmup 1 --no-gather > $OUT/bench_2ranks_gloo_strong.json 2> $OUT/bench_2ranks.err; echo "2 ranks rc=$?"; tail -3 $OUT/bench_2ranks.err
python - $OUT <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1] + "/bench_2ranks_gloo_strong.json").read().strip().splitlines()[-1])
    print("2 ranks: value", d["value"], "job_ms", d["job_ms"], "scaling", d["scaling"], "parity", d["mask_parity_vs_reference_golden"])
    print("replica", d.get("replica_throughput"))
    print("bound", d.get("strong_scaling_bound"))
    print("exchange", d.get("exchange_rank0"))
except Exception as e:
    print("2 ranks unreadable", e)
PY
