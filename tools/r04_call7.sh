#!/bin/bash
# Round-4 call 7 (GPU box): whole GPU suite on the cleaned-up library, the new bench line, --sequential-alpha.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_call7
mkdir -p $OUT
cd $R
timeout -k 5 900 python -m pytest tests -m gpu -q -x < /dev/null > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
timeout -k 5 120 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
cd /tmp && export TMPDIR=/tmp
( time timeout -k 5 600 python $R/bench.py > $OUT/bench_vgg16.json 2> $OUT/bench_vgg16.err ) 2> $OUT/bench_vgg16.time; echo "bench rc=$?"; tail -3 $OUT/bench_vgg16.time
timeout -k 5 400 python $R/bench.py --sequential-alpha --steps 5 --warmup 2 > $OUT/bench_seq_alpha.json 2> $OUT/bench_seq_alpha.err; echo "seq rc=$?"
python - $OUT <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + "/bench_vgg16.json").read().strip().splitlines()[-1])
print("job_ms", d["job_ms"], "value", d["value"], "parity", d["mask_parity_vs_reference_golden"])
print(json.dumps(d["roofline"], indent=1)[:3000])
print(json.dumps(d["cpu_baseline"], indent=1)[:1200])
print("pcie", d["pcie_inclusive"])
s = json.loads(open(sys.argv[1] + "/bench_seq_alpha.json").read().strip().splitlines()[-1])
print("seq alpha: job_ms", s["job_ms"], "value", s["value"], "masks==cpu", s.get("masks_identical_to_cpu_port_with_carry"), "cpu", s.get("cpu_baseline", {}).get("job_seconds_cpu"))
PY
