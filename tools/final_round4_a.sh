#!/bin/bash
# Round-4 final check (GPU box), part A: suite, smoke, the bench lines.  Every step bounded; nothing reads stdin.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final4
mkdir -p $OUT
cd $R
timeout -k 5 400 python -m pytest tests -m gpu -q -x < /dev/null > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
timeout -k 5 120 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
cd /tmp && export TMPDIR=/tmp
timeout -k 5 400 python $R/bench.py < /dev/null > $OUT/bench_vgg16.json 2> $OUT/bench_vgg16.err; echo "vgg rc=$?"
timeout -k 5 300 python $R/bench.py --workload resnet50 --no-cpu-baseline < /dev/null > $OUT/bench_resnet50.json 2> $OUT/bench_resnet50.err; echo "resnet rc=$?"
timeout -k 5 300 python $R/bench.py --workload vgg16_5x --no-cpu-baseline < /dev/null > $OUT/bench_vgg16_5x.json 2> $OUT/bench_vgg16_5x.err; echo "5x rc=$?"
timeout -k 5 300 python $R/bench.py --workload r3 --steps 2 --warmup 1 --no-cpu-baseline < /dev/null > $OUT/bench_r3.json 2> $OUT/bench_r3.err; echo "r3 rc=$?"
for f in bench_vgg16 bench_resnet50 bench_vgg16_5x bench_r3; do python - $OUT/$f.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print(sys.argv[1].split('/')[-1], d["value"], d.get("job_ms"), d.get("mask_parity_vs_reference_golden"), r.get("frac"), r.get("peak_measured"),
          (d.get("cpu_baseline") or {}).get("job_speedup_wall_clock"), (d.get("two_jobs_in_flight") or {}).get("value"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
