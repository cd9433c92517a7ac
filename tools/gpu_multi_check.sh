#!/bin/bash
# GPU box: suite + smoke + the two benches the multi-CU CD team touches, and the kernel table of the ResNet-50 job.
# Every step bounded; nothing reads stdin.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/multi
mkdir -p $OUT
cd $R
timeout -k 5 400 python -m pytest tests -m gpu -q -x < /dev/null > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
timeout -k 5 120 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 python $R/bench.py --workload resnet50 < /dev/null > $OUT/bench_resnet50.json 2> $OUT/bench_resnet50.err; echo "resnet rc=$?"
timeout -k 5 300 python $R/bench.py < /dev/null > $OUT/bench_vgg16.json 2> $OUT/bench_vgg16.err; echo "vgg rc=$?"
rm -rf /tmp/kt_rn
timeout -k 5 200 rocprofv3 --kernel-trace -d /tmp/kt_rn -o r -- python $R/bench.py --workload resnet50 --profile-mode --steps 2 --warmup 1 --jobs-per-step 4 < /dev/null > $OUT/bench_under_rocprof_resnet50.json 2> $OUT/kt_rn.err; echo "rocprof rc=$?"
DB=$(find /tmp/kt_rn -name '*.db' 2>/dev/null | head -1)
if [ -n "$DB" ]; then timeout -k 5 120 python $R/tools/rocpd_kernels.py $DB 10 < /dev/null > $OUT/kernels_resnet50.md 2>&1; fi
for f in bench_resnet50 bench_vgg16 bench_under_rocprof_resnet50; do python - $OUT/$f.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d["value"], d.get("job_ms"), d.get("mask_parity_vs_reference_golden"), d["roofline"]["frac"])
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
