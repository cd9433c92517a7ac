#!/bin/bash
# Round 5, GPU call 6: where the job stands after the band-wise write-back: per-stream timeline + kernel table of the vgg16 job,
# the other two jobs, cd_bench (cycles per coordinate step by width).
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_call06
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
timeout -k 5 240 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py --profile-mode --steps 2 --warmup 1 --jobs-per-step 4 > $OUT/under_rocprof.json 2> $OUT/kt.err
DB=$(find /tmp/kt -name '*.db' | head -1)
[ -n "$DB" ] && python $R/tools/rocpd_timeline.py $DB --anchor=k_lasso_prep:12 --streams=14 > $OUT/timeline_all_streams.md 2>&1
[ -n "$DB" ] && python $R/tools/rocpd_kernels.py $DB 10 > $OUT/kernels_vgg16.md 2>&1
head -30 $OUT/kernels_vgg16.md | cut -c1-200
Q="--steps 3 --warmup 2 --no-cpu-baseline --no-gather --no-pcie-f64"
for W in resnet50 vgg16_5x; do
  timeout -k 5 300 python $R/bench.py --workload $W $Q < /dev/null > $OUT/bench_$W.json 2> $OUT/bench_$W.err
  python - $OUT/bench_$W.json $W <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], "job_ms", d.get("job_ms"), "value", d["value"], "parity", d.get("mask_parity_vs_reference_golden"), "two", (d.get("two_jobs_in_flight") or {}).get("value"))
except Exception as e:
    print(sys.argv[2], "unreadable", e)
PY
done
timeout -k 5 300 python $R/tools/cd_bench.py > $OUT/cd_bench.txt 2>&1; tail -25 $OUT/cd_bench.txt
