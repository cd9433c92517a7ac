#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_call16
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for V in base full5; do
  rm -rf /tmp/kt_$V
  E="CP_NOP=1"; [ $V = full5 ] && E="CP_JOB_LATENCY_KIND=full CP_JOB_PRECOMPUTE=5"
  env $E timeout -k 5 300 rocprofv3 --kernel-trace -d /tmp/kt_$V -o r -- python $R/bench.py --no-cpu-baseline --no-gather --no-block --no-pcie-f64 --profile-mode --steps 2 --warmup 1 --jobs-per-step 4 > $OUT/bench_$V.json 2> $OUT/kt_$V.err
  DB=$(find /tmp/kt_$V -name '*.db' | head -1)
  [ -n "$DB" ] && python $R/tools/rocpd_timeline.py $DB --anchor=k_lasso_prep:12 --streams=13 > $OUT/timeline_$V.md 2>&1
done
ls -la $OUT
