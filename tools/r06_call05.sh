#!/bin/bash
# round 6, call 5: timeline of the vgg16 job with the persistent factorisation (per-stream digest), kernel table
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_call05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
timeout -k 5 300 rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py --profile-mode --steps 2 --warmup 1 --jobs-per-step 4 --detail '' > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
DB=$(find /tmp/kt -name '*.db' | head -1)
[ -n "$DB" ] && python $R/tools/rocpd_kernels.py $DB 10 > $OUT/kernels_vgg16.md
[ -n "$DB" ] && python $R/tools/rocpd_timeline.py $DB --anchor=k_lasso_prep:12 --streams=14 > $OUT/timeline_vgg16.md 2>&1
head -40 $OUT/kernels_vgg16.md
