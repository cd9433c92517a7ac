#!/bin/bash
# kernel trace of ONE c = 512 layer alone: what a factorisation step costs without the other layers
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${CALL_NAME:-r04_call31}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt1
timeout -k 5 200 rocprofv3 --kernel-trace -d /tmp/kt1 -o r -- python $R/tools/one_layer_trace.py 512 > $OUT/one_layer.log 2> $OUT/one_layer.err; echo "rc $?"; tail -2 $OUT/one_layer.log
DB=$(find /tmp/kt1 -name '*.db' | head -1)
python $R/tools/rocpd_timeline.py $DB --anchor=k_lasso_prep:1 --streams=3 > $OUT/timeline_one_layer.md 2>&1
python $R/tools/rocpd_kernels.py $DB 3 > $OUT/kernels_one_layer.md 2>&1
head -40 $OUT/kernels_one_layer.md | cut -c1-200
grep -n "k_chol_step" $OUT/timeline_one_layer.md | head -50 | cut -c1-160
