#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${CALL_NAME:-r04_call27}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 200 $R/tools/ubench/gemm_hybrid > $OUT/gemm_hybrid.md 2>&1
cat $OUT/gemm_hybrid.md
