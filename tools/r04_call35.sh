#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${CALL_NAME:-r04_call35}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 120 $R/tools/ubench/chol_bulk > $OUT/chol_bulk.md 2>&1
cat $OUT/chol_bulk.md
