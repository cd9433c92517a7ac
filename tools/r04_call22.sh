#!/bin/bash
# what bounds the f64 GEMM: accumulator register file, LDS-read pattern, tile shapes and tile counts
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04_call22
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 200 $R/tools/ubench/gemm_probe > $OUT/gemm_probe.md 2>&1
cat $OUT/gemm_probe.md
