#!/bin/bash
# MFMA utilisation of the GEMM kernels: separate --pmc passes (one counter each) over a one-pass-at-a-time bench run
set -u
OUT=$GRAFT_REPO_ROOT/${1:-gpurun_out/pmc_mfma}
R=$GRAFT_REPO_ROOT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAVE_CYCLES; do
  rm -rf /tmp/pm_$C
  timeout 200 rocprofv3 --pmc $C --kernel-trace -d /tmp/pm_$C -o r -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --inflight 1 > /dev/null 2> $OUT/$C.err
  DB=$(find /tmp/pm_$C -name '*.db' | head -1)
  [ -n "$DB" ] && python $R/tools/rocpd_pmc.py $DB $C k_gemm > $OUT/$C.md 2>&1
done
ls $OUT
