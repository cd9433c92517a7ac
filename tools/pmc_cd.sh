#!/bin/bash
# per-counter rocprofv3 passes over one single-wave CD fit; results under gpurun_out/pmc_cd/
set -u
OUT=${1:-gpurun_out/pmc_cd}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --list-avail > $R/$OUT/avail.txt 2>&1
for C in SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA; do
  rm -rf /tmp/pc_$C
  timeout 120 rocprofv3 --pmc $C --kernel-trace -d /tmp/pc_$C -o r -- python $R/tools/cd_one.py 256 3 > $R/$OUT/$C.log 2>&1
  DB=$(find /tmp/pc_$C -name '*.db' | head -1)
  if [ -n "$DB" ]; then python $R/tools/rocpd_pmc.py $DB $C k_cd > $R/$OUT/$C.md 2>&1; fi
done
grep -h "k_cd" $R/$OUT/*.md > $R/$OUT/summary.txt
