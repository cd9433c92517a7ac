#!/bin/bash
# Round-6 profile set (GPU box): bash tools/profile_round6.sh   (CP_COMMIT = the commit of the library, set by the caller)
#   kernels_<job>.md            rocprofv3 --kernel-trace of whole jobs only (per-job averages)
#   timeline_vgg16.md           per-stream digest of the last job of that trace (tools/rocpd_timeline.py, every stream)
#   pmc_fetch_size_kb.md / pmc_write_size_kb.md   separate --pmc passes over the vgg16 job (MI355X_MICROARCH.md: HBM section)
#   pmc_mfma_*                  SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES of the MFMA kernels (vgg16 job)
#   chol_chain.md               the persistent factorisation against the launch-per-step form (bit for bit), alone and five side by side
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof6
R=$GRAFT_REPO_ROOT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for W in vgg16 resnet50 vgg16_5x; do
  rm -rf /tmp/kt_$W
  timeout -k 5 300 rocprofv3 --kernel-trace -d /tmp/kt_$W -o r -- python $R/bench.py --workload $W --profile-mode --steps 2 --warmup 1 --jobs-per-step 4 > $OUT/bench_under_rocprof_$W.json 2> $OUT/kt_$W.err
  DB=$(find /tmp/kt_$W -name '*.db' | head -1)
  [ -n "$DB" ] && python $R/tools/rocpd_kernels.py $DB 10 > $OUT/kernels_$W.md
  if [ "$W" = vgg16 ] && [ -n "$DB" ]; then python $R/tools/rocpd_timeline.py $DB --anchor=k_lasso_prep:12 --streams=14 > $OUT/timeline_vgg16.md 2>&1; fi
done
if [ "${CP_PROFILE_PMC:-1}" = 1 ]; then
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_$C
  timeout -k 5 300 rocprofv3 --pmc $C --kernel-trace -d /tmp/p_$C -o r -- python $R/bench.py --profile-mode --steps 1 --warmup 1 --jobs-per-step 1 > /dev/null 2> $OUT/p_$C.err
  DB=$(find /tmp/p_$C -name '*.db' | head -1)
  [ -n "$DB" ] && python $R/tools/rocpd_pmc.py $DB $C > $OUT/pmc_$(echo $C | tr A-Z a-z)_kb.md
done
for C in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES; do
  rm -rf /tmp/pm_$C
  timeout -k 5 300 rocprofv3 --pmc $C --kernel-trace -d /tmp/pm_$C -o r -- python $R/bench.py --profile-mode --steps 1 --warmup 1 --jobs-per-step 1 > /dev/null 2> $OUT/pm_$C.err
  DB=$(find /tmp/pm_$C -name '*.db' | head -1)
  [ -n "$DB" ] && python $R/tools/rocpd_pmc.py $DB $C k_ > $OUT/pmc_mfma_$C.md 2>&1
done
fi
timeout -k 5 300 $R/tools/ubench/chol_chain > $OUT/chol_chain.md 2>&1
ls -la $OUT
