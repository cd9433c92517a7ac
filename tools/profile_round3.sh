#!/bin/bash
# Round-3 profile set (GPU box): bash tools/profile_round3.sh [gpurun_out/prof3]
#   bench_vgg16.json / bench_resnet50.json / bench_vgg16_5x.json      the three job workloads (default flags)
#   bench_vgg16_cpu_full.json     vgg16 with the CPU port timed on all 12 layers (job_speedup_wall_clock)
#   kernels_<job>.md              rocprofv3 --kernel-trace of whole jobs only (per-job averages)
#   pmc_fetch_size_kb.md / pmc_write_size_kb.md   separate --pmc passes over the vgg16 job (MI355X_MICROARCH.md: HBM section)
#   pmc_mfma_*.md                 SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE of the GEMM kernels (vgg16 job)
set -u
OUT=$GRAFT_REPO_ROOT/${1:-gpurun_out/prof3}
R=$GRAFT_REPO_ROOT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py > $OUT/bench_vgg16.json 2> $OUT/bench_vgg16.err
timeout 400 python $R/bench.py --workload resnet50 > $OUT/bench_resnet50.json 2> $OUT/bench_resnet50.err
timeout 400 python $R/bench.py --workload vgg16_5x > $OUT/bench_vgg16_5x.json 2> $OUT/bench_vgg16_5x.err
timeout 600 python $R/bench.py --cpu-full --no-gather --no-block --no-pcie-f64 --steps 5 > $OUT/bench_vgg16_cpu_full.json 2> $OUT/bench_vgg16_cpu_full.err
for W in vgg16 resnet50 vgg16_5x; do
  rm -rf /tmp/kt_$W
  timeout 300 rocprofv3 --kernel-trace -d /tmp/kt_$W -o r -- python $R/bench.py --workload $W --profile-mode --steps 2 --warmup 1 --jobs-per-step 4 > $OUT/bench_under_rocprof_$W.json 2> $OUT/kt_$W.err
  python $R/tools/rocpd_kernels.py $(find /tmp/kt_$W -name '*.db' | head -1) 10 > $OUT/kernels_$W.md
done
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/p_$C
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d /tmp/p_$C -o r -- python $R/bench.py --profile-mode --steps 1 --warmup 1 --jobs-per-step 1 > /dev/null 2> $OUT/p_$C.err
  python $R/tools/rocpd_pmc.py $(find /tmp/p_$C -name '*.db' | head -1) $C > $OUT/pmc_$(echo $C | tr A-Z a-z)_kb.md
done
for C in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE; do
  rm -rf /tmp/pm_$C
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d /tmp/pm_$C -o r -- python $R/bench.py --profile-mode --steps 1 --warmup 1 --jobs-per-step 1 > /dev/null 2> $OUT/pm_$C.err
  DB=$(find /tmp/pm_$C -name '*.db' | head -1)
  [ -n "$DB" ] && python $R/tools/rocpd_pmc.py $DB $C k_gemm > $OUT/pmc_mfma_$C.md 2>&1
done
ls -la $OUT
