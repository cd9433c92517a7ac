#!/bin/bash
# Round-2 profile set (GPU box): bash tools/profile_round2.sh gpurun_out/prof2
#   bench.json                      the default bench line (vgg16 job)
#   kernels_vgg16_job.md            rocprofv3 --kernel-trace of whole jobs only (per-job averages)
#   pmc_fetch_size_kb.md / pmc_write_size_kb.md   separate --pmc passes (MI355X_MICROARCH.md: HBM section)
#   kernels_block_single.md         the conv3_x block, one instance at a time
set -u
OUT=$GRAFT_REPO_ROOT/${1:-gpurun_out/prof2}
R=$GRAFT_REPO_ROOT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
rm -rf /tmp/kt1 /tmp/ktb /tmp/pf /tmp/pw
# 1 + 1 + 2 x 4 = 10 jobs
rocprofv3 --kernel-trace -d /tmp/kt1 -o r -- python $R/bench.py --profile-mode --steps 2 --warmup 1 --jobs-per-step 4 > $OUT/bench_under_rocprof_vgg16.json 2> $OUT/kt1.err
python $R/tools/rocpd_kernels.py $(find /tmp/kt1 -name '*.db' | head -1) 10 > $OUT/kernels_vgg16_job.md
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf -o r -- python $R/bench.py --profile-mode --steps 1 --warmup 1 --jobs-per-step 1 > /dev/null 2> $OUT/pf.err
python $R/tools/rocpd_pmc.py $(find /tmp/pf -name '*.db' | head -1) FETCH_SIZE > $OUT/pmc_fetch_size_kb.md
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pw -o r -- python $R/bench.py --profile-mode --steps 1 --warmup 1 --jobs-per-step 1 > /dev/null 2> $OUT/pw.err
python $R/tools/rocpd_pmc.py $(find /tmp/pw -name '*.db' | head -1) WRITE_SIZE > $OUT/pmc_write_size_kb.md
rocprofv3 --kernel-trace -d /tmp/ktb -o r -- python $R/bench.py --workload block --inflight 1 --batch 1 --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench_under_rocprof_block_inflight1.json 2> $OUT/ktb.err
python $R/tools/rocpd_kernels.py $(find /tmp/ktb -name '*.db' | head -1) 1 > $OUT/kernels_block_inflight1_raw.md
ls -la $OUT
