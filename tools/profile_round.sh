#!/bin/bash
# round profile set: headline bench line, kernel traces (one pass at a time / 6 in flight), PMC passes.
# usage (GPU box): bash tools/profile_round.sh gpurun_out/prof
set -u
OUT=$GRAFT_REPO_ROOT/${1:-gpurun_out/prof}
R=$GRAFT_REPO_ROOT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
rm -rf /tmp/kt1 /tmp/kt6 /tmp/pf /tmp/pw
rocprofv3 --kernel-trace -d /tmp/kt1 -o r -- python $R/bench.py --steps 24 --warmup 6 --no-cpu-baseline --inflight 1 --batch 1 > $OUT/bench_rocprof_inflight1.json 2> $OUT/kt1.err
python $R/tools/rocpd_kernels.py $(find /tmp/kt1 -name '*.db' | head -1) 33 > $OUT/kernels_inflight1.md
rocprofv3 --kernel-trace -d /tmp/kt6 -o r -- python $R/bench.py --steps 96 --warmup 48 --no-cpu-baseline --inflight 6 --batch 8 > $OUT/bench_rocprof_inflight6.json 2> $OUT/kt6.err
python $R/tools/rocpd_kernels.py $(find /tmp/kt6 -name "*.db" | head -1) 147 > $OUT/kernels_inflight6.md
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf -o r -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --inflight 1 --batch 1 > /dev/null 2> $OUT/pf.err
python $R/tools/rocpd_pmc.py $(find /tmp/pf -name '*.db' | head -1) FETCH_SIZE > $OUT/pmc_fetch_size_kb.md
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pw -o r -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --inflight 1 --batch 1 > /dev/null 2> $OUT/pw.err
python $R/tools/rocpd_pmc.py $(find /tmp/pw -name '*.db' | head -1) WRITE_SIZE > $OUT/pmc_write_size_kb.md
ls -la $OUT
