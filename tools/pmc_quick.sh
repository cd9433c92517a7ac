set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r2l; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --profile-mode --steps 3 --warmup 2 > $OUT/bench.json 2>/dev/null
python -c "
import json; d=json.load(open('$OUT/bench.json')); print('job_ms',d['job_ms'],'roof in-job',d['roofline']['frac'],d['roofline']['avg_launch_ms'])"
rm -rf /tmp/pf /tmp/kt
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf -o r -- python $R/bench.py --profile-mode --steps 1 --warmup 1 --jobs-per-step 1 > /dev/null 2> $OUT/pf.err
python $R/tools/rocpd_pmc.py $(find /tmp/pf -name '*.db' | head -1) FETCH_SIZE | head -6
rocprofv3 --kernel-trace -d /tmp/kt -o r -- python $R/bench.py --workload block --inflight 1 --batch 1 --steps 2 --warmup 2 --no-cpu-baseline > /dev/null 2> $OUT/kt.err
python $R/tools/rocpd_kernels.py $(find /tmp/kt -name '*.db' | head -1) 1 | grep "gemm_tn_f64<1, 2"
