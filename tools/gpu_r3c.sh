#!/bin/bash
# round 3, third GPU pass: super-tile GEMM order, coalesced q pass, colsum ILP; CD placement options in the vgg16 job
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3c
R=$GRAFT_REPO_ROOT
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
Q="--steps 5 --warmup 2 --no-cpu-baseline --no-gather --no-pcie-f64 --no-block"
timeout 300 python bench.py $Q > $OUT/b_default.json 2> $OUT/b_default.err
CP_CD_PRIO=1 timeout 300 python bench.py $Q > $OUT/b_prio.json 2> $OUT/b_prio.err
CP_CD_SPREAD=1 timeout 300 python bench.py $Q > $OUT/b_spread.json 2> $OUT/b_spread.err
CP_CD_SPREAD=1 CP_CD_PRIO=1 timeout 300 python bench.py $Q > $OUT/b_spread_prio.json 2> $OUT/b_spread_prio.err
CP_CD_EXCLUSIVE=1 timeout 300 python bench.py $Q > $OUT/b_excl.json 2> $OUT/b_excl.err
CP_CD_EXCLUSIVE=1 CP_CD_PRIO=1 timeout 300 python bench.py $Q > $OUT/b_excl_prio.json 2> $OUT/b_excl_prio.err
CP_CD_TEAM=0 timeout 300 python bench.py $Q > $OUT/b_noteam.json 2> $OUT/b_noteam.err
for f in default prio spread spread_prio excl excl_prio noteam; do python - <<PY
import json
try:
    d=json.load(open("$OUT/b_$f.json"))
    print("$f", d["job_ms"], d["value"], d["mask_parity_vs_reference_golden"], d["roofline"]["frac"], d["roofline"]["avg_launch_ms"])
except Exception as e:
    print("$f ERR", e)
PY
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pf /tmp/kt1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pf -o r -- python $R/bench.py --profile-mode --steps 1 --warmup 1 --jobs-per-step 1 > /dev/null 2> $OUT/pf.err
python $R/tools/rocpd_pmc.py $(find /tmp/pf -name '*.db' | head -1) FETCH_SIZE > $OUT/pmc_fetch_size_kb.md
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt1 -o r -- python $R/bench.py --profile-mode --steps 2 --warmup 1 --jobs-per-step 4 > $OUT/bench_under_rocprof_vgg16.json 2> $OUT/kt1.err
python $R/tools/rocpd_kernels.py $(find /tmp/kt1 -name '*.db' | head -1) 10 > $OUT/kernels_vgg16_job.md
head -30 $OUT/pmc_fetch_size_kb.md
