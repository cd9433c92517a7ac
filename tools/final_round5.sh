#!/bin/bash
# Round-5 final check (GPU box): suite, smoke, the bench lines, the profile set, two ranks on one GPU.  Every step bounded.
# CP_PROFILE_PMC=0: without the four counter passes and the two-rank run (their kernels did not change since the last full run).
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final5
mkdir -p $OUT
cd $R
timeout -k 5 500 python -m pytest tests -m gpu -q -x < /dev/null > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
timeout -k 5 120 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
cd /tmp && export TMPDIR=/tmp
timeout -k 5 400 python $R/bench.py < /dev/null > $OUT/bench_vgg16.json 2> $OUT/bench_vgg16.err; echo "vgg rc=$?"
timeout -k 5 300 python $R/bench.py --workload resnet50 --no-cpu-baseline --no-gather < /dev/null > $OUT/bench_resnet50.json 2> $OUT/bench_resnet50.err; echo "resnet rc=$?"
timeout -k 5 300 python $R/bench.py --workload vgg16_5x --no-cpu-baseline --no-gather < /dev/null > $OUT/bench_vgg16_5x.json 2> $OUT/bench_vgg16_5x.err; echo "5x rc=$?"
timeout -k 5 300 python $R/bench.py --workload r3 --steps 2 --warmup 1 --no-cpu-baseline < /dev/null > $OUT/bench_r3.json 2> $OUT/bench_r3.err; echo "r3 rc=$?"
timeout -k 5 300 python $R/bench.py --sequential-alpha --steps 3 --warmup 1 --no-cpu-baseline < /dev/null > $OUT/bench_vgg16_sequential_alpha.json 2> $OUT/bench_seq.err; echo "seq rc=$?"
for f in bench_vgg16 bench_resnet50 bench_vgg16_5x bench_r3 bench_vgg16_sequential_alpha; do python - $OUT/$f.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print(sys.argv[1].split('/')[-1], d["value"], d.get("job_ms"), d.get("mask_parity_vs_reference_golden"), r.get("frac"), r.get("peak_measured"),
          (d.get("cpu_baseline") or {}).get("job_speedup_wall_clock"), (d.get("two_jobs_in_flight") or {}).get("value"),
          (d.get("value_conv3_block") or {}).get("value"), d.get("masks_and_alpha_chain_identical_to_the_reference_chain"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
bash $R/tools/profile_round5.sh > $OUT/profile_round5.log 2>&1; tail -3 $OUT/profile_round5.log
cd $R
[ "${CP_PROFILE_PMC:-1}" = 1 ] && CP_BENCH_DIST_BACKEND=gloo timeout -k 5 300 python bench.py --gpus 2 --steps 3 --warmup 1 --no-gather < /dev/null > $OUT/bench_2ranks_gloo.json 2> $OUT/bench_2ranks_gloo.err; echo "2 ranks rc=$?"
