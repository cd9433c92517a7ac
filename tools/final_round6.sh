#!/bin/bash
# Round-6 final check (GPU box): suite, smoke, the driver's bench command and the other bench lines, two ranks on one GPU in
# every exchange mode, the profile set (CP_PROFILE_SET=0 leaves it out).  Every step bounded.  Outputs: gpurun_out/final6/.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final6
mkdir -p $OUT
cd $R
timeout -k 5 900 python -m pytest tests -m gpu -q -x < /dev/null > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
timeout -k 5 120 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
T0=$(date +%s)
timeout -k 5 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail $OUT/bench_vgg16.detail.json < /dev/null > $OUT/bench_vgg16.json 2> $OUT/bench_vgg16.err; echo "vgg rc=$? wall=$(( $(date +%s) - T0 )) s bytes=$(wc -c < $OUT/bench_vgg16.json)"
timeout -k 5 300 python bench.py --workload resnet50 --no-cpu-baseline --no-gather --detail $OUT/bench_resnet50.detail.json < /dev/null > $OUT/bench_resnet50.json 2> $OUT/bench_resnet50.err; echo "resnet rc=$?"
timeout -k 5 300 python bench.py --workload vgg16_5x --no-cpu-baseline --no-gather --detail $OUT/bench_vgg16_5x.detail.json < /dev/null > $OUT/bench_vgg16_5x.json 2> $OUT/bench_vgg16_5x.err; echo "5x rc=$?"
timeout -k 5 400 python bench.py --workload r3 --steps 2 --warmup 1 --detail $OUT/bench_r3.detail.json < /dev/null > $OUT/bench_r3.json 2> $OUT/bench_r3.err; echo "r3 rc=$?"
timeout -k 5 300 python bench.py --sequential-alpha --steps 3 --warmup 1 --no-cpu-baseline --detail $OUT/bench_seq.detail.json < /dev/null > $OUT/bench_vgg16_sequential_alpha.json 2> $OUT/bench_seq.err; echo "seq rc=$?"
timeout -k 5 300 python bench.py --workload block --no-cpu-baseline --detail $OUT/bench_block.detail.json < /dev/null > $OUT/bench_block.json 2> $OUT/bench_block.err; echo "block rc=$?"
for M in gather allgather masks; do
  CP_BENCH_DIST_BACKEND=gloo timeout -k 5 300 python bench.py --gpus 2 --steps 3 --warmup 1 --no-gather --exchange $M --detail $OUT/bench_2ranks_gloo_$M.detail.json < /dev/null > $OUT/bench_2ranks_gloo_$M.json 2> $OUT/bench_2ranks_gloo_$M.err; echo "2 ranks $M rc=$?"
done
CP_BENCH_DIST_BACKEND=gloo CP_BENCH_ASSISTS=7:1 timeout -k 5 300 python bench.py --gpus 2 --steps 2 --warmup 1 --no-gather --detail $OUT/bench_2ranks_gloo_assist.detail.json < /dev/null > $OUT/bench_2ranks_gloo_assist.json 2> $OUT/bench_2ranks_gloo_assist.err; echo "2 ranks forced assist rc=$?"
for f in $OUT/bench_*.json; do case $f in *.detail.json) continue;; esac; python - $f <<'PY'
import json, sys
try:
    text = open(sys.argv[1]).read().strip().splitlines()[-1]
    d = json.loads(text)
    r = d.get("roofline") or {}
    print(sys.argv[1].split('/')[-1], len(text), "B", d["value"], d.get("job_ms"), d.get("mask_parity_vs_reference_golden"), r.get("frac"), r.get("sum_ms_per_job"),
          (d.get("cpu_baseline") or {}).get("job_speedup_wall_clock"), d.get("two_jobs_in_flight_layers_per_s"),
          (d.get("value_conv3_block") or {}).get("value"), d.get("other_workloads"), d.get("r3"), d.get("exchange"),
          d.get("masks_and_alpha_chain_identical_to_the_reference_chain"))
except Exception as e:
    print(sys.argv[1], "unreadable", e)
PY
done
if [ "${CP_PROFILE_SET:-1}" = 1 ]; then bash $R/tools/profile_round6.sh > $OUT/profile_round6.log 2>&1; tail -3 $OUT/profile_round6.log; fi
