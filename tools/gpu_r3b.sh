#!/bin/bash
# round 3, second GPU pass: team CD kernels -- parity suite, step cost by channel count, the jobs
OUT=gpurun_out/r3b
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -4 $OUT/pytest.log
for team in 1 0; do
  CP_CD_TEAM=$team timeout 300 python tools/cd_bench.py > $OUT/cd_bench_team$team.log 2>&1
done
CP_CD_EXACT_DIV=1 CD_BENCH_FLAGS=0 timeout 300 python tools/cd_bench.py > $OUT/cd_bench_team1_exactdiv.log 2>&1
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-gather --no-pcie-f64 > $OUT/bench_vgg16.json 2> $OUT/bench_vgg16.err
echo "vgg16 rc=$?"
timeout 400 python bench.py --workload resnet50 --steps 5 --warmup 2 --no-cpu-baseline --no-gather --no-pcie-f64 > $OUT/bench_resnet50.json 2> $OUT/bench_resnet50.err
echo "resnet50 rc=$?"
timeout 400 python bench.py --workload vgg16_5x --steps 5 --warmup 2 --no-cpu-baseline --no-gather --no-pcie-f64 > $OUT/bench_vgg16_5x.json 2> $OUT/bench_vgg16_5x.err
echo "vgg16_5x rc=$?"
cat $OUT/cd_bench_team1.log
