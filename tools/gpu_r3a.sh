#!/bin/bash
# round 3, first GPU pass: whole GPU suite + the three job workloads
OUT=gpurun_out/r3a
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
timeout 600 python bench.py > $OUT/bench_vgg16.json 2> $OUT/bench_vgg16.err
echo "vgg16 rc=$?"
timeout 400 python bench.py --workload resnet50 --steps 5 --warmup 2 --no-cpu-baseline --no-gather > $OUT/bench_resnet50.json 2> $OUT/bench_resnet50.err
echo "resnet50 rc=$?"
timeout 400 python bench.py --workload vgg16_5x --steps 5 --warmup 2 --no-cpu-baseline --no-gather > $OUT/bench_vgg16_5x.json 2> $OUT/bench_vgg16_5x.err
echo "vgg16_5x rc=$?"
tail -5 $OUT/pytest.log
