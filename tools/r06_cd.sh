#!/bin/bash
# round 6: CD team kernel variants -- bit-exactness tests, then cd_bench per variant library (arguments: library names under build_variants/)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_cd
mkdir -p $OUT
cd $R
for LIB in "$@"; do
  echo "== $LIB"
  CP_LIB_PATH=$R/build_variants/$LIB.so timeout -k 5 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "cd_ or tie or multi_cu or dictionary_matches_reference_golden" > $OUT/pytest_$LIB.log 2>&1
  tail -2 $OUT/pytest_$LIB.log
  CD_BENCH_FLAGS=0 timeout -k 5 300 python tools/cd_bench.py $R/build_variants/$LIB.so > $OUT/cd_bench_$LIB.txt 2>&1
  grep "ns/step" $OUT/cd_bench_$LIB.txt | cut -c1-150
done
