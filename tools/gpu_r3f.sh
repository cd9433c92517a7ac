#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3f
R=$GRAFT_REPO_ROOT
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -12 $OUT/pytest.log
