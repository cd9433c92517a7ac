#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3g
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_net_gpu.py -m gpu -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -12 $OUT/pytest.log
