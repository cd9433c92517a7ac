#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/check
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?" >> $OUT/pytest.log
tail -8 $OUT/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
bash tools/profile_round3.sh gpurun_out/prof3 > $OUT/profile.log 2>&1
tail -3 $OUT/profile.log
