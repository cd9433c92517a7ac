#!/bin/bash
# round 6, call 3: the persistent factorisation against the launch-per-step form (bit for bit), alone and five side by side
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_call03
mkdir -p $OUT
cd $R
timeout -k 5 60 tools/ubench/chol_chain quick > $OUT/chol_chain_quick.md 2>&1
echo "quick rc=$?"; tail -5 $OUT/chol_chain_quick.md
timeout -k 5 300 tools/ubench/chol_chain > $OUT/chol_chain.md 2>&1
echo "full rc=$?"; tail -40 $OUT/chol_chain.md
