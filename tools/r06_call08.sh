#!/bin/bash
# round 6, call 8: why a 64-channel layer alone takes 10 ms since the persistent-factorisation commit (prefactor route)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for LIB in lib_before_chain lib_cur lib_diag1024; do
  for FORM in steps chain; do
    echo "== $LIB $FORM"
    CP_LIB_PATH=$R/build_variants/$LIB.so CP_CHOL_FORM=$FORM timeout -k 5 120 python tools/probes/small_layer_latency.py V01 V03 V08 2>&1 | tail -4
  done
done
