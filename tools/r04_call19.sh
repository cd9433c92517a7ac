#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do CP_JACOBI_INNER=$i timeout -k 5 200 python tests/tools/svd_inner_probe.py 2>&1 | tail -2; done
