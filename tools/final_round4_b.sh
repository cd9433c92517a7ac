#!/bin/bash
# Round-4 final check (GPU box), part B: the profile set of the final library (tools/profile_round4.sh), the gather's
# FETCH_SIZE pass, the matrix-pipe probes, the two-rank flow on one GPU.   CP_COMMIT = the commit of the library.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final4
mkdir -p $OUT
bash $R/tools/profile_round4.sh > $OUT/profile_round4.log 2>&1; tail -3 $OUT/profile_round4.log
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pg_$C
  timeout -k 5 200 rocprofv3 --pmc $C --kernel-trace -d /tmp/pg_$C -o r -- python -c "import sys; sys.path.insert(0, '$R'); import bench; print(bench.bench_patch_gather(0, reps=5))" > $OUT/pg_$C.log 2> $OUT/pg_$C.err
  DB=$(find /tmp/pg_$C -name '*.db' | head -1)
  [ -n "$DB" ] && python $R/tools/rocpd_pmc.py $DB $C k_patch_gather > $OUT/pmc_gather_$(echo $C | tr A-Z a-z)_kb.md
done
cat $OUT/pmc_gather_fetch_size_kb.md
timeout -k 5 120 $R/tools/ubench/mfma_clock > $OUT/mfma_clock.md 2>&1; tail -22 $OUT/mfma_clock.md
cd $R
CP_BENCH_DIST_BACKEND=gloo timeout -k 5 300 python bench.py --gpus 2 --steps 3 --warmup 1 --no-gather < /dev/null > $OUT/bench_2ranks_gloo.json 2> $OUT/bench_2ranks_gloo.err; echo "2 ranks rc=$?"
python - $OUT/bench_2ranks_gloo.json <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("2 ranks (gloo, one GPU):", d["value"], d.get("job_ms"), d.get("mask_parity_vs_reference_golden"), (d.get("replica_throughput") or {}).get("value"))
except Exception as e:
    print("unreadable", e)
PY
