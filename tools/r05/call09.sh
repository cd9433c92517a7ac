#!/bin/bash
# Round 5, GPU call 9: host side of a job: when each layer's thread gets going after start() and when it ends, against job_ms.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r05_call09
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 python $R/bench.py --steps 3 --warmup 2 --jobs-per-step 8 --no-cpu-baseline --no-block --no-gather --no-pcie-f64 --no-pipelined < /dev/null > $OUT/bench.json 2> $OUT/bench.err
python - $OUT/bench.json <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("job_ms", d["job_ms"])
for c in d["chunks_rank0_last_job"]:
    print(c)
PY
