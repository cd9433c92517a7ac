#!/bin/bash
# round 6, call 12: the default line with the extras in a child process; the headline job with nothing before it (profile mode)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_call12
mkdir -p $OUT
cd $R
T0=$(date +%s)
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --detail $OUT/bench.detail.json > $OUT/bench.json 2> $OUT/bench.err
echo "rc=$? wall=$(( $(date +%s) - T0 )) s bytes=$(wc -c < $OUT/bench.json)"; cat $OUT/bench.json; tail -3 $OUT/bench.err
for i in 1 2; do
timeout 300 python3 bench.py --profile-mode --steps 20 --warmup 5 --detail '' > $OUT/pm$i.json 2> $OUT/pm$i.err
python3 -c "
import json; d=json.loads(open('$OUT/pm$i.json').read().strip().splitlines()[-1]); print('profile-mode', d['value'], d['job_ms'])"
done
