#!/bin/bash
# round 6, call 16: the default line with the A/B of the factorisation's two forms inside
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_call16
mkdir -p $OUT
cd $R
timeout 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --detail $OUT/bench.detail.json > $OUT/bench.json 2> $OUT/bench.err
echo "rc=$? bytes=$(wc -c < $OUT/bench.json)"; tail -3 $OUT/bench.err
python3 -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['job_ms'], d.get('chol_form_ab_job_ms'), d.get('two_jobs_in_flight_layers_per_s'), d.get('other_workloads'), d.get('r3'))
print(json.load(open('$OUT/bench.detail.json'))['chol_form_ab'])"
