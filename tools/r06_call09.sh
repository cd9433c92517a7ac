#!/bin/bash
# round 6, call 9: is the 10 ms of a 64-channel layer "alone" inside bench.py reproducible, and which leg switches it on
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06_call09
mkdir -p $OUT
cd $R
show() {
python3 - "$1" "$2" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
pl = d["per_layer_rank0"]
st = d["stage_ms_alone_by_shape_rank0"]["c64_k3_n64"]
print(sys.argv[2], d["job_ms"], {k[:3]: v["ms_alone"] for k, v in pl.items() if k[:3] in ("V01", "V02", "V03", "V04")}, st.get("prefactor_cholesky"), st.get("refit_wait_prefactor"))
PY
}
B="python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-gather --no-extras --no-pcie-f64"
timeout 300 $B --detail $OUT/a.json > /dev/null 2>$OUT/a.err; show $OUT/a.json "block+pipelined"
timeout 300 $B --no-block --detail $OUT/b.json > /dev/null 2>$OUT/b.err; show $OUT/b.json "pipelined only"
timeout 300 $B --no-pipelined --detail $OUT/c.json > /dev/null 2>$OUT/c.err; show $OUT/c.json "block only"
timeout 300 $B --no-pipelined --no-block --detail $OUT/d.json > /dev/null 2>$OUT/d.err; show $OUT/d.json "neither"
CP_LIB_PATH=$R/build_variants/lib_before_chain.so timeout 300 $B --detail $OUT/e.json > /dev/null 2>$OUT/e.err; show $OUT/e.json "old lib, block+pipelined"
