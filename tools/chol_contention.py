"""Is the Cholesky chain slow in a busy chip because its 158 KB-LDS workgroup waits for a free CU, or because it is a
latency-bound kernel like the alpha search?  The refit of one c = 256 layer (p ~ 2000) alone, next to a register-only MFMA
loop (occupies issue slots / clocks, no LDS, no memory) and next to an HBM copy loop (no LDS, no MFMA).
python tools/chol_contention.py"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd"))
import numpy as np
import bench
from cpmi355 import capi, LayerProblem

ctx, other = capi.Context(0), capi.Context(0)
spec = [s for s in bench.vgg16_specs() if s["c"] == 256][0]
X, W2, Y = bench.synth(spec["layer_id"], spec["c"], spec["n"])[:3]
prob = LayerProblem(ctx, X, W2, Y)
mask = np.zeros(spec["c"], dtype=bool); mask[:spec["rank"]] = True
ctx.enable_stage_timing(1)
stop = False

def load(kind):
    while not stop:
        other.probe_mfma_f64() if kind == "mfma" else other.probe_hbm_copy(1 << 30)

for kind in ("idle", "mfma", "copy", "idle"):
    stop = False
    th = None
    if kind != "idle":
        th = threading.Thread(target=load, args=(kind,)); th.start(); time.sleep(0.05)
    rows = []
    for rep in range(3):
        prob.refit(mask)
        st = dict(ctx.last_stage_times())
        rows.append({k: round(st[k], 2) for k in ("refit_gram_gemm", "refit_cholesky", "refit_solve") if k in st})
    stop = True
    if th: th.join()
    print(kind, rows)
