"""What slows the single-workgroup alpha search when the rest of the chip is busy: the search of one c = 512 layer alone,
next to a register-only f64 MFMA loop on every CU (no memory traffic: clocks / issue slots), and next to an HBM copy loop
(no MFMA: L2 / fabric).  python tools/cd_contention.py"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd"))
import numpy as np
import bench
from cpmi355 import capi, LayerProblem, prune_layer

ctx, other = capi.Context(0), capi.Context(0)
spec = [s for s in bench.vgg16_specs() if s["c"] == 512][0]
X, W2, Y = bench.synth(spec["layer_id"], spec["c"], spec["n"])[:3]
prob = LayerProblem(ctx, X, W2, Y)
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
    res = []
    for rep in range(3):
        prune_layer(prob, spec["rank"], 1e-3, rng=np.random.RandomState(1234 + spec["layer_id"]), mode="device", latency_mode=False)
        st = dict(ctx.last_stage_times())
        steps = sum(f[2] for f in prob.fits) * spec["c"]
        res.append((round(st["cd_alpha_search"], 2), round(st["cd_alpha_search"] * 1e3 / steps, 4)))
    stop = True
    if th: th.join()
    print(kind, "search ms, us/step:", res)
