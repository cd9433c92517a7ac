"""Per-call cost of the drop-in flow for one layer: upload (LayerProblem), prune_layer, free -- what dictionary() does.
python tools/dropin_latency.py [c ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd"))
import numpy as np
import bench
from cpmi355 import capi, LayerProblem, prune_layer

ctx = capi.Context(0)
widths = [int(a) for a in sys.argv[1:]] or [512, 256]
for c in widths:
    spec = [s for s in bench.vgg16_specs() if s["c"] == c][0]
    X, W2, Y = bench.synth(spec["layer_id"], spec["c"], spec["n"])[:3]
    rows = []
    ctx.enable_stage_timing(1)
    for rep in range(int(os.environ.get("REPS", "8"))):
        t0 = time.perf_counter()
        pr = LayerProblem(ctx, X, W2, Y)
        ctx.sync()
        t1 = time.perf_counter()
        prune_layer(pr, spec["rank"], 1e-3, rng=np.random.RandomState(1234 + spec["layer_id"]), mode="device")
        t2 = time.perf_counter()
        import ctypes
        h4 = (ctypes.c_double * 4)()
        ctx.lib.cp_debug_host_times.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)]
        ctx.lib.cp_debug_host_times(ctx.h, h4)
        if (t2 - t1) > 0.03:
            print("   slow call: host ms (operands, search, refit, copies+wait) =", [round(v, 2) for v in h4],
                  "device stages:", {k: round(v, 2) for k, v in ctx.last_stage_times() if v > 0.5})
        pr.free()
        t3 = time.perf_counter()
        rows.append(tuple(round((b - a) * 1e3, 2) for a, b in ((t0, t1), (t1, t2), (t2, t3))))
    print("c=%d upload / prune / free ms:" % c, rows)
