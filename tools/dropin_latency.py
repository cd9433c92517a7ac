"""Per-call cost of the drop-in flow for one layer, from HOST arrays as dictionary() receives them:
  resident : LayerProblem(X, W2, Y) uploaded first, then prune_layer          (upload / prune / free)
  streamed : LayerProblem(..., defer_upload=True) + prune_layer               (what lib.decompose.dictionary() does: only
             the sampled rows go up before the alpha search, X and Y stream in behind it -- cp_prune_layer_h2d)
for X as float32 and as the float64 array the reference hands over.  python tools/dropin_latency.py [c ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd"))
import numpy as np
import bench
from cpmi355 import capi, LayerProblem, prune_layer

ctx = capi.Context(0)
widths = [int(a) for a in sys.argv[1:]] or [512, 256]
specs = bench.cpjobs.JOBS["vgg16"]()
for c in widths:
    spec = [s for s in specs if s["c"] == c][-1]
    X32, W2, Y = bench.cpjobs.synth(spec)[:3]
    for xname, X in (("float32", X32), ("float64", X32.astype(np.float64))):
        ref = None
        for mode in ("resident", "streamed"):
            rows = []
            for rep in range(int(os.environ.get("REPS", "6"))):
                t0 = time.perf_counter()
                pr = LayerProblem(ctx, X, W2, Y, defer_upload=(mode == "streamed"))
                ctx.sync()
                t1 = time.perf_counter()
                out = prune_layer(pr, spec["rank"], 1e-3, rng=np.random.RandomState(1234 + spec["layer_id"]), mode="device")
                t2 = time.perf_counter()
                pr.free()
                t3 = time.perf_counter()
                rows.append((t1 - t0, t2 - t1, t3 - t2, t3 - t0))
                if ref is None:
                    ref = out
                else:
                    assert np.array_equal(ref[0], out[0]) and np.array_equal(ref[1], out[1]) and np.array_equal(ref[2], out[2]), \
                        "streamed upload changed the result"
            best = min(rows, key=lambda r: r[3])
            med = sorted(r[3] for r in rows)[len(rows) // 2]
            print("c=%d X %s %-8s: setup+upload %.2f  prune %.2f  free %.2f  total %.2f ms (median total %.2f)" % (
                c, xname, mode, best[0] * 1e3, best[1] * 1e3, best[2] * 1e3, best[3] * 1e3, med * 1e3))
