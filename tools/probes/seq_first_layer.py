"""Where does the first layer of a sequential drop-in pass spend its time?  (bench: pcie_inclusive.per_layer_ms shows the
64-channel layer that opens a pass at ~8.8 ms against 1.4 ms for its twin that follows.)  Times LayerProblem() / prune / free
per layer over several passes, in the bench's order and reversed."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from benchkit import common as _c   # noqa
from benchkit.common import CD_FLAGS, cpjobs
import cpmi355
from cpmi355.pruner import LayerProblem, prune_layer

specs = cpjobs.JOBS["vgg16"]()
data = {s["layer_id"]: cpjobs.synth(s)[:3] for s in specs}
ctx0 = cpmi355.Context(0)
for order_name, order in (("bench order", specs), ("reversed", specs[::-1]), ("bench order", specs)):
    for p in range(2):
        t_pass = time.perf_counter()
        rows = []
        for spec in order:
            X, W2, Y = data[spec["layer_id"]]
            t0 = time.perf_counter()
            pr = LayerProblem(ctx0, X, W2, Y, flags=CD_FLAGS, defer_upload=True)
            t1 = time.perf_counter()
            prune_layer(pr, spec["rank"], 1e-3, rank_tol=.1, rng=np.random.RandomState(1234 + spec["layer_id"]), mode="device")
            t2 = time.perf_counter()
            pr.free()
            t3 = time.perf_counter()
            rows.append((spec["name"][:3], spec["c"], (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, dict(pr.ctx.last_stage_times()) if False else None))
        print("%s pass %d: %.2f ms | " % (order_name, p, (time.perf_counter() - t_pass) * 1e3) +
              "  ".join("%s c%d %.2f/%.2f/%.2f" % r[:5] for r in rows))
ctx0.close()
