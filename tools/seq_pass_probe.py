"""Where the time of a layer-after-layer drop-in pass goes: per layer, LayerProblem set-up / prune / free, for three
passes over the 12 vgg16 layers from distinct pageable host arrays (the bench's pcie_inclusive pass)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd"))
import numpy as np
import bench
from cpmi355 import capi, LayerProblem, prune_layer

ctx = capi.Context(0)
specs = bench.cpjobs.JOBS["vgg16"]()
data = [bench.cpjobs.synth(s)[:3] for s in specs]
for defer in (True, False):
    for rep in range(3):
        rows = []
        t_pass = time.perf_counter()
        for spec, (X, W2, Y) in zip(specs, data):
            t0 = time.perf_counter()
            pr = LayerProblem(ctx, X, W2, Y, defer_upload=defer)
            t1 = time.perf_counter()
            prune_layer(pr, spec["rank"], 1e-3, rng=np.random.RandomState(1234 + spec["layer_id"]), mode="device")
            t2 = time.perf_counter()
            pr.free()
            t3 = time.perf_counter()
            rows.append("%s %.1f/%.1f/%.1f" % (spec["name"][:3], (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
        print("defer=%s pass %d: %.1f ms | setup/prune/free per layer: %s" % (defer, rep, (time.perf_counter() - t_pass) * 1e3, " ".join(rows)))
