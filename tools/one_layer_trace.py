"""One c = 512 layer of the vgg16 job, resident operands, pruned three times: the workload of a kernel trace of the layer
ALONE (python tools/one_layer_trace.py under rocprofv3 --kernel-trace; digest with tools/rocpd_timeline.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd"))
import numpy as np
import bench
from cpmi355 import capi, LayerProblem, prune_layer

ctx = capi.Context(0)
c = int(sys.argv[1]) if len(sys.argv) > 1 else 512
spec = [s for s in bench.cpjobs.JOBS["vgg16"]() if s["c"] == c][0]
X, W2, Y = bench.cpjobs.synth(spec)[:3]
pr = LayerProblem(ctx, X, W2, Y)
for rep in range(3):
    out = prune_layer(pr, spec["rank"], 1e-3, rng=np.random.RandomState(1234 + spec["layer_id"]), mode="device")
    ctx.sync()
print("kept", int(out[0].sum()))
pr.free()
