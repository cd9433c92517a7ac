"""One layer alone through prune_layer(latency_mode=True), stage by stage: which stage of a small layer's call is slow.
usage: CP_LIB_PATH=... CP_CHOL_FORM=... python tools/probes/small_layer_latency.py [V01 V03 V08]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd"))
import cpmi355                                   # noqa: E402
from cpmi355 import jobs                         # noqa: E402
from cpmi355.pruner import LayerProblem, prune_layer   # noqa: E402

want = sys.argv[1:] or ["V01", "V03", "V08"]
ctx = cpmi355.Context(0)
for spec in jobs.vgg16_4x():
    if spec["name"][:3] not in want:
        continue
    X, W2, Y, _ = jobs.synth(spec)
    pr = LayerProblem(ctx, X, W2, Y, flags=0)
    ctx.enable_stage_timing(1)
    ts = []
    for it in range(4):
        t0 = time.perf_counter()
        prune_layer(pr, spec["rank"], 1e-3, rank_tol=.1, rng=np.random.RandomState(1234 + spec["layer_id"]), mode="device")
        ts.append((time.perf_counter() - t0) * 1e3)
    st = dict(ctx.last_stage_times())
    print(spec["name"][:3], "ms", [round(t, 2) for t in ts], {k: round(v, 3) for k, v in st.items() if v > 0.2})
    pr.free()
ctx.close()
