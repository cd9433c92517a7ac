"""Which leg of bench.py leaves a process slower?  Runs ONE leg (argv[1]) and then the short vgg16_5x job + one R3 conv; prints both
times.  Legs: none | job (ResidentLayerSet vgg16, 20 jobs) | seq (the PCIe-inclusive sequential pass) | block | pipelined |
gather | cpu (the CPU port on three small layers) | alone (every layer alone, latency mode)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench                                   # noqa: E402,F401  (GPU_MAX_HW_QUEUES, sys.path)
import cpmi355                                 # noqa: E402
from benchkit.common import CD_FLAGS, cpjobs   # noqa: E402
from benchkit.extras import short_job          # noqa: E402
from cpmi355 import shard                      # noqa: E402
from cpmi355.pruner import LayerProblem, prune_layer, rng_rewind   # noqa: E402

leg = sys.argv[1] if len(sys.argv) > 1 else "none"
specs = cpjobs.JOBS["vgg16"]()
data = {s["layer_id"]: cpjobs.synth(s)[:3] for s in specs} if leg in ("job", "seq", "pipelined", "alone") else {}
t0 = time.perf_counter()
if leg in ("job", "pipelined", "alone"):
    rset = shard.ResidentLayerSet(0, specs, lambda s: data[s["layer_id"]], per_stream=1, flags=CD_FLAGS, borrow_results=True)
    for _ in range(20):
        rset()
    if leg == "pipelined":
        import threading
        rset2 = shard.ResidentLayerSet(0, specs, lambda s: data[s["layer_id"]], per_stream=1, flags=CD_FLAGS, borrow_results=True)
        th = [threading.Thread(target=lambda r=r: [r() for _ in range(15)]) for r in (rset, rset2)]
        [t.start() for t in th]
        [t.join() for t in th]
        rset2.close()
    if leg == "alone":
        for j, pr in rset.problems().items():
            ch = [c_ for c_ in rset.chunks if j in c_["members"]][0]
            rng, mark = ch["rngs"][ch["members"].index(j)], ch["marks"][ch["members"].index(j)]
            for _ in range(2):
                rng_rewind(rng, mark)
                prune_layer(pr, specs[j]["rank"], 1e-3, rank_tol=.1, rng=rng, mode="device")
    rset.close()
elif leg == "seq":
    ctx0 = cpmi355.Context(0)
    for _ in range(3):
        for spec in specs:
            X, W2, Y = data[spec["layer_id"]]
            pr = LayerProblem(ctx0, X, W2, Y, flags=CD_FLAGS, defer_upload=True)
            prune_layer(pr, spec["rank"], 1e-3, rank_tol=.1, rng=np.random.RandomState(1234 + spec["layer_id"]), mode="device")
            pr.free()
    ctx0.close()
elif leg == "block":
    from benchkit.block import block_single_instance, close_workers
    single, group = block_single_instance(0)
    close_workers(group)
elif leg == "gather":
    from benchkit.gather import bench_patch_gather
    bench_patch_gather(0)
elif leg == "cpu":
    from benchkit.cpu_legs import cpu_port_seconds
    cpu_port_seconds(specs[:3], threads=8)
t_leg = time.perf_counter() - t0
r = short_job(0, "vgg16_5x", min_seconds=0.6, warmup_seconds=0.4)
print("after %-9s (%.1f s): vgg16_5x job %.2f ms" % (leg, t_leg, r["job_ms"]), flush=True)
