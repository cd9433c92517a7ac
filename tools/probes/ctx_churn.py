"""Does a long-lived process get slower as it creates and closes contexts?  Builds the vgg16_5x job's ResidentLayerSet, times a few
jobs, closes it -- several times over -- and prints the job time of every generation next to the process's thread count,
open file descriptors and the time of 2000 empty-ish launches on a fresh context."""
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd"))
import cpmi355                      # noqa: E402
from cpmi355 import jobs, shard     # noqa: E402

specs = jobs.vgg16_5x()
data = {s["layer_id"]: jobs.synth(s)[:3] for s in specs}


def launches():
    ctx = cpmi355.Context(0)
    buf = ctx.zeros(1 << 20)
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(2000):
        ctx._check(ctx.lib.cp_memset(ctx.h, buf.ptr, 0, 4096), "cp_memset")
    ctx.sync()
    dt = (time.perf_counter() - t0) / 2000 * 1e6
    buf.free()
    ctx.close()
    return dt


for gen in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    rset = shard.ResidentLayerSet(0, specs, lambda s: data[s["layer_id"]], per_stream=1, flags=0, borrow_results=True)
    roots = [ch["ctxs"][0] for ch in rset.chunks]
    for _ in range(6):
        rset()
    for cx in roots:
        cx.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        rset()
    for cx in roots:
        cx.sync()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    rset.close()
    print("generation %d: job %.2f ms, threads %d, fds %d, 4 KB memset launch %.1f us" % (
        gen, ms, threading.active_count(), len(os.listdir("/proc/self/fd")), launches()), flush=True)
