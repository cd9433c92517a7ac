"""Pageable host -> HBM upload rate of cp_memcpy_h2d: one call against the array cut into row slabs uploaded by several
threads (own context = own stream each; ctypes drops the GIL during the call).  python tools/h2d_bench.py"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd"))
import numpy as np
from cpmi355 import capi

ctx = capi.Context(0)
X = np.random.RandomState(0).rand(5000, 512 * 9)            # 184 MB float64, pageable
dst = ctx.empty(X.nbytes)
for nthreads in (1, 2, 4, 8, 16):
    ctxs = [capi.Context(0) for _ in range(nthreads)]
    rows = np.linspace(0, X.shape[0], nthreads + 1).astype(int)

    def work(k):
        part = X[rows[k]:rows[k + 1]]
        ctxs[k]._check(ctxs[k].lib.cp_memcpy_h2d(ctxs[k].h, int(dst.ptr + int(rows[k]) * X.shape[1] * 8), part.ctypes.data, int(part.nbytes)), "h2d")
        ctxs[k].sync()

    best = 1e9
    for rep in range(4):
        t0 = time.perf_counter()
        ts = [threading.Thread(target=work, args=(k,)) for k in range(nthreads)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        best = min(best, time.perf_counter() - t0)
    back = ctx.to_host(dst, X.shape, np.float64)
    print("threads %2d: %.1f ms  %.1f GB/s  ok=%s" % (nthreads, best * 1e3, X.nbytes / best / 1e9, bool(np.array_equal(back, X))))
    for c in ctxs:
        c.close()
