"""Pageable host -> HBM upload rate of cp_memcpy_h2d (Context.to_device) for a fresh array every time (what the drop-in
sees) and for the same array again; CP_UPLOAD_THREADS=1 selects the plain hipMemcpyAsync path.  python tools/h2d_bench.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd"))
import numpy as np
from cpmi355 import capi

ctx = capi.Context(0)
shape = (5000, 512 * 9)
dst = ctx.empty(int(np.prod(shape)) * 4)
rs = np.random.RandomState(0)
fresh, same = [], []
X = None
for rep in range(6):
    X = rs.rand(*shape).astype(np.float32)          # 92 MB, never seen by the runtime
    ctx.sync()
    t0 = time.perf_counter(); ctx.to_device(X, dst); ctx.sync(); fresh.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); ctx.to_device(X, dst); ctx.sync(); same.append(time.perf_counter() - t0)
back = ctx.to_host(dst, shape, np.float32)
print("threads=%s  fresh ms %s  same-array ms %s  GB/s fresh %.1f  ok=%s" % (
    os.environ.get("CP_UPLOAD_THREADS", "default"), [round(t * 1e3, 2) for t in fresh], [round(t * 1e3, 2) for t in same],
    X.nbytes / min(fresh) / 1e9, bool(np.array_equal(back, X))))
