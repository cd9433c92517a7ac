"""CD kernel micro-benchmark: ns/step, cycles/step, implied clock, for several channel counts.
Usage (GPU box): python tools/cd_bench.py [path/to/libcpmi355_variant.so]"""
import ctypes
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd"))
import numpy as np  # noqa: E402
import cpmi355  # noqa: E402
from cpmi355 import capi  # noqa: E402

CS = tuple(int(v) for v in os.environ.get("CD_BENCH_C", "64,128,256,512,1024,2048").split(","))
FLAGS = tuple(int(v) for v in os.environ.get("CD_BENCH_FLAGS", "0,3").split(","))
if len(sys.argv) > 1:
    capi.LIB_PATH = sys.argv[1]
print("CP_CD_TEAM=%s CP_CD_EXACT_DIV=%s" % (os.environ.get("CP_CD_TEAM", "(default 1)"), os.environ.get("CP_CD_EXACT_DIV", "(default 0)")))
ctx = cpmi355.Context(0)
lib = ctx.lib
lib.cp_debug_cd_cycles.restype = ctypes.c_int
lib.cp_debug_cd_cycles.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong)]


def problem(c, M=20000, seed=3):
    rs = np.random.RandomState(seed)
    Z = rs.randn(M, c) * (0.2 + rs.rand(c))
    wtrue = np.where(rs.rand(c) < 0.4, rs.randn(c), 0.0)
    y = Z @ wtrue + 0.1 * rs.randn(M)
    Zc = Z - Z.mean(0)
    yc = y - y.mean()
    return np.ascontiguousarray(Zc.T @ Zc), Zc.T @ yc, float(yc @ yc), M


for c in CS:
    Q, q, yty, M = problem(c)
    Qd, qd = ctx.to_device(Q), ctx.to_device(q)
    sd = ctx.to_device(np.array([yty, 0, M, 0], dtype=np.float64))
    for flags in FLAGS:
        wd = ctx.zeros(c * 8)
        l1 = 0.05 * np.abs(q).max()
        ctx.enet_cd_gram(Qd, c, qd, sd, c, l1, 0.0, 7, wd, flags=flags)  # warm
        rows = []
        for rep in range(3):
            wd = ctx.zeros(c * 8)
            ctx.sync()
            t0 = time.perf_counter()
            r = ctx.enet_cd_gram(Qd, c, qd, sd, c, l1, 0.0, 7 + rep, wd, flags=flags, tol=0.0, max_iter=20)
            dt = time.perf_counter() - t0
            dbg = (ctypes.c_ulonglong * 8)()
            lib.cp_debug_cd_cycles(ctx.h, dbg)
            steps = r.n_iter * c
            rows.append((dt / steps * 1e9, dbg[0] / max(dbg[1], 1), dbg[0] / dt / 1e9, r.n_iter, r.nnz))
            phases = [dbg[2 + i] / max(dbg[1], 1) for i in range(4)]
        best = min(rows)
        print("c=%4d flags=%d  ns/step %.1f  cycles/step %.1f  implied GHz %.2f  n_iter %d nnz %d" % (
            (c, flags) + best), " per step: chain wave waits %.1f cycles, %.3f repairs, keeper 0 waits %.1f cycles (%.4f blocks)" % tuple(phases), flush=True)
