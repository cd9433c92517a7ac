"""cp_svd_rows vs numpy on random and on weight-shaped matrices: singular values, V up to sign, V SH = M."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd"))
import numpy as np, cpmi355
ctx = cpmi355.Context(0)
lib = ctx.lib
lib.cp_svd_rows.restype = ctypes.c_int
lib.cp_svd_rows.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                            ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
for (m, n, r) in [(24, 40, 24), (96, 96, 48), (383, 768, 200), (768, 768, 384)]:
    rs = np.random.RandomState(m)
    M = rs.randn(m, n) * (0.05 + rs.rand(m, 1))
    Md = ctx.to_device(M)
    sd, Vd, Hd = ctx.empty(r * 8), ctx.empty(r * m * 8), ctx.empty(r * n * 8)
    sw = ctypes.c_int()
    rc = lib.cp_svd_rows(ctx.h, Md.ptr, m, n, r, sd.ptr, Vd.ptr, Hd.ptr, ctypes.byref(sw))
    ctx.sync(); t0 = time.perf_counter()
    rc = lib.cp_svd_rows(ctx.h, Md.ptr, m, n, r, sd.ptr, Vd.ptr, Hd.ptr, ctypes.byref(sw))
    dt = time.perf_counter() - t0
    assert rc == 0, lib.cp_last_error(ctx.h)
    s = ctx.to_host(sd, (r,), np.float64); Vt = ctx.to_host(Vd, (r, m), np.float64); SH = ctx.to_host(Hd, (r, n), np.float64)
    U, S, Ht = np.linalg.svd(M, full_matrices=False)
    es = np.abs(s - S[:r]).max() / S[0]
    sign = np.sign(np.sum(Vt * U[:, :r].T, axis=1))
    ev = np.abs(Vt * sign[:, None] - U[:, :r].T).max()
    eh = np.abs(SH * sign[:, None] - (S[:r, None] * Ht[:r])).max() / S[0]
    rec = np.linalg.norm(Vt.T @ SH - (U[:, :r] * S[:r]) @ Ht[:r]) / np.linalg.norm(M)
    print("m=%d n=%d r=%d sweeps=%d  %.1f ms  sigma %.1e  V %.1e  SH %.1e  recon %.1e" % (m, n, r, sw.value, dt * 1e3, es, ev, eh, rec))
