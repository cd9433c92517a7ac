"""cp_svd_rows on Gaussian weight-shaped matrices (the r3 workload's (c*3) x (n*3) shapes), one library per run:
    CP_LIB_PATH=build_variants/lib_x.so python tools/probes/svd_variants.py
prints ms per SVD (best of 3), sweeps, and the error against numpy for the smaller shapes."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd"))
import numpy as np, cpmi355
ctx = cpmi355.Context(0)
lib = ctx.lib
lib.cp_svd_rows.restype = ctypes.c_int
lib.cp_svd_rows.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                            ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
out = []
for (m, n, r) in [(333, 768, 110), (768, 1536, 233), (1335, 1536, 302), (1536, 1536, 398)]:
    rs = np.random.RandomState(m)
    M = rs.randn(m, n) * 0.05
    Md = ctx.to_device(M)
    sd, Vd, Hd = ctx.empty(r * 8), ctx.empty(r * m * 8), ctx.empty(r * n * 8)
    sw = ctypes.c_int()
    best = 1e9
    for _ in range(3):
        ctx.sync(); t0 = time.perf_counter()
        rc = lib.cp_svd_rows(ctx.h, Md.ptr, m, n, r, sd.ptr, Vd.ptr, Hd.ptr, ctypes.byref(sw))
        best = min(best, time.perf_counter() - t0)
        assert rc == 0
    s = ctx.to_host(sd, (r,), np.float64)
    chk = ""
    if m <= 768:
        S = np.linalg.svd(M, compute_uv=False)
        chk = " sigma err %.1e" % (np.abs(s - S[:r]).max() / S[0])
    import hashlib
    out.append("m=%d n=%d: %.1f ms, %d sweeps%s, digest %s" % (m, n, best * 1e3, sw.value, chk, hashlib.md5(ctx.to_host(Vd, (r, m), np.float64).tobytes()).hexdigest()[:8]))
print(os.environ.get("CP_LIB_PATH", "default"), "|", " | ".join(out))
