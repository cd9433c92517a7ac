"""benchkit.gather -- row a1 (Net.extract_XY): the sampled-point im2col measured against the HBM roofline."""
import time

import numpy as np

def bench_patch_gather(device, C=256, H=56, W=56, B=10, P=10, nb=50, k=3, pad=1, reps=20):
    """a1 (Net.extract_XY, lib/net.py:534-684): the sampled-point im2col of SURVEY.md 8d's gather workload -- B = 10 images,
    C = 256 channels of 56 x 56, 10 sampled points per batch, 50 batches => N = 5000 rows of C*k*k floats (ReLU fused).
    The feature maps of all batches are resident in HBM ([nb, B, C, H, W] float32 = 1.6 GB); timed: (a) ONE launch over
    all batches (cp_patch_gather_batches), (b) one cp_patch_gather call per batch as the facade issues them while the
    provider's forward passes run.  Algorithmic bytes = the rows written + the same bytes read (8 N C k^2)."""
    import cpmi355
    ctx = cpmi355.Context(device)
    try:
        rs = np.random.RandomState(7)
        one = rs.randn(B, C, H, W).astype(np.float32)
        fm = ctx.empty(nb * one.nbytes)
        for b in range(nb):           # the same batch image nb times: contents are irrelevant to a gather's speed
            ctx._check(ctx.lib.cp_memcpy_h2d(ctx.h, fm.ptr + b * one.nbytes, one.ctypes.data, one.nbytes), "cp_memcpy_h2d")
        xs = rs.randint(0, H, nb * P).astype(np.int32)
        ys = rs.randint(0, W, nb * P).astype(np.int32)
        N = nb * P * B
        out = ctx.empty(N * C * k * k * 4)
        alg = 8.0 * N * C * k * k
        res = {}
        for name in ("one_launch", "per_batch_calls"):
            def run():
                if name == "one_launch":
                    ctx.patch_gather_batches(fm, nb, B, C, H, W, xs, ys, P, k, pad, 1, True, out)
                else:
                    for b in range(nb):
                        ctx.patch_gather(fm.ptr + b * one.nbytes, B, C, H, W, xs[b * P:(b + 1) * P],
                                         ys[b * P:(b + 1) * P], k, pad, 1, True, out, b * P * B)
            run()
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(reps):
                run()
            ctx.sync()
            ms = (time.perf_counter() - t0) / reps * 1e3
            res[name] = {"ms": round(ms, 4), "GBps_algorithmic": round(alg / ms / 1e6, 1),
                         "frac_of_hbm_peak": round(alg / (ms * 1e-3) / 8.0e12, 4)}
            if name == "one_launch":     # the kernel alone, HIP events on the launch stream (the call also uploads the points)
                ctx.enable_stage_timing(1)
                kms = []
                for _ in range(5):
                    run()
                    ctx.sync()
                    kms.append(dict(ctx.last_stage_times()).get("gather_kernel", 0.0))
                ctx.enable_stage_timing(0)
                kms = float(np.median(kms))
                if kms > 0:
                    res[name].update(kernel_ms=round(kms, 4), kernel_GBps_algorithmic=round(alg / kms / 1e6, 1),
                                     kernel_frac_of_hbm_peak=round(alg / (kms * 1e-3) / 8.0e12, 4))
        res["workload"] = "B=%d C=%d %dx%d k=%d pad=%d, %d points x %d batches: N=%d rows, %.1f MB written" % (
            B, C, H, W, k, pad, P, nb, N, alg / 2e6)
        res["note"] = ("HBM-bound gather: every sampled k-wide run of a channel row costs a whole 64-byte fabric request "
                       "(3 x 64 B fetched per 36 B used at k = 3), so the algorithmic rate is bounded near "
                       "8 TB/s x 72 / (192 + 36) = 2.5 TB/s; see DESIGN.md")
        return res
    finally:
        ctx.close()

