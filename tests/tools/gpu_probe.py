"""Quick on-GPU probe: MFMA f64 / HBM copy ceilings, CD step latency, per-stage times."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import cpmi355, cp_oracle
from cpmi355 import capi
ctx = cpmi355.Context(0)
out = {}
out["mfma_f64_tflops"] = ctx.probe_mfma_f64()
out["hbm_copy_gbps"] = ctx.probe_hbm_copy(1 << 30)
print(out, flush=True)
for (c, n) in ((128, 256), (256, 256), (512, 512)):
    X, W2, Y, B2 = cp_oracle.synth_layer(40, 5000, c, n, 3)
    prob = cpmi355.LayerProblem(ctx, X, W2, Y)
    ctx.enable_stage_timing(True)
    rs = np.random.RandomState(7)
    samples = rs.randint(0, 5000, 250)
    prob.lasso_gram(samples); prob.lasso_gram(samples)
    st = ctx.last_stage_times()
    ctx.enable_stage_timing(False)
    # CD: time individual fits
    res = {}
    for recip in (0, 1):
        prob.flags = capi.CP_CD_RECIPROCAL if recip else 0
        prob.reset_w(); prob.fits = []
        t0 = time.perf_counter()
        for a, s in ((0.008, 11), (0.012, 12), (0.010, 13), (0.011, 14)):
            prob.solve(a, s)
        ctx.sync(); dt = time.perf_counter() - t0
        steps = sum(f[2] for f in prob.fits) * c
        res["recip%d" % recip] = dict(fits=prob.fits, seconds=dt, steps=steps, ns_per_step=dt / steps * 1e9)
    # search in one launch
    rng = np.random.RandomState(3)
    t0 = time.perf_counter(); a = prob.alpha_search(c // 2, 1e-3, .1, rng, mode="device"); dt_dev = time.perf_counter() - t0
    nsteps = sum(f[2] for f in prob.fits) * c
    idxs = prob.mask()
    ctx.enable_stage_timing(True)
    prob.refit(idxs); prob.refit(idxs)
    st2 = ctx.last_stage_times()
    ctx.enable_stage_timing(False)
    t0 = time.perf_counter(); prob.refit(idxs); dt_refit = time.perf_counter() - t0
    print(json.dumps(dict(c=c, n=n, lasso_stages=st, cd=res, search_s=dt_dev, search_steps=nsteps,
                          search_ns_per_step=dt_dev / nsteps * 1e9, kept=int(idxs.sum()), refit_stages=st2,
                          refit_s=dt_refit)), flush=True)
    prob.free()
