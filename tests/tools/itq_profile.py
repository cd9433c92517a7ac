"""Where ITQ_decompose's time goes at conv3 size: stage times of cp_itq_iterate and the host-side remainder."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, cp_oracle
import lib.decompose as D
from cpmi355 import default_context
ctx = default_context()
X, W2, Y, B2 = cp_oracle.synth_layer(42, 5000, 256, 256, 3)
feature = Y + 0.05 * np.random.RandomState(42).randn(*Y.shape)
ctx.itq_iterate(feature[:600], Y[:600], 128)
ctx.enable_stage_timing(1)
for rep in range(2):
    t0 = time.perf_counter(); T, ym, um = ctx.itq_iterate(feature, Y, 128); dt = time.perf_counter() - t0
    print("cp_itq_iterate %.1f ms" % (dt * 1e3), dict(ctx.last_stage_times()), "jacobi sweeps over the 50 alternations:",
          ctx.lib.cp_debug_itq_sweeps(ctypes.c_void_p(ctx.h)))
for lr in (False, True):
    t0 = time.perf_counter(); s, Lt, R = ctx.svd_rows(T, 128, lowrank=lr); print("final svd_rows(T, lowrank=%s) %.1f ms, %d sweeps" % (lr, (time.perf_counter() - t0) * 1e3, ctx.last_svd_sweeps))
t0 = time.perf_counter(); D.ITQ_decompose(feature, Y, W2.astype(np.float64), 128, bias=B2.astype(np.float64)); print("ITQ_decompose total %.1f ms" % ((time.perf_counter() - t0) * 1e3))
