"""Wall-clock split of one resident prune_layer() call (host view) next to the device stage times."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, cpmi355, cp_oracle
ctx = cpmi355.Context(0)
X, W2, Y, B2 = cp_oracle.synth_layer(32, 5000, 256, 256, 3)
prob = cpmi355.LayerProblem(ctx, X, W2, Y, flags=3)
for rep in range(4):
    rng = np.random.RandomState(1266)
    ctx.enable_stage_timing(rep == 3)
    t = [time.perf_counter()]
    samples = rng.randint(0, 5000, 250); prob.lasso_gram(samples); t.append(time.perf_counter())
    a = prob.alpha_search(128, 1e-3, .1, rng, mode="device"); t.append(time.perf_counter())
    idxs = prob.mask(); t.append(time.perf_counter())
    info = ctx.lstsq_refit(prob.Xd, prob.x_dtype, prob.N, prob.c, prob.kk, idxs.astype(np.uint8), prob.Yd, prob.n, 0.0, prob.Wout, prob.bout); t.append(time.perf_counter())
    W = ctx.to_host(prob.Wout, (prob.n, int(info.p)), np.float64); b = ctx.to_host(prob.bout, (prob.n,), np.float64); t.append(time.perf_counter())
names = ["lasso_gram (async)", "alpha_search (sync)", "mask d2h", "refit (sync)", "W,b d2h"]
for n, a0, a1 in zip(names, t[:-1], t[1:]): print("%-22s %.3f ms" % (n, (a1 - a0) * 1e3))
print("total %.3f ms" % ((t[-1] - t[0]) * 1e3))
st = ctx.last_stage_times(); print(st); print("device stage sum %.3f ms" % sum(v for _, v in st))
