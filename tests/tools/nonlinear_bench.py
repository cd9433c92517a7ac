"""nonlinear_fc at BASELINE size (N=5000, 131 kept channels x 9, n=256): device time vs the numpy restatement."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, cpmi355, cp_oracle
ctx = cpmi355.Context(0)
X, W2, Y, B2 = cp_oracle.synth_layer(40, 5000, 256, 256, 3)
prob = cpmi355.LayerProblem(ctx, X, W2, Y)
mask = np.zeros(256, bool); mask[:131] = True
prob.refit_nonlinear(mask)
ctx.sync(); t0 = time.perf_counter(); W, b = prob.refit_nonlinear(mask); dt = time.perf_counter() - t0
print("device nonlinear_fc (50 regressions, p=%d): %.1f ms" % (W.shape[1], dt * 1e3))
for name, ms in ctx.last_stage_times() if False else []: print(name, ms)
if len(sys.argv) > 1 and sys.argv[1] == "cpu":
    Xk = X[:, mask].reshape(5000, -1).astype(np.float64)
    t0 = time.perf_counter(); cref, bref = cp_oracle.nonlinear_fc_oracle(Xk, Y, engine="sklearn"); dc = time.perf_counter() - t0
    print("CPU port (sklearn LinearRegression x 50): %.1f s; rel.err W %.2e" % (dc, np.linalg.norm(W - cref) / np.linalg.norm(cref)))
