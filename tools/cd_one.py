"""One CD fit (c = argv[1], flags = argv[2], 20 epochs, tol 0) -- the target of per-counter rocprofv3 passes.
Prints the number of coordinate steps so counters can be divided by it."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd"))
import numpy as np  # noqa: E402
import cpmi355  # noqa: E402

c = int(sys.argv[1]) if len(sys.argv) > 1 else 256
flags = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = cpmi355.Context(0)
rs = np.random.RandomState(3)
M = 20000
Z = rs.randn(M, c) * (0.2 + rs.rand(c))
wtrue = np.where(rs.rand(c) < 0.4, rs.randn(c), 0.0)
y = Z @ wtrue + 0.1 * rs.randn(M)
Zc = Z - Z.mean(0)
yc = y - y.mean()
Q, q, yty = np.ascontiguousarray(Zc.T @ Zc), Zc.T @ yc, float(yc @ yc)
Qd, qd = ctx.to_device(Q), ctx.to_device(q)
sd = ctx.to_device(np.array([yty, 0, M, 0], dtype=np.float64))
wd = ctx.zeros(c * 8)
r = ctx.enet_cd_gram(Qd, c, qd, sd, c, 0.05 * np.abs(q).max(), 0.0, 7, wd, flags=flags, tol=0.0, max_iter=20)
print("steps", r.n_iter * c, "n_iter", r.n_iter, "nnz", r.nnz)
