import sys, os
ROOT="/root/repo"
sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.join(ROOT,"tests"))
import numpy as np, cpmi355, cp_oracle
from test_gpu_parity import _cd_problem
ctx = cpmi355.Context(0)
for c in (8, 16, 24):
  for flags in (0,1,2,3):
    Q, q, yty, M = _cd_problem(c)
    Qd, qd = ctx.to_device(Q), ctx.to_device(q)
    sd = ctx.to_device(np.array([yty, 0, M, 0], dtype=np.float64))
    w_ref = np.zeros(c); wd = ctx.zeros(c * 8)
    amax = np.abs(q).max() / M
    for i, (frac, seed) in enumerate([(0.5, 12345), (0.2, 987654321), (0.05, 1), (0.3, 2147483646)]):
        l1 = frac * amax * M
        _, stats, n_ref = cp_oracle.enet_cd_gram(w_ref, l1, 0.0, Q, q, yty, 1000, 1e-4, seed, recip=bool(flags & 1), delta=bool(flags & 2))
        r = ctx.enet_cd_gram(Qd, c, qd, sd, c, l1, 0.0, seed, wd, flags=flags)
        w = ctx.to_host(wd, (c,), np.float64)
        print(c, flags, i, r.n_iter, n_ref, np.abs(w - w_ref).max(), np.abs(w_ref).max())
