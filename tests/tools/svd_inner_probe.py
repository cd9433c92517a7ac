"""Cold block-Jacobi SVD at the VH sizes of VGG-16 (768^2: conv3, 1536^2: conv4 / conv5): time and sweeps for the
CP_JACOBI_INNER setting of this process (inner sweeps of the 16 x 16 diagonalisation per block pair)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd"))
import numpy as np
from cpmi355 import default_context
ctx = default_context()
rs = np.random.RandomState(3)
for m, r in ((768, 384), (1536, 398)):
    M = rs.randn(m, m) * (1.0 / np.sqrt(np.arange(1, m + 1)))[None, :]     # graded columns, full rank
    ctx.svd_rows(M[:64, :64].copy(), 8)
    ts = []
    for rep in range(2):
        t0 = time.perf_counter(); s, Vt, SH = ctx.svd_rows(M, r); ts.append(time.perf_counter() - t0)
    sref = np.linalg.svd(M, compute_uv=False)[:r]
    print("inner=%s m=%d: %.1f ms, %d sweeps, sigma rel.err %.1e" % (os.environ.get("CP_JACOBI_INNER", "1"), m, min(ts) * 1e3,
          ctx.last_svd_sweeps, np.abs(s - sref).max() / sref[0]))
