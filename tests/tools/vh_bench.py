"""VH_decompose at VGG conv3 size (weights 256x256x3x3, rank 128, N=5000 sampled patches): device path vs the
numpy/scipy restatement of the reference on this box's host."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, cp_oracle
import lib.decompose as D
X, W2, Y, B2 = cp_oracle.synth_layer(41, 5000, 256, 256, 3)
W = W2.astype(np.float64)
D.VH_decompose(W, rank=128)                      # warm-up
t0 = time.perf_counter(); V, H, VHr = D.VH_decompose(W, rank=128); t_svd = time.perf_counter() - t0
t0 = time.perf_counter(); V2, H2, VHr2, b = D.VH_decompose(W, rank=128, X=X, Y=Y); t_full = time.perf_counter() - t0
print("device VH_decompose: SVD-only %.1f ms, with X/Y refit (nonlinear_fc on Xv) %.1f ms (incl. 46 MB upload)" % (t_svd * 1e3, t_full * 1e3))
if len(sys.argv) > 1 and sys.argv[1] == "cpu":
    t0 = time.perf_counter(); Vr, Hr, VHrr = cp_oracle.vh_decompose_oracle(W, rank=128); c_svd = time.perf_counter() - t0
    t0 = time.perf_counter(); Vr2, Hr2, VHrr2, br = cp_oracle.vh_decompose_oracle(W, rank=128, X=X.astype(np.float64), Y=Y); c_full = time.perf_counter() - t0
    print("CPU port: SVD-only %.1f ms, with refit %.1f s" % (c_svd * 1e3, c_full))
    print("VHr rel.err: svd %.2e  refit %.2e" % (np.linalg.norm(VHr - VHrr) / np.linalg.norm(VHrr), np.linalg.norm(VHr2 - VHrr2) / np.linalg.norm(VHrr2)))
