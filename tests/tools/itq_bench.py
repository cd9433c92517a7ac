"""ITQ_decompose at VGG conv3 size (N=5000 sampled outputs, n=256 filters, rank 128): device path vs the
numpy/scipy restatement of the reference on this box's host."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "channel-pruning_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, cp_oracle
import lib.decompose as D
X, W2, Y, B2 = cp_oracle.synth_layer(42, 5000, 256, 256, 3)
W = W2.astype(np.float64)
feature = Y + 0.05 * np.random.RandomState(42).randn(*Y.shape)
D.ITQ_decompose(feature[:600], Y[:600], W, 128)   # warm-up
t0 = time.perf_counter(); W1, Wo2, B, W12 = D.ITQ_decompose(feature, Y, W, 128, bias=B2.astype(np.float64)); dt = time.perf_counter() - t0
print("device ITQ_decompose (51 SVDs of 5000 x 256): %.2f s" % dt)
if len(sys.argv) > 1 and sys.argv[1] == "cpu":
    t0 = time.perf_counter(); r1, r2, rb, r12 = cp_oracle.itq_decompose_oracle(feature, Y, W, 128, bias=B2.astype(np.float64)); dc = time.perf_counter() - t0
    print("CPU port: %.1f s;  rel.err W12 %.2e  B %.2e" % (dc, np.linalg.norm(W12 - r12) / np.linalg.norm(r12), np.linalg.norm(B - rb) / np.linalg.norm(rb)))
