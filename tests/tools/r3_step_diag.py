"""Diagnostic: first 3C step (conv1_2) of the n02 net -- device VH_decompose / ITQ_decompose vs the CPU restatements on
identical inputs."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "channel-pruning_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import cp_oracle
import portable_net
import lib.decompose as D
from lib.net import Net
from portable_provider import PortableProvider
g = np.load(os.path.join(ROOT, "tests/golden/n02_vgg_r3_3c.npz"))
p = json.loads(str(g["params"]))
layers, batches = portable_net.vgg_like(seed=p["seed"], chans=[tuple(c) for c in p["chans"]], B=p["B"], HW=p["HW"], nBatches=p["nBatches"])
net = Net(None, PortableProvider(layers, batches), nBatches=p["nBatches"], nPointsPerLayer=p["nPoints"], graph=layers)
np.random.seed(5)
feats, points = net.extract_features(names=net.convs, save=1)
net.load_frozen(feats_dict=feats, points_dict=points)
rel = lambda a, b: np.linalg.norm(np.asarray(a, float) - np.asarray(b, float)) / np.linalg.norm(b)
conv, rank = "conv1_2", 22
weights = net.param_data(conv)
Y = feats[conv] - net.param_b_data(conv)
x = net.extract_XY(net.bottom_names[conv][0], conv)
X = np.rollaxis(x.reshape((-1, 3, 3, x.shape[1])), 3, 1).copy()
V, H, VHr, b = D.VH_decompose(weights, rank=rank, DEBUG=True, X=X, Y=Y)
Vo, Ho, VHro, bo = cp_oracle.vh_decompose_oracle(weights.astype(np.float64), rank, X, Y)
print("VH: VHr rel %.2e  b rel %.2e" % (rel(VHr, VHro), rel(b, bo)))
# svd-only part
V0, H0, VHr0 = D.VH_decompose(weights, rank=rank)
V0o, H0o, VHr0o = cp_oracle.vh_decompose_oracle(weights.astype(np.float64), rank)
print("VH svd only: VHr rel %.2e" % rel(VHr0, VHr0o))
# nonlinear_fc on the projected X
Xv = np.tensordot(X, Vo.reshape(Vo.shape[0], Vo.shape[1], Vo.shape[2]), [[1, 2], [1, 2]])   # N, w, rank
Xv = np.transpose(Xv, [0, 2, 1]).reshape(X.shape[0], -1)
c1, i1 = D.nonlinear_fc(Xv, Y)
c2, i2 = cp_oracle.nonlinear_fc_oracle(Xv, Y)
print("nonlinear_fc on oracle's Xv: coef rel %.2e intercept rel %.2e" % (rel(c1, c2), rel(i1, i2)))
Xc = Xv - Xv.mean(0)
sv = np.linalg.svd(Xc, compute_uv=False)
print("cond(Xv centred) %.3e  N %d p %d" % (sv[0] / sv[-1], Xv.shape[0], Xv.shape[1]), "fallback info", D.last_call_info.get("refit_info"))
# ITQ on the oracle's state
net.set_param_b(conv, bo)
net.set_param_data(conv, VHro)
cur, _ = net.extract_features(names=conv, points_dict=net._points_dict, save=1)
W1, W2, B, W12 = D.ITQ_decompose(cur[conv], feats[conv], Ho, rank, bias=net.param_b_data(conv), Wr=VHro)
W1o, W2o, Bo, W12o = cp_oracle.itq_decompose_oracle(cur[conv], feats[conv], Ho, rank, bias=net.param_b_data(conv), Wr=VHro)
print("ITQ: W12 rel %.2e  B rel %.2e" % (rel(W12, W12o), rel(B, Bo)))
G = cur[conv] - cur[conv].mean(0)
ev = np.linalg.eigvalsh(G.T @ G)[::-1]
print("eig(G^T G) ratio to max (smallest 6):", (ev[-6:] / ev[0]))
