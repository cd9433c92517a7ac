"""Diagnostic: Net.R3() vs the reference golden n02 -- per-layer relative differences and selection mismatches."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "channel-pruning_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import portable_net
import lib.cfgs as cfgs
import lib.decompose as D
from lib.cfgs import c as dcfgs
from lib.net import Net
from portable_provider import PortableProvider
g = np.load(os.path.join(ROOT, "tests/golden/n02_vgg_r3_3c.npz"))
p = json.loads(str(g["params"]))
layers, batches = portable_net.vgg_like(seed=p["seed"], chans=[tuple(c) for c in p["chans"]], B=p["B"], HW=p["HW"], nBatches=p["nBatches"])
net = Net(None, PortableProvider(layers, batches), nBatches=p["nBatches"], nPointsPerLayer=p["nPoints"], graph=layers)
np.random.seed(5)
feats, points = net.extract_features(names=net.convs, save=1)
net.load_frozen(feats_dict=feats, points_dict=points)
cfgs.alpha = 1e-3
np.random.seed(78)
orig = net.dictionary_kernel
def logged(*a, **k):
    r = orig(*a, **k)
    print("dictionary_kernel", a[0], a[3], "rank", a[2], "kept", int(r[0].sum()), "fits", D.last_call_info["fits"], "alpha", cfgs.alpha)
    return r
net.dictionary_kernel = logged
WPQ, _ = net.R3()
rel = lambda a, b: np.linalg.norm(np.asarray(a, float) - b) / np.linalg.norm(b)
for name in net.convs:
    print(name, "finalW rel %.2e  finalb abs %.2e" % (rel(net.param_data(name), g["finalW:" + name]), np.abs(net.param_b_data(name) - g["finalb:" + name]).max()))
for k in json.loads(str(g["sel_keys"])):
    print(k, "selection mismatches", int((net.selection[k] != g["sel:" + k]).sum()), "kept", int(net.selection[k].sum()), int(g["sel:" + k].sum()))
print("alpha", cfgs.alpha, float(g["alpha_out"]))
