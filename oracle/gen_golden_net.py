"""TEST INFRASTRUCTURE ONLY -- golden vectors for the net.py rows of SURVEY.md section 8 (a1, a2, a6, a8, f3, f4) from the
UNMODIFIED reference ``lib/net.py`` (oracle/ref_net_loader.py: fake pycaffe net over oracle/portable_net.py), run in the
BUILD CONTAINER.

    python oracle/gen_golden_net.py        ->  tests/golden/n01_vgg_pruning.npz (+ n01_frozen.pickle),
                                               n02_vgg_r3_3c.npz, n03_resnet_residual.npz

n01  extract_features / freeze_images / load_frozen / extract_XY / dictionary_kernel (net.py:368-532, 749-802, 839-876,
     534-684, 1685-1735) on a VGG-shaped net: points, features, the frozen pickle the reference writes, the sampled
     patches of three (producer, consumer) pairs, and the pruning result of each pair with the alpha carry.
n02  Net.R3() -- the whole 3C loop (net.py:1292-1471): WPQ with the reference's keys and the selections.
n03  dictionary_kernel on a ResNet-shaped net with the residual-aware target: appresb + invBN (net.py:1641-1683,
     1200-1217), dcfgs.model = resnet, res.short = 1, dic.option = resnet; the shortcut was perturbed after freezing
     (what pruning the earlier layers does), so the residual term is non-zero.
"""
import contextlib
import io
import json
import os
import pickle
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import portable_net  # noqa: E402
import ref_net_loader  # noqa: E402
from gen_golden import versions  # noqa: E402

GOLDEN_DIR = os.path.join(os.path.dirname(HERE), "tests", "golden")

VGG_PARAMS = dict(seed=11, chans=((3, 24), (24, 32), (32, 64), (64, 72), (72, 120)), B=8, HW=16, nBatches=10, nPoints=10)
RES_PARAMS = dict(seed=12, B=8, HW=12, nBatches=10, nPoints=10, width=32, mid=24)


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def set_cfg(D, p):
    D.dcfgs.nBatches = p["nBatches"]
    D.dcfgs.nBatches_fc = p["nBatches"]
    D.dcfgs.nPointsPerLayer = p["nPoints"]
    D.dcfgs.dic.fitfc = 0


def gen_n01():
    R, D, cfgs = ref_net_loader.load()
    p = VGG_PARAMS
    layers, batches = portable_net.vgg_like(seed=p["seed"], chans=p["chans"], B=p["B"], HW=p["HW"], nBatches=p["nBatches"])
    net = ref_net_loader.make_reference_net(layers, batches)
    set_cfg(D, p)
    out = dict(params=json.dumps(p), versions=json.dumps(versions()))
    # freeze_images(): extract_features(save=1) + the pickle [feats_dict, points_dict] (net.py:749-802)
    tmp = tempfile.mkdtemp()
    net.pt_dir = os.path.join(tmp, "fake.prototxt")
    cfgs.frozenname = None
    np.random.seed(3)
    frozen = quiet(net.freeze_images)
    with open(frozen, "rb") as f:
        blob = f.read()
    with open(os.path.join(GOLDEN_DIR, "n01_frozen.pickle"), "wb") as f:
        f.write(blob)
    feats, points = pickle.loads(blob)
    quiet(net.load_frozen, feats_dict=feats, points_dict=points)
    net._mem = True
    out["rng_after_freeze"] = int(np.random.randint(0, 2147483647))
    pairs = [("conv1_1", "conv1_2", 20), ("pool1", "conv2_1", 27), ("conv2_1", "conv2_2", 55)]
    cfgs.alpha = 1e-3
    D.dcfgs.model = cfgs.Models.vgg
    np.random.seed(77)
    for i, (X_name, Y_name, d_prime) in enumerate(pairs):
        X = quiet(net.extract_XY, X_name, Y_name)
        assert np.array_equal(X.astype(np.float32).astype(np.float64), X)
        out["xy%d" % i] = X.astype(np.float32)
        idxs, W2, B2 = quiet(net.dictionary_kernel, X_name, None, d_prime, Y_name, None)
        out["idxs%d" % i], out["W%d" % i], out["B%d" % i] = np.asarray(idxs, dtype=bool), W2, B2
        out["alpha%d" % i] = float(cfgs.alpha)
    out["pairs"] = json.dumps(pairs)
    out["rng_next"] = int(np.random.randint(0, 2147483647))
    np.savez_compressed(os.path.join(GOLDEN_DIR, "n01_vgg_pruning.npz"), **out)
    print("n01: kept", [int(out["idxs%d" % i].sum()) for i in range(3)], "alpha", [out["alpha%d" % i] for i in range(3)],
          "pickle", len(blob), "bytes")


def gen_n02():
    R, D, cfgs = ref_net_loader.load()
    p = VGG_PARAMS
    layers, batches = portable_net.vgg_like(seed=p["seed"], chans=p["chans"], B=p["B"], HW=p["HW"], nBatches=p["nBatches"])
    net = ref_net_loader.make_reference_net(layers, batches)
    set_cfg(D, p)
    np.random.seed(5)
    feats, points = quiet(net.extract_features, names=net.convs, save=1)
    quiet(net.load_frozen, feats_dict=feats, points_dict=points)
    cfgs.alpha = 1e-3
    D.dcfgs.model = cfgs.Models.vgg
    D.dcfgs.dic.keep = 3.
    D.dcfgs.dic.vh = 1
    np.random.seed(78)
    WPQ, new_pt = quiet(net.R3)
    out = dict(params=json.dumps(p), versions=json.dumps(versions()), new_pt=new_pt, alpha_out=float(cfgs.alpha),
               rng_next=int(np.random.randint(0, 2147483647)))
    keys = []
    for k, v in WPQ.items():
        tag = k if isinstance(k, str) else "%s|%d" % k
        keys.append(tag)
        out["WPQ:" + tag] = np.asarray(v)
    out["wpq_keys"] = json.dumps(keys)
    for k, v in net.selection.items():
        out["sel:" + k] = np.asarray(v, dtype=bool)
    out["sel_keys"] = json.dumps(list(net.selection.keys()))
    for name in net.convs:       # what the net computes with after the loop (set_param_data / set_param_b)
        out["finalW:" + name] = net.param_data(name).copy()
        out["finalb:" + name] = net.param_b_data(name).copy()
    np.savez_compressed(os.path.join(GOLDEN_DIR, "n02_vgg_r3_3c.npz"), **out)
    print("n02: WPQ keys", keys, "selections", {k: int(v.sum()) for k, v in net.selection.items()}, new_pt)


def gen_n03():
    R, D, cfgs = ref_net_loader.load()
    p = RES_PARAMS
    layers, batches = portable_net.resnet_like(seed=p["seed"], B=p["B"], HW=p["HW"], nBatches=p["nBatches"],
                                               width=p["width"], mid=p["mid"])
    net = ref_net_loader.make_reference_net(layers, batches)
    set_cfg(D, p)
    D.dcfgs.model = cfgs.Models.resnet
    D.dcfgs.res.short = 1
    D.dcfgs.dic.option = cfgs.pruning_options.resnet
    try:
        names = net.convs + ["bn2a_branch1", "res2a"]
        np.random.seed(9)
        feats, points = quiet(net.extract_features, names=names, save=1)
        quiet(net.load_frozen, feats_dict=feats, points_dict=points)
        net._mem = True
        # what pruning the earlier layers does to the shortcut: perturb conv1 after freezing
        rs = np.random.RandomState(99)
        W = net.param_data("conv1")
        W[...] = (W * (1 + 0.05 * rs.randn(*W.shape))).astype(np.float32)
        out = dict(params=json.dumps(p), versions=json.dumps(versions()), conv1_W=W.copy(), names=json.dumps(names))
        for key in (("bn2a_branch1", "randx"), ("res2a", "randx"), ("res2b_branch2c", "randx"), ("res2a_branch2c", "randx")):
            out["pt:%s:%s" % key] = np.stack([points[(b, key[0], key[1])] for b in range(p["nBatches"])])
        out["feat:bn2a_branch1"], out["feat:res2a"] = feats["bn2a_branch1"], feats["res2a"]
        cfgs.alpha = 1e-3
        np.random.seed(79)
        cases = [("res2a_branch2b_relu", "res2a_branch2c", 16), ("res2b_branch2b_relu", "res2b_branch2c", 16)]
        for i, (X_name, Y_name, d_prime) in enumerate(cases):
            resY = net.invBN(quiet(net.appresb, Y_name), Y_name)
            out["resY%d" % i] = np.asarray(resY)
            idxs, W2, B2 = quiet(net.dictionary_kernel, X_name, None, d_prime, Y_name, None)
            out["idxs%d" % i], out["W%d" % i], out["B%d" % i] = np.asarray(idxs, dtype=bool), W2, B2
            out["alpha%d" % i] = float(cfgs.alpha)
        out["cases"] = json.dumps(cases)
        out["rng_next"] = int(np.random.randint(0, 2147483647))
        np.savez_compressed(os.path.join(GOLDEN_DIR, "n03_resnet_residual.npz"), **out)
        print("n03: kept", [int(out["idxs%d" % i].sum()) for i in range(2)],
              "|resY|", [float(np.abs(out["resY%d" % i]).mean()) for i in range(2)])
    finally:
        D.dcfgs.model = cfgs.Models.vgg if hasattr(cfgs.Models, "vgg") else "vgg"
        D.dcfgs.res.short = 0
        D.dcfgs.dic.option = cfgs.pruning_options.prb


RES_KEEP = {"res2a_branch2a": 20, "res2a_branch2b": 16, "res2a_branch2c": 14,
            "res2b_branch2a": 22, "res2b_branch2b": 18, "res2b_branch2c": 16}


def producer_handle(net, blob):
    """what W1keep is called with for the producer of `blob`: its BatchNorm when there is one, else the conv (ReLUs are
    looked through) -- the argument lib/net.py:1547-1569 expects"""
    name = blob
    while name in net.relus:
        name = net.bottom_names[name][0]
    return name


def gen_n04():
    """The bottleneck-by-bottleneck ResNet loop, composed of the REFERENCE's own methods (dictionary_kernel with appresb /
    invBN inside, W1keep, W2keep, select: lib/net.py:1685-1735, 1521-1630): the loop itself is not in the reference
    (component 12), so it is spelled out here and in channel-pruning_amd/lib/net.py::prune_resnet;
    what this golden pins is that the helpers it drives behave like the reference's: WPQ, nonWPQ, bottoms2ch, the live
    parameters after every write-back, alpha and the RNG stream."""
    R, D, cfgs = ref_net_loader.load()
    p = RES_PARAMS
    layers, batches = portable_net.resnet_like(seed=p["seed"], B=p["B"], HW=p["HW"], nBatches=p["nBatches"],
                                               width=p["width"], mid=p["mid"])
    net = ref_net_loader.make_reference_net(layers, batches)
    net.blobs_shape = lambda name: net.net.blobs[name].data.shape      # selector() only passes it on
    set_cfg(D, p)
    D.dcfgs.model = cfgs.Models.resnet
    D.dcfgs.res.short = 1
    D.dcfgs.dic.option = cfgs.pruning_options.resnet
    try:
        names = net.convs + ["bn2a_branch1", "res2a"]
        np.random.seed(9)
        feats, points = quiet(net.extract_features, names=names, save=1)
        quiet(net.load_frozen, feats_dict=feats, points_dict=points)
        net._mem = True
        cfgs.alpha = 1e-3
        np.random.seed(80)
        out = dict(params=json.dumps(p), versions=json.dumps(versions()), names=json.dumps(names), keep=json.dumps(RES_KEEP))
        steps = []
        for blk in net.sums:
            for consumer in [blk + "_branch2" + t for t in "abc"]:
                X_name = net.bottom_names[consumer][0]
                idxs, W2, B2 = quiet(net.dictionary_kernel, X_name, None, RES_KEEP[consumer], consumer, None)
                if consumer.endswith("_branch2a"):
                    quiet(net.select, X_name, consumer, idxs)
                else:
                    quiet(net.W1keep, producer_handle(net, X_name), idxs)
                quiet(net.W2keep, consumer, idxs, W2, B2)
                i = len(steps)
                steps.append([X_name, consumer, RES_KEEP[consumer]])
                out["idxs%d" % i], out["W%d" % i], out["B%d" % i] = np.asarray(idxs, dtype=bool), W2, B2
                out["alpha%d" % i] = float(cfgs.alpha)
        out["steps"] = json.dumps(steps)
        keys = []
        for k, v in net.WPQ.items():
            tag = "%s|%d" % k
            keys.append(tag)
            out["WPQ:" + tag] = np.asarray(v)
        out["wpq_keys"] = json.dumps(keys)
        out["nonwpq_keys"] = json.dumps(list(net.nonWPQ.keys()))
        for k, v in net.nonWPQ.items():
            out["nonWPQ:" + k] = np.asarray(v)
        out["bottoms2ch"] = json.dumps([[a, b] for a, b, _ in net.bottoms2ch])
        out["filters"] = json.dumps(getattr(net.net_param, "filters", []))
        for name in net.convs + net.bns + net.affines:
            out["finalW:" + name] = net.param_data(name).copy()
            out["finalb:" + name] = net.param_b_data(name).copy()
        out["rng_next"] = int(np.random.randint(0, 2147483647))
        np.savez_compressed(os.path.join(GOLDEN_DIR, "n04_resnet_loop.npz"), **out)
        print("n04: kept", [int(out["idxs%d" % i].sum()) for i in range(len(steps))], "WPQ keys", keys,
              "nonWPQ", list(net.nonWPQ.keys()))
    finally:
        D.dcfgs.model = cfgs.Models.vgg if hasattr(cfgs.Models, "vgg") else "vgg"
        D.dcfgs.res.short = 0
        D.dcfgs.dic.option = cfgs.pruning_options.prb


def gen_n05():
    """The reference's combineHP (lib/net.py:1473-1504) on a net that holds decomposed layers conv_V / conv_H / conv_P:
    one chain where 3 m >= 2 o (P is folded into H), one where it is not."""
    R, D, cfgs = ref_net_loader.load()
    rs = np.random.RandomState(21)
    layers = []
    prev = "data"
    shapes = {"conv1_1": (3, 5, 6, 8), "conv1_2": (8, 4, 3, 12)}       # (c, rank of V, m = filters of H, o = filters of P)
    for name, (c, r, m, o) in shapes.items():
        # (the fake forward only knows square kernels: V and H are 3x3 here; combineHP reshapes to (m, -1) whatever k is)
        for suf, W, pad in (("_V", rs.randn(r, c, 3, 3), 1), ("_H", rs.randn(m, r, 3, 3), 1), ("_P", rs.randn(o, m, 1, 1), 0)):
            lname = name + suf
            layers.append(dict(name=lname, type="Convolution", bottom=[prev], top=[lname], W=W.astype(np.float32),
                               b=rs.randn(W.shape[0]).astype(np.float32), pad=pad, stride=1))
            prev = lname
    batches = [rs.randn(2, 3, 12, 12).astype(np.float32) for _ in range(2)]
    net = ref_net_loader.make_reference_net(layers, batches)
    net.save = lambda *a, **k: ("temp/cb_fake.prototxt", "temp/cb_fake.caffemodel")
    out = dict(versions=json.dumps(versions()), shapes=json.dumps(shapes))
    for L in layers:
        out["W:" + L["name"]], out["b:" + L["name"]] = L["W"], L["b"]
    quiet(net.combineHP)
    removed = getattr(net.net_param, "removed", [])
    for L in layers:
        out["newW:" + L["name"]] = net.param_data(L["name"]).copy()
        out["newb:" + L["name"]] = net.param_b_data(L["name"]).copy()
    out["removed"] = json.dumps(removed)
    np.savez_compressed(os.path.join(GOLDEN_DIR, "n05_combine_hp.npz"), **out)
    print("n05: removed", removed, {L["name"]: net.param_data(L["name"]).shape for L in layers if L["name"].endswith("_H")})


if __name__ == "__main__":
    which = sys.argv[1:] or ["n01", "n02", "n03", "n04", "n05"]
    for w in which:
        {"n01": gen_n01, "n02": gen_n02, "n03": gen_n03, "n04": gen_n04, "n05": gen_n05}[w]()
