"""CPU: the host-side rows of the caffe-free Net facade against what the UNMODIFIED reference lib/net.py produced
(goldens n01 / n03, oracle/gen_golden_net.py) -- sampling, the frozen-feature pickle, the ResNet residual term -- and the
torch graph provider against the bit-portable forward pass."""
import json
import os
import pickle

import numpy as np

from conftest import GOLDEN_DIR


def _vgg(p):
    import portable_net
    from lib.net import Net
    from portable_provider import PortableProvider
    layers, batches = portable_net.vgg_like(seed=p["seed"], chans=[tuple(c) for c in p["chans"]], B=p["B"], HW=p["HW"],
                                            nBatches=p["nBatches"])
    return Net(None, PortableProvider(layers, batches), nBatches=p["nBatches"], nPointsPerLayer=p["nPoints"], graph=layers), layers, batches


def test_frozen_pickle_is_the_reference_format(tmp_path):
    """extract_features + freeze_images (lib/net.py:368-532, 749-802): same RNG consumption, same keys -- "nPointsPerLayer",
    "nBatches", "data", "label", (batch, 0), (batch, 1), (batch, name, "randx"/"randy") -- and the same values as the
    pickle the reference wrote; load_frozen reads the reference's file."""
    g = np.load(os.path.join(GOLDEN_DIR, "n01_vgg_pruning.npz"))
    p = json.loads(str(g["params"]))
    net, _, _ = _vgg(p)
    np.random.seed(3)
    path = net.freeze_images(path=str(tmp_path / "frozen.pickle"), convs=net.convs)
    assert int(np.random.randint(0, 2147483647)) == int(g["rng_after_freeze"])
    feats, points = pickle.load(open(path, "rb"))
    rfeats, rpoints = pickle.load(open(os.path.join(GOLDEN_DIR, "n01_frozen.pickle"), "rb"))
    assert set(points.keys()) == set(rpoints.keys()) and set(feats.keys()) == set(rfeats.keys())
    for k in rpoints:
        assert np.array_equal(np.asarray(points[k]), np.asarray(rpoints[k])), k
    for k in rfeats:
        assert feats[k].dtype == np.float64 and np.array_equal(feats[k], rfeats[k]), k
    net2, _, _ = _vgg(p)
    net2.load_frozen(path=os.path.join(GOLDEN_DIR, "n01_frozen.pickle"))
    assert net2._mem and np.array_equal(net2.provider.batches[3], rpoints[(3, 0)])


def test_resnet_shortcut_points_and_residual_match_reference():
    """extract_features' point sharing for shortcut blobs (net.py:466-487) and appresb + invBN (net.py:1641-1683,
    1200-1217): identical to the reference's numbers."""
    import lib.cfgs as cfgs
    import portable_net
    from lib.cfgs import c as dcfgs
    from lib.net import Net
    from portable_provider import PortableProvider
    g = np.load(os.path.join(GOLDEN_DIR, "n03_resnet_residual.npz"))
    p = json.loads(str(g["params"]))
    layers, batches = portable_net.resnet_like(seed=p["seed"], B=p["B"], HW=p["HW"], nBatches=p["nBatches"], width=p["width"],
                                               mid=p["mid"])
    net = Net(None, PortableProvider(layers, batches), nBatches=p["nBatches"], nPointsPerLayer=p["nPoints"], graph=layers,
              model=cfgs.Models.resnet)
    dcfgs.model, dcfgs.res.short, dcfgs.dic.option = cfgs.Models.resnet, 1, cfgs.pruning_options.resnet
    try:
        np.random.seed(9)
        feats, points = net.extract_features(names=json.loads(str(g["names"])), save=1)
        for key in ("bn2a_branch1", "res2a", "res2b_branch2c", "res2a_branch2c"):
            got = np.stack([points[(b, key, "randx")] for b in range(p["nBatches"])])
            assert np.array_equal(got, g["pt:%s:randx" % key]), key
        assert np.array_equal(feats["bn2a_branch1"], g["feat:bn2a_branch1"]) and np.array_equal(feats["res2a"], g["feat:res2a"])
        net.load_frozen(feats_dict=feats, points_dict=points)
        assert net.appresb("res2a_branch2c").max() == 0          # nothing drifted yet
        net.set_param_data("conv1", g["conv1_W"])
        for i, (_, Y_name, _) in enumerate(json.loads(str(g["cases"]))):
            resY = net.invBN(net.appresb(Y_name), Y_name)
            assert np.abs(resY).max() > 0 and np.array_equal(resY, g["resY%d" % i])
        dcfgs.res.short = 0
        assert net.appresb("res2a_branch2c") == 0
    finally:
        dcfgs.model, dcfgs.res.short, dcfgs.dic.option = '', 0, cfgs.pruning_options.prb


def test_torch_graph_provider_matches_portable_forward():
    import portable_net
    from lib.net import Net
    from lib.provider import TorchGraphProvider
    layers, batches = portable_net.resnet_like(seed=4, B=3, HW=10, nBatches=2, width=12, mid=8)
    net = Net(None, TorchGraphProvider(layers, batches, num_threads=1), nBatches=2, nPointsPerLayer=3, graph=layers)
    ref = portable_net.forward(layers, batches[1])
    got = net.forward(1)
    for name, v in ref.items():
        assert got[name].shape == v.shape and np.abs(got[name] - v).max() <= 1e-4 * max(1.0, np.abs(v).max()), name
    W = net.param_data("res2a_branch2a").copy()
    W[:2] = 0
    net.set_param_data("res2a_branch2a", W)                       # live: the blobs follow the net's weights
    assert np.abs(net.forward(1)["res2a_branch2a"][:, :2] - net.param_b_data("res2a_branch2a")[:2][None, :, None, None]).max() <= 1e-6


def test_w1keep_w2keep_select_combinehp_bookkeeping():
    """The write-back helpers of the layer-by-layer drivers (net.py:1521-1630, 1473-1504) on the facade's data model."""
    import portable_net
    from lib.net import Net
    from portable_provider import PortableProvider
    layers, batches = portable_net.resnet_like(seed=4, B=2, HW=8, nBatches=1, width=12, mid=8)
    net = Net(None, PortableProvider(layers, batches), nBatches=1, nPointsPerLayer=2, graph=layers, model="resnet")
    idxs = np.zeros(8, dtype=bool)
    idxs[[0, 2, 5]] = True
    W0, b0 = net.param_data("res2a_branch2b").copy(), net.param_b_data("res2a_branch2b").copy()
    k0 = net.param_data("scale2a_branch2b").copy()
    net.W1keep("res2a_branch2b", idxs)                            # producer keeps 3 filters (+ its BatchNorm / Scale rows)
    assert np.array_equal(net.WPQ[("res2a_branch2b", 0)], W0[idxs]) and np.array_equal(net.WPQ[("res2a_branch2b", 1)], b0[idxs])
    assert np.array_equal(net.WPQ[("scale2a_branch2b", 0)], k0[idxs]) and ("bn2a_branch2b", 1) in net.WPQ
    assert np.all(net.param_data("res2a_branch2b")[~idxs] == 0) and np.all(net.param_data("scale2a_branch2b")[~idxs] == 0)
    assert net.num_output["res2a_branch2b"] == 3
    Wc = net.param_data("res2a_branch2c").copy()
    W2 = np.random.RandomState(0).randn(Wc.shape[0], 3, 1, 1)
    bold = net.param_b_data("res2a_branch2c").copy()
    net.W2keep("res2a_branch2c", idxs, W2, B2=np.ones(Wc.shape[0]))
    assert np.allclose(net.param_data("res2a_branch2c")[:, idxs], W2) and np.all(net.param_data("res2a_branch2c")[:, ~idxs] == 0)
    assert np.allclose(net.WPQ[("res2a_branch2c", 1)], 1 + bold)
    net.W1keep("res2a", np.ones(12, dtype=bool))                  # a sum blob: deferred (bottoms2ch), as in the reference
    assert net.bottoms2ch and net.bottoms2ch[-1][0] == "res2a"
    fname = net.select("res2a", "res2b_branch2a", np.arange(12) % 2 == 0)
    assert net.nonWPQ[fname].sum() == 6 and net.layer_bottom("res2b_branch2a") == fname
    # combineHP: P (o x m) folded into H (m x r x 1 x k) when 3 m >= 2 o
    rs = np.random.RandomState(1)
    net.WPQ = {("convA_H", 0): rs.randn(6, 4, 1, 3), ("convA_H", 1): rs.randn(6), ("convA_P", 0): rs.randn(8, 6, 1, 1),
               ("convA_P", 1): rs.randn(8), ("convB_H", 0): rs.randn(2, 4, 1, 3), ("convB_H", 1): rs.randn(2),
               ("convB_P", 0): rs.randn(8, 2, 1, 1), ("convB_P", 1): rs.randn(8)}
    ref_w = np.einsum("om,mrhw->orhw", net.WPQ[("convA_P", 0)][:, :, 0, 0], net.WPQ[("convA_H", 0)])
    ref_b = net.WPQ[("convA_P", 1)] + net.WPQ[("convA_P", 0)][:, :, 0, 0] @ net.WPQ[("convA_H", 1)]
    assert net.combineHP() == ["convA_H"]
    assert np.allclose(net.WPQ[("convA_H", 0)], ref_w) and np.allclose(net.WPQ[("convA_H", 1)], ref_b)
    assert ("convA_P", 0) not in net.WPQ and ("convB_P", 0) in net.WPQ and net.removed == ["convA_P"]


def test_provider_without_data_blob_freezes_features_and_points_only(tmp_path):
    """The provider contract is batch -> {blob name: array}; the input images are optional.  Without a "data" blob
    freeze_images() / dictionary_kernel()'s non-frozen path store no (batch, 0) / (batch, 1) / "data" / "label" entries
    (same features, same sample points, same RNG consumption) and load_frozen() leaves the provider's batches alone."""
    g = np.load(os.path.join(GOLDEN_DIR, "n01_vgg_pruning.npz"))
    p = json.loads(str(g["params"]))
    net, _, _ = _vgg(p)
    inner = net.provider

    class NoData:
        batches = inner.batches

        def __call__(self, batch, *a):
            return {k: v for k, v in inner(batch, *a).items() if k not in ("data", "label")}

    net.provider = NoData()
    net._blob_cache = (None, None)
    np.random.seed(3)
    path = net.freeze_images(path=str(tmp_path / "frozen.pickle"), convs=net.convs)
    assert int(np.random.randint(0, 2147483647)) == int(g["rng_after_freeze"])
    feats, points = pickle.load(open(path, "rb"))
    rfeats, rpoints = pickle.load(open(os.path.join(GOLDEN_DIR, "n01_frozen.pickle"), "rb"))
    dropped = {"data", "label"} | {(b, i) for b in range(p["nBatches"]) for i in (0, 1)}
    assert set(points.keys()) == set(rpoints.keys()) - dropped
    for k in points:
        assert np.array_equal(np.asarray(points[k]), np.asarray(rpoints[k])), k
    for k in rfeats:
        assert np.array_equal(feats[k], rfeats[k]), k
    assert net._mem


def _resnet(p, model=None):
    import lib.cfgs as cfgs
    import portable_net
    from lib.net import Net
    from portable_provider import PortableProvider
    layers, batches = portable_net.resnet_like(seed=p["seed"], B=p["B"], HW=p["HW"], nBatches=p["nBatches"], width=p["width"],
                                               mid=p["mid"])
    return Net(None, PortableProvider(layers, batches), nBatches=p["nBatches"], nPointsPerLayer=p["nPoints"], graph=layers,
               model=cfgs.Models.resnet)


def test_write_back_helpers_match_the_reference_on_the_resnet_loop():
    """W1keep / W2keep / select (lib/net.py:1521-1630) against the REFERENCE's own methods: golden n04 ran them through
    oracle/ref_net_loader.py on the portable ResNet, step by step; fed with the same (idxs, W2, B2) per step the facade must
    leave the same WPQ (keys, order, values), nonWPQ, bottoms2ch and live parameters -- bit for bit."""
    import lib.cfgs as cfgs
    from lib.cfgs import c as dcfgs
    g = np.load(os.path.join(GOLDEN_DIR, "n04_resnet_loop.npz"))
    p = json.loads(str(g["params"]))
    net = _resnet(p)
    net._mem = True
    saved = (dcfgs.model, dcfgs.res.short, dcfgs.dic.option)
    dcfgs.model, dcfgs.res.short, dcfgs.dic.option = cfgs.Models.resnet, 1, cfgs.pruning_options.resnet
    try:
        for i, (X_name, consumer, d_prime) in enumerate(json.loads(str(g["steps"]))):
            assert X_name == net.bottom_names[consumer][0]
            idxs, W2, B2 = g["idxs%d" % i], g["W%d" % i], g["B%d" % i]
            if consumer.endswith("_branch2a"):
                net.select(X_name, consumer, idxs)
            else:
                net.W1keep(net._producer_handle(X_name), idxs)
            net.W2keep(consumer, idxs, W2, B2)
    finally:
        dcfgs.model, dcfgs.res.short, dcfgs.dic.option = saved
    keys = json.loads(str(g["wpq_keys"]))
    assert ["%s|%d" % k for k in net.WPQ.keys()] == keys
    for tag in keys:
        name, idx = tag.split("|")
        got, ref = np.asarray(net.WPQ[(name, int(idx))]), g["WPQ:" + tag]
        assert got.shape == ref.shape and np.array_equal(got, ref), tag
    assert list(net.nonWPQ.keys()) == json.loads(str(g["nonwpq_keys"]))
    for k in net.nonWPQ:
        assert np.array_equal(net.nonWPQ[k], g["nonWPQ:" + k])
    assert [[a, b] for a, b, _ in net.bottoms2ch] == json.loads(str(g["bottoms2ch"]))
    for name in net.convs + net.bns + net.affines:
        assert np.array_equal(net.param_data(name), g["finalW:" + name]), name
        assert np.array_equal(net.param_b_data(name), g["finalb:" + name]), name
    # the samplers re-wire their consumers (lib/builder.py:666-672)
    for fname, bottom, top, kept in json.loads(str(g["filters"])):
        assert net.layer_bottom(top) == fname and int(net.nonWPQ[fname].sum()) == kept


def test_combine_hp_matches_the_reference():
    """combineHP (lib/net.py:1473-1504) against the reference's own method (golden n05): conv_P folded into conv_H where
    3 m >= 2 o, left alone otherwise; merged weights, bias and the list of removed layers identical."""
    from lib.net import ConvSpec, Net
    g = np.load(os.path.join(GOLDEN_DIR, "n05_combine_hp.npz"))
    shapes = json.loads(str(g["shapes"]))
    convs = [ConvSpec(n, np.zeros((2, 3, 3, 3), np.float32), np.zeros(2, np.float32), "data") for n in shapes]
    net = Net(convs, lambda b: {}, nBatches=1, nPointsPerLayer=1)
    for n in shapes:
        for suf in ("_H", "_P"):
            net.WPQ[(n + suf, 0)] = g["W:" + n + suf].copy()
            net.WPQ[(n + suf, 1)] = g["b:" + n + suf].copy()
        net.WPQ[n + "_V"] = g["W:" + n + "_V"].copy()
    merged = net.combineHP()
    removed = json.loads(str(g["removed"]))
    assert net.removed == removed and merged == [r[:-2] + "_H" for r in removed]
    for n in shapes:
        h = n + "_H"
        assert np.array_equal(np.asarray(net.WPQ[(h, 0)]), g["newW:" + h]) and np.array_equal(np.asarray(net.WPQ[(h, 1)]), g["newb:" + h])
        if n + "_P" in removed:
            assert (n + "_P", 0) not in net.WPQ and net.num_output[h] == g["newW:" + h].shape[0]
        else:
            assert np.array_equal(net.WPQ[(n + "_P", 0)], g["W:" + n + "_P"])
