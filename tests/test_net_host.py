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
