"""GPU: the caffe-free Net facade (lib/net.py) -- extract_XY, dictionary_kernel, R3 -- on a small
VGG-shaped network whose activations come from a torch CPU forward (standing in for Caffe)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_net(seed=0, B=4, HW=16, nBatches=6, nPoints=5):
    import torch
    import torch.nn.functional as F
    from lib.net import ConvSpec, Net
    torch.set_num_threads(1)
    rs = np.random.RandomState(seed)
    chans = [("conv1_1", 3, 16), ("conv1_2", 16, 16), ("conv2_1", 16, 32), ("conv2_2", 32, 32), ("conv3_1", 32, 48)]
    bottoms = {"conv1_1": "data", "conv1_2": "conv1_1_relu", "conv2_1": "pool1", "conv2_2": "conv2_1_relu",
               "conv3_1": "pool2"}
    specs = []
    for name, cin, cout in chans:
        W = (rs.randn(cout, cin, 3, 3) * (1.5 / np.sqrt(cin * 9))).astype(np.float32)
        b = (rs.randn(cout) * 0.1).astype(np.float32)
        specs.append(ConvSpec(name, W, b, bottoms[name], pad=1, stride=1))
    data = [rs.randn(B, 3, HW, HW).astype(np.float32) for _ in range(nBatches)]
    orig = {s.name: (s.W.copy(), s.b.copy()) for s in specs}

    def provider(batch):
        x = torch.from_numpy(data[batch])
        blobs = {"data": data[batch]}
        for name, _, _ in chans:
            W, b = orig[name]
            inp = torch.from_numpy(blobs[bottoms[name]])
            y = F.conv2d(inp, torch.from_numpy(W), torch.from_numpy(b), padding=1)
            blobs[name] = y.numpy()
            blobs[name + "_relu"] = F.relu(y).numpy()
            if name == "conv1_2":
                blobs["pool1"] = F.max_pool2d(F.relu(y), 2).numpy()
            if name == "conv2_2":
                blobs["pool2"] = F.max_pool2d(F.relu(y), 2).numpy()
        del x
        return blobs

    net = Net(specs, provider, nBatches=nBatches, nPointsPerLayer=nPoints)
    return net, orig


def test_extract_xy_layout_and_debug_invariant(ctx):
    """relu(patch) . W2^T + b2 == sampled response of the consumer (the reference's own DEBUG
    check, net.py:659-679, CHECK_EQ 1e-4), and the [N*k*k, C] layout equals the C restatement."""
    import cp_oracle
    from lib.utils import CHECK_EQ
    net, _ = make_net()
    np.random.seed(3)
    net.freeze_images(convs=net.convs)
    for X_name, Y_name in (("conv1_1", "conv1_2"), ("pool1", "conv2_1"), ("conv2_1", "conv2_2")):
        X = net.extract_XY(X_name, Y_name)
        k = 3
        C = X.shape[1]
        newX = np.rollaxis(X.reshape((-1, k, k, C)), 3, 1)
        W2, b2 = net.param_data(Y_name), net.param_b_data(Y_name)
        fake = np.maximum(newX, 0).reshape(newX.shape[0], -1) @ W2.reshape(W2.shape[0], -1).T.astype(np.float64) + b2
        CHECK_EQ(fake, net._feats_dict[Y_name])
        # against the oracle gather, batch by batch
        nP = net._points_dict["nPointsPerLayer"]
        rows = []
        for batch in range(net._points_dict["nBatches"]):
            blob = net.forward(batch)[X_name]
            rows.append(cp_oracle.patch_gather(blob, net._points_dict[(batch, Y_name, "randx")],
                                               net._points_dict[(batch, Y_name, "randy")], k, 1, 1, 0))
        ref = np.concatenate(rows, 0)
        assert ref.shape[0] == nP * blob.shape[0] * net._points_dict["nBatches"]
        assert np.array_equal(newX, ref.astype(np.float64))


def test_R3_prunes_the_reference_layer_pairs_and_matches_oracle(ctx):
    import cp_oracle
    import lib.cfgs as cfgs
    net, orig = make_net(seed=1)
    np.random.seed(5)
    net.freeze_images(convs=net.convs)
    feats = {k: v.copy() for k, v in net._feats_dict.items()}
    cfgs.alpha = 1e-3
    np.random.seed(77)
    rankdic = {"conv1_2": 8, "conv2_1": 16, "conv2_2": 16}
    WPQ, new_pt = net.R3(rankdic=rankdic)
    assert new_pt.startswith("3C4x")
    pairs = [("conv1_2", "conv2_1", "pool1"), ("conv2_1", "conv2_2", "conv2_1"), ("conv2_2", "conv3_1", "pool2")]
    assert sorted(net.selection) == sorted(p[1] for p in pairs)
    # replay with the oracle: same RNG stream, same alpha carry, same operands
    net2, _ = make_net(seed=1)
    net2.load_frozen(feats_dict=feats, points_dict=net._points_dict)
    alpha = 1e-3
    np.random.seed(77)
    expected = {}
    for conv, convnext, X_name in pairs:
        c_out = orig[conv][0].shape[0]
        d_c = max(int(c_out / 1.15), rankdic[conv])
        X = net2.extract_XY(X_name, convnext)
        newX = np.maximum(np.rollaxis(X.reshape((-1, 3, 3, X.shape[1])), 3, 1), 0)
        W2, b2 = orig[convnext]
        Y = feats[convnext] - b2
        idxs, nW, nB, alpha = cp_oracle.dictionary_oracle(newX, W2, Y, d_c, b2, alpha_in=alpha, lasso="c_gram",
                                                          ls="numpy")
        assert np.array_equal(net.selection[convnext], idxs)
        expected[convnext] = nW                                 # consumer: compact input channels
        expected[conv] = expected.get(conv, orig[conv][0].astype(np.float64))[idxs]   # producer: kept filters
        assert WPQ[(conv, 1)].shape[0] == int(idxs.sum())
    for name, W in expected.items():
        assert WPQ[(name, 0)].shape == W.shape
        assert np.linalg.norm(WPQ[(name, 0)] - W) <= 1e-5 * np.linalg.norm(W)
    assert cfgs.alpha == alpha


# ---------------------------------------------------------------------------------------------
# the whole 3C loop (spatial decomposition -> channel decomposition -> channel pruning per conv)
# ---------------------------------------------------------------------------------------------
CHANS_3C = [("conv1_1", 3, 12), ("conv1_2", 12, 12), ("conv2_1", 12, 16), ("conv2_2", 16, 16), ("conv3_1", 16, 24)]
BOTTOMS_3C = {"conv1_1": "data", "conv1_2": "conv1_1_relu", "conv2_1": "pool1", "conv2_2": "conv2_1_relu",
              "conv3_1": "pool2"}


def _wire(blobs, name, y):
    """ReLU / pooling that follow conv `name` in the little VGG used here"""
    import torch.nn.functional as F
    blobs[name] = y.numpy()
    blobs[name + "_relu"] = F.relu(y).numpy()
    if name == "conv1_2":
        blobs["pool1"] = F.max_pool2d(F.relu(y), 2).numpy()
    if name == "conv2_2":
        blobs["pool2"] = F.max_pool2d(F.relu(y), 2).numpy()


def make_live_net(seed=0, B=4, HW=12, nBatches=10, nPoints=8):
    """lib/provider.py::TorchSequentialProvider: torch CPU forward with the net's CURRENT weights (what the
    reference's Caffe net does)"""
    from lib.net import ConvSpec, Net
    from lib.provider import TorchSequentialProvider
    rs = np.random.RandomState(seed)
    specs = []
    for name, cin, cout in CHANS_3C:
        W = (rs.randn(cout, cin, 3, 3) * (1.5 / np.sqrt(cin * 9))).astype(np.float32)
        b = (rs.randn(cout) * 0.1).astype(np.float32)
        specs.append(ConvSpec(name, W, b, BOTTOMS_3C[name], pad=1, stride=1))
    data = [rs.randn(B, 3, HW, HW).astype(np.float32) for _ in range(nBatches)]
    # one thread: tiny convolutions, the intra-op thread pool of a many-core host costs 30 ms per call
    provider = TorchSequentialProvider(data, pools={"conv1_2": ("pool1", 2, 2), "conv2_2": ("pool2", 2, 2)}, num_threads=1)
    return Net(specs, provider, nBatches=nBatches, nPointsPerLayer=nPoints), data


def _emitted_forward(layers, x):
    import torch
    import torch.nn.functional as F
    blobs = {"data": x}
    for L in layers:
        y = F.conv2d(torch.from_numpy(blobs[L["bottom"]]), torch.from_numpy(L["W"]),
                     None if L["b"] is None else torch.from_numpy(L["b"]), padding=L["pad"], stride=L["stride"])
        if L["top"] in BOTTOMS_3C:
            _wire(blobs, L["top"], y)
        else:
            blobs[L["top"]] = y.numpy()
    return blobs


def test_R3_full_3C_loop_matches_oracle_replay(ctx):
    """Net.R3 with a live provider = the reference's loop (net.py:1339-1470): per conv VH_decompose (+ ReLU-aware
    refit), ITQ_decompose on the features of the modified net, dictionary() on its outputs.  Replayed with the CPU
    restatements on a second net: channel selections identical, every layer's final k x k weights / bias equal (they
    are stored as float32 like Caffe blobs), WPQ has the reference's keys and shapes, and the emitted V -> H -> P
    network computes what the net with the k x k weights computes."""
    import cp_oracle
    import lib.cfgs as cfgs
    from lib.utils import underline
    rankdic = {"conv1_2": 6, "conv2_1": 8, "conv2_2": 8, "conv3_1": 12}
    net, data = make_live_net(seed=2)
    np.random.seed(11)
    net.freeze_images(convs=net.convs)
    feats = {k: v.copy() for k, v in net._feats_dict.items()}
    cfgs.alpha = 1e-3
    np.random.seed(99)
    WPQ, new_pt = net.R3(rankdic=rankdic)
    assert new_pt.startswith("3C4x") and net._decomposed

    # ---- replay with the oracle on a second, identical net ----
    net2, _ = make_live_net(seed=2)
    net2.load_frozen(feats_dict={k: v.copy() for k, v in feats.items()}, points_dict=net._points_dict)
    alpha = 1e-3
    np.random.seed(99)
    selection, shapes = {}, {}
    convs = net2.convs

    def set_conv(c, d):
        if c in selection:
            Wc = net2.param_data(c).copy()
            Wc[:, selection[c]] = d
            net2.set_param_data(c, Wc)
        else:
            net2.set_param_data(c, d)

    for conv, convnext in zip(convs[1:], convs[2:] + ["pool5"]):
        n_out = net2.param_shape(conv)[0]
        rank = rankdic[conv]
        d_c = max(int(n_out / 1.15), rank)
        weights = net2.param_data(conv).astype(np.float64)
        x = net2.extract_XY(net2.bottom_names[conv][0], conv)
        X = np.rollaxis(x.reshape((-1, 3, 3, x.shape[1])), 3, 1).copy()
        if conv in selection:
            weights, X = weights[:, selection[conv]], X[:, selection[conv]]
        Y = feats[conv] - net2.param_b_data(conv)
        V, H, VHr, b = cp_oracle.vh_decompose_oracle(weights, rank, X, Y)
        net2.set_param_b(conv, b)
        set_conv(conv, VHr)
        cur, _ = net2.extract_features(names=conv, points_dict=net2._points_dict, save=1)
        W1, W2, B, W12 = cp_oracle.itq_decompose_oracle(cur[conv], feats[conv], H, rank, bias=net2.param_b_data(conv),
                                                        Wr=VHr)
        set_conv(conv, W12)
        net2.set_param_b(conv, B)
        shapes[underline(conv, "V")] = V.shape
        shapes[(underline(conv, "H"), 0)] = (rank, H.shape[1], H.shape[2], H.shape[3])
        shapes[(underline(conv, "P"), 0)] = (n_out, rank, 1, 1)
        if convnext in convs:
            X_name = net2.bottom_names[convnext][0] if conv in ("conv1_2", "conv2_2") else conv
            x = net2.extract_XY(X_name, convnext)
            newX = np.maximum(np.rollaxis(x.reshape((-1, 3, 3, x.shape[1])), 3, 1), 0)
            W2n, b2n = net2.param_data(convnext), net2.param_b_data(convnext)
            idxs, nW, nB, alpha = cp_oracle.dictionary_oracle(newX, W2n, feats[convnext] - b2n, d_c, b2n, alpha_in=alpha,
                                                              lasso="c_gram", ls="numpy")
            selection[convnext] = idxs
            Wn = net2.param_data(convnext).copy()
            Wn[:, ~idxs] = 0
            Wn[:, idxs] = nW
            net2.set_param_data(convnext, Wn)
            net2.set_param_b(convnext, nB)
            shapes[(underline(conv, "P"), 0)] = (int(idxs.sum()), rank, 1, 1)

    assert sorted(net.selection) == sorted(selection)
    for name in selection:
        assert np.array_equal(net.selection[name], selection[name]), name
    assert cfgs.alpha == alpha
    for name in convs:
        Wa, Wb = net.param_data(name).astype(np.float64), net2.param_data(name).astype(np.float64)
        assert np.linalg.norm(Wa - Wb) <= 2e-4 * np.linalg.norm(Wb), name   # f32 weight storage between the steps
        ba, bb = net.param_b_data(name).astype(np.float64), net2.param_b_data(name).astype(np.float64)
        assert np.linalg.norm(ba - bb) <= 2e-4 * max(1.0, np.linalg.norm(bb)), name
    for key, shp in shapes.items():
        assert tuple(WPQ[key].shape) == tuple(shp), key
    # ---- the emitted network is the same function as the net with the final k x k weights ----
    layers = net.emit_layers()
    assert [L["name"] for L in layers] == ["conv1_1"] + [underline(c, s) for c in convs[1:] for s in "VHP"]
    for batch in (0, 3):
        full = net.forward(batch)
        small = _emitted_forward(layers, data[batch])
        for name in ("conv2_1", "conv3_1"):     # a pruned producer emits only the channels its consumer kept
            a, b_ = small[name], full[name]
            nxt = convs[convs.index(name) + 1] if convs.index(name) + 1 < len(convs) else None
            kept = net.selection[nxt] if nxt in net.selection else np.ones(b_.shape[1], dtype=bool)
            assert a.shape[1] == int(kept.sum())
            assert np.abs(a - b_[:, kept]).max() <= 2e-4 * np.abs(b_).max(), name


def test_R3_decomposition_needs_a_live_provider(ctx):
    net, _ = make_net()
    np.random.seed(3)
    net.freeze_images(convs=net.convs)
    with pytest.raises(ValueError, match="live"):
        net.R3(rankdic={"conv1_2": 8, "conv2_1": 16, "conv2_2": 16}, decompose=True)


# ---------------------------------------------------------------------------------------------
# against the UNMODIFIED reference lib/net.py (goldens n01-n03, oracle/gen_golden_net.py: fake pycaffe net over
# oracle/portable_net.py; the same bit-portable forward pass feeds the facade here)
# ---------------------------------------------------------------------------------------------
import json
import os
import pickle

from conftest import GOLDEN_DIR


def _rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _portable_vgg(p):
    import portable_net
    from lib.net import Net
    from portable_provider import PortableProvider
    layers, batches = portable_net.vgg_like(seed=p["seed"], chans=[tuple(c) for c in p["chans"]], B=p["B"], HW=p["HW"],
                                            nBatches=p["nBatches"])
    return Net(None, PortableProvider(layers, batches), nBatches=p["nBatches"], nPointsPerLayer=p["nPoints"], graph=layers)


def test_net_rows_match_reference_net_py_vgg(ctx, tmp_path):
    """a1 extract_XY, a2 dictionary_kernel, a7 alpha carry, f3 frozen pickle: identical points / features / patches / RNG
    consumption / masks / alpha, weights <= 1e-5, against what /root/reference/lib/net.py itself produced."""
    import lib.cfgs as cfgs
    g = np.load(os.path.join(GOLDEN_DIR, "n01_vgg_pruning.npz"))
    p = json.loads(str(g["params"]))
    net = _portable_vgg(p)
    np.random.seed(3)
    path = net.freeze_images(path=str(tmp_path / "frozen.pickle"), convs=net.convs)
    assert int(np.random.randint(0, 2147483647)) == int(g["rng_after_freeze"])
    with open(path, "rb") as f:
        feats, points = pickle.load(f)
    with open(os.path.join(GOLDEN_DIR, "n01_frozen.pickle"), "rb") as f:
        rfeats, rpoints = pickle.load(f)
    assert set(points.keys()) == set(rpoints.keys()) and set(feats.keys()) == set(rfeats.keys())
    for k in rpoints:
        assert np.array_equal(np.asarray(points[k]), np.asarray(rpoints[k])), k
    for k in rfeats:
        assert feats[k].dtype == rfeats[k].dtype and np.array_equal(feats[k], rfeats[k]), k
    cfgs.alpha = 1e-3
    np.random.seed(77)
    for i, (X_name, Y_name, d_prime) in enumerate(json.loads(str(g["pairs"]))):
        X = net.extract_XY(X_name, Y_name)
        assert X.dtype == np.float64 and np.array_equal(X, g["xy%d" % i].astype(np.float64))
        idxs, W2, B2 = net.dictionary_kernel(X_name, None, d_prime, Y_name, None)
        assert np.array_equal(idxs, g["idxs%d" % i])
        assert _rel(W2, g["W%d" % i]) <= 1e-5 and _rel(B2, g["B%d" % i]) <= 1e-5
        assert cfgs.alpha == float(g["alpha%d" % i])
    assert int(np.random.randint(0, 2147483647)) == int(g["rng_next"])


def test_load_frozen_reads_the_reference_pickle(ctx):
    """The reference's own frozen<nBatches>.pickle drives the facade: images (batch, 0) are handed to the provider,
    features and points are adopted; the pruning of a pair then reproduces the reference's result."""
    import lib.cfgs as cfgs
    g = np.load(os.path.join(GOLDEN_DIR, "n01_vgg_pruning.npz"))
    p = json.loads(str(g["params"]))
    net = _portable_vgg(p)
    net.provider.set_batches([np.zeros_like(b) for b in net.provider.batches])      # would give wrong patches
    net.load_frozen(path=os.path.join(GOLDEN_DIR, "n01_frozen.pickle"))
    cfgs.alpha = 1e-3
    np.random.seed(77)
    X_name, Y_name, d_prime = json.loads(str(g["pairs"]))[0]
    idxs, W2, B2 = net.dictionary_kernel(X_name, None, d_prime, Y_name, None)
    assert np.array_equal(idxs, g["idxs0"]) and _rel(W2, g["W0"]) <= 1e-5


def _composite(WPQ, conv):
    """k x k weights the chain conv_V -> conv_H -> conv_P computes: [n, c, kh, kw]"""
    from lib.utils import underline
    V = np.asarray(WPQ[underline(conv, "V")], dtype=np.float64)            # [r, c, k, 1]
    H = np.asarray(WPQ[(underline(conv, "H"), 0)], dtype=np.float64)       # [d, r, 1, k]
    P = np.asarray(WPQ[(underline(conv, "P"), 0)], dtype=np.float64)       # [n, d, 1, 1]
    HV = np.einsum("drw,rch->dchw", H[:, :, 0, :], V[:, :, :, 0])
    return np.einsum("nd,dchw->nchw", P[:, :, 0, 0], HV)


def test_R3_3C_loop_matches_reference_net_py(ctx):
    """f4 / a8: Net.R3() -- VH -> ITQ -> pruning per conv -- against the reference's own R3 (insert / set_conv / save_pt
    stubbed there): WPQ with the reference's keys and shapes, the alpha carry, the RNG consumption, and -- as far as the
    loop is a well-posed function of its inputs -- selections and weights.

    How far that is was measured on the reference itself (oracle/gen_golden_net.py's net, frozen features perturbed by
    1e-12 relative): final weights move by 3e-10 (conv1_2), 8e-6 (conv2_1), 1e-2 (conv2_2), 0.67 (conv3_1) and one
    mask bit of the last pair flips -- 50 + 50 alternations per conv amplify rounding by ~1e3 per layer.  On top, the
    reference factors the float32 Caffe weights with a SINGLE-precision LAPACK SVD (scipy gesvd on a float32 array,
    decompose.py:45-47, 100; net.py:1353), the device in float64: the first layer agrees to 1e-4, not 1e-13 (device vs
    the float64 CPU restatement on identical inputs: 3e-14, tests/tools/r3_step_diag.py).  Hence: exact selections
    for the first two pairs, weights of the first two decomposed convs within the budget that noise leaves, at most
    two differing mask bits on the last pair."""
    import lib.cfgs as cfgs
    from lib.cfgs import c as dcfgs
    from lib.utils import underline
    g = np.load(os.path.join(GOLDEN_DIR, "n02_vgg_r3_3c.npz"))
    p = json.loads(str(g["params"]))
    net = _portable_vgg(p)
    np.random.seed(5)
    feats, points = net.extract_features(names=net.convs, save=1)
    net.load_frozen(feats_dict=feats, points_dict=points)
    cfgs.alpha = 1e-3
    dcfgs.dic.keep, dcfgs.dic.vh = 3., 1
    np.random.seed(78)
    WPQ, new_pt = net.R3()
    sel_keys = json.loads(str(g["sel_keys"]))
    assert sorted(net.selection) == sorted(sel_keys)
    for k in sel_keys[:2]:
        assert np.array_equal(net.selection[k], g["sel:" + k]), k
    assert int((net.selection[sel_keys[2]] != g["sel:" + sel_keys[2]]).sum()) <= 2
    ref = {}
    for tag in json.loads(str(g["wpq_keys"])):
        key = tag if "|" not in tag else (tag.split("|")[0], int(tag.split("|")[1]))
        ref[key] = g["WPQ:" + tag]
    assert set(WPQ.keys()) == set(ref.keys())
    for key, v in ref.items():
        name = key if isinstance(key, str) else key[0]
        shp, rshp = tuple(np.asarray(WPQ[key]).shape), tuple(v.shape)
        if name.startswith("conv3_1") or name.startswith("conv2_2_P"):   # sized by the last pair's kept channels
            assert len(shp) == len(rshp) and all(abs(a_ - b_) <= 2 for a_, b_ in zip(shp, rshp)), key
        else:
            assert shp == rshp, key
    for conv, tol in (("conv1_2", 1e-3), ("conv2_1", 2e-2)):
        assert _rel(_composite(WPQ, conv), _composite(ref, conv)) <= tol, conv          # sign-free comparison of the factors
        assert _rel(WPQ[(underline(conv, "P"), 1)], ref[(underline(conv, "P"), 1)]) <= tol
        assert _rel(net.param_data(conv), g["finalW:" + conv]) <= tol, conv
    assert np.array_equal(net.param_data("conv1_1"), g["finalW:conv1_1"])


def test_resnet_residual_target_matches_reference_net_py(ctx):
    """a6: appresb + invBN + the no-ReLU branch of dictionary_kernel on a ResNet-shaped net whose shortcut drifted after
    freezing: shared sample points, residual term, masks, alpha identical to the reference; weights <= 1e-5."""
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
        names = json.loads(str(g["names"]))
        np.random.seed(9)
        feats, points = net.extract_features(names=names, save=1)
        for key in ("bn2a_branch1", "res2a", "res2b_branch2c", "res2a_branch2c"):
            got = np.stack([points[(b, key, "randx")] for b in range(p["nBatches"])])
            assert np.array_equal(got, g["pt:%s:randx" % key]), key
        assert np.array_equal(feats["bn2a_branch1"], g["feat:bn2a_branch1"]) and np.array_equal(feats["res2a"], g["feat:res2a"])
        net.load_frozen(feats_dict=feats, points_dict=points)
        net.set_param_data("conv1", g["conv1_W"])            # the drift of the shortcut since freezing
        cfgs.alpha = 1e-3
        np.random.seed(79)
        for i, (X_name, Y_name, d_prime) in enumerate(json.loads(str(g["cases"]))):
            resY = net.invBN(net.appresb(Y_name), Y_name)
            assert np.abs(resY - g["resY%d" % i]).max() <= 1e-12 * max(1.0, np.abs(g["resY%d" % i]).max())
            idxs, W2, B2 = net.dictionary_kernel(X_name, None, d_prime, Y_name, None)
            assert np.array_equal(idxs, g["idxs%d" % i])
            assert _rel(W2, g["W%d" % i]) <= 1e-5 and _rel(B2, g["B%d" % i]) <= 1e-5
            assert cfgs.alpha == float(g["alpha%d" % i])
        assert int(np.random.randint(0, 2147483647)) == int(g["rng_next"])
    finally:
        dcfgs.model, dcfgs.res.short, dcfgs.dic.option = '', 0, cfgs.pruning_options.prb


def test_prune_resnet_loop_matches_the_reference_helpers_driven_the_same_way(ctx):
    """f4: Net.prune_resnet() -- every bottleneck: channel sampler in front of branch2a, branch2a -> branch2b, branch2b ->
    branch2c with the residual-aware target -- against golden n04, the same loop composed of the REFERENCE's own
    dictionary_kernel / appresb / invBN / W1keep / W2keep / select on the bit-portable ResNet: every selection identical,
    alpha carry and RNG stream identical, refitted weights <= 1e-5, WPQ / nonWPQ keys and shapes identical, values <= 1e-5."""
    import lib.cfgs as cfgs
    import portable_net
    from lib.net import Net
    from portable_provider import PortableProvider
    g = np.load(os.path.join(GOLDEN_DIR, "n04_resnet_loop.npz"))
    p = json.loads(str(g["params"]))
    layers, batches = portable_net.resnet_like(seed=p["seed"], B=p["B"], HW=p["HW"], nBatches=p["nBatches"], width=p["width"],
                                               mid=p["mid"])
    net = Net(None, PortableProvider(layers, batches), nBatches=p["nBatches"], nPointsPerLayer=p["nPoints"], graph=layers,
              model=cfgs.Models.resnet)
    saved = (cfgs.c.model, cfgs.c.res.short, cfgs.c.dic.option)
    cfgs.c.model, cfgs.c.res.short, cfgs.c.dic.option = cfgs.Models.resnet, 1, cfgs.pruning_options.resnet
    try:
        np.random.seed(9)
        feats, points = net.extract_features(names=json.loads(str(g["names"])), save=1)   # shared shortcut points (net.py:466-487)
    finally:
        cfgs.c.model, cfgs.c.res.short, cfgs.c.dic.option = saved
    net.load_frozen(feats_dict=feats, points_dict=points)
    cfgs.alpha = 1e-3
    np.random.seed(80)
    WPQ, nonWPQ = net.prune_resnet(json.loads(str(g["keep"])))
    steps = json.loads(str(g["steps"]))
    assert [s[1] for s in steps] == list(net.selection.keys())
    for i, (X_name, consumer, d_prime) in enumerate(steps):
        assert np.array_equal(net.selection[consumer], g["idxs%d" % i]), consumer
    assert cfgs.alpha == float(g["alpha%d" % (len(steps) - 1)])
    assert int(np.random.randint(0, 2147483647)) == int(g["rng_next"])
    keys = json.loads(str(g["wpq_keys"]))
    assert ["%s|%d" % k for k in WPQ.keys()] == keys
    for tag in keys:
        name, idx = tag.split("|")
        got, ref = np.asarray(WPQ[(name, int(idx))]), g["WPQ:" + tag]
        assert got.shape == ref.shape and _rel(got, ref) <= 1e-5, tag
    assert list(nonWPQ.keys()) == json.loads(str(g["nonwpq_keys"]))
    for k in nonWPQ:
        assert np.array_equal(nonWPQ[k], g["nonWPQ:" + k])
    for name in net.convs + net.bns + net.affines:
        assert _rel(net.param_data(name), g["finalW:" + name]) <= 1e-5 and _rel(net.param_b_data(name), g["finalb:" + name]) <= 1e-5


def test_dictionary_kernel_honours_the_refit_flags_of_dictionary(ctx):
    """Net.dictionary_kernel goes through the body of lib.decompose.dictionary(): dcfgs.nonlinear_fc / nofc give the same
    result on the Net path as calling dictionary() on the same operands (the reference routes both through dictionary(),
    net.py:1728 -> decompose.py:615-620)."""
    import lib.cfgs as cfgs
    import lib.decompose as D
    from lib.cfgs import c as dcfgs
    net, _ = make_net(seed=4, nBatches=12, nPoints=8)
    np.random.seed(21)
    net.freeze_images(convs=net.convs)
    X = net.extract_XY("conv2_1", "conv2_2")
    newX = np.maximum(np.rollaxis(X.reshape((-1, 3, 3, X.shape[1])), 3, 1), 0)
    W2, b2 = net.param_data("conv2_2"), net.param_b_data("conv2_2")
    Y = net._feats_dict["conv2_2"] - b2
    for flag in ("nonlinear_fc", "nofc"):
        setattr(dcfgs, flag, 1)
        try:
            cfgs.alpha = 1e-3
            np.random.seed(5)
            ref = D.dictionary(newX, W2, Y, rank=16, B2=b2)
            cfgs.alpha = 1e-3
            np.random.seed(5)
            got = net.dictionary_kernel("conv2_1", None, 16, "conv2_2", None)
        finally:
            setattr(dcfgs, flag, 0)
        assert np.array_equal(got[0], ref[0])
        assert np.allclose(got[1], ref[1], rtol=1e-9, atol=1e-12) and np.allclose(got[2], ref[2], rtol=1e-9, atol=1e-12)
    cfgs.alpha = 1e-3
    np.random.seed(5)
    lin = net.dictionary_kernel("conv2_1", None, 16, "conv2_2", None)
    assert not np.allclose(lin[1], got[1])      # the flags do change the answer


_ROCM_PROVIDER_SCRIPT = r'''
import os, sys, json
import torch                                   # BEFORE the first cpmi355 Context: the first HIP runtime in the process serves both
ROOT = sys.argv[1]
sys.path[:0] = [os.path.join(ROOT, "channel-pruning_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import portable_net
import lib.cfgs as cfgs
from lib.net import Net
from lib.provider import TorchGraphProvider
assert torch.cuda.is_available()
g = np.load(os.path.join(ROOT, "tests/golden/n01_vgg_pruning.npz"))
p = json.loads(str(g["params"]))
layers, batches = portable_net.vgg_like(seed=p["seed"], chans=[tuple(c) for c in p["chans"]], B=p["B"], HW=p["HW"], nBatches=p["nBatches"])
net = Net(None, TorchGraphProvider(layers, batches, device="cuda"), nBatches=p["nBatches"], nPointsPerLayer=p["nPoints"], graph=layers)
ref = portable_net.forward(layers, batches[2])
got = net.forward(2)
err = max(float(np.abs(got[k] - v).max() / max(1.0, np.abs(v).max())) for k, v in ref.items())
np.random.seed(3)
net.freeze_images(convs=net.convs)
cfgs.alpha = 1e-3
np.random.seed(77)
X_name, Y_name, d_prime = json.loads(str(g["pairs"]))[0]
idxs, W2, B2 = net.dictionary_kernel(X_name, None, d_prime, Y_name, None)
W = net.param_data(Y_name)
res_new = np.linalg.norm(net._feats_dict[Y_name] - net.param_b_data(Y_name))
print(json.dumps(dict(forward_err=err, kept=int(idxs.sum()), same_mask=bool(np.array_equal(idxs, g["idxs0"])),
                      w_rel=float(np.linalg.norm(W2 - g["W0"]) / np.linalg.norm(g["W0"])))))
'''


def test_torch_rocm_provider_drives_the_facade():
    """f3: the activation provider on torch-ROCm (lib/provider.py::TorchGraphProvider(device="cuda")) next to libcpmi355 in one
    process (torch imported first).  Its float32 convolutions differ from the bit-portable forward pass in the last bits, so
    the check is: blobs within 1e-5, and the pruning of the first pair lands on the reference's mask / weights up to what
    that noise moves (masks are compared, not asserted equal)."""
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _ROCM_PROVIDER_SCRIPT, ROOT], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["forward_err"] <= 1e-5
    assert abs(out["kept"] - 22) <= 2
    if out["same_mask"]:
        assert out["w_rel"] <= 1e-3
