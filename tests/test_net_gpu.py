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
