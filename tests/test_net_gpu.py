"""GPU: the caffe-free Net facade (lib/net.py) -- extract_XY, dictionary_kernel, R3 -- on a small
VGG-shaped network whose activations come from a torch CPU forward (standing in for Caffe)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_net(seed=0, B=4, HW=16, nBatches=6, nPoints=5):
    import torch
    import torch.nn.functional as F
    from lib.net import ConvSpec, Net
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
