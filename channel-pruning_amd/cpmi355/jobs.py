"""Workload tables of the whole-network pruning jobs (BASELINE.json configs[2..4]) and their synthetic operands.

Every job is a list of independent dictionary() problems (lib/decompose.py:386-634) -- one per pruned
(producer, consumer) conv pair -- described by a spec dict

    {layer_id, name, N, c, n, k, rank, residual, relu}

with the kept-channel counts taken from the reference's own fixtures (cited per table).  layer_id seeds the operands
(RandomState(1000 + layer_id)) and the layer's own NumPy stream (seed 1234 + layer_id), so
that a reference golden `tests/golden/<name>.npz` pins every layer on its own (parity is per layer: SURVEY.md 8e).

    vgg16_4x()    configs[2]: the 12 conv -> conv pairs of VGG-16, d_c = int(c / 1.15) (the reference's 3C-4x table), N = 5000
    vgg16_5x()    configs[4]: the 10 pruned pairs of the released 5x model, N = 20000 samples per layer
    resnet50_2x() configs[3]: the 40 selections of the released ResNet-50 2x model (channel samplers in front of
                  branch2a with c up to 2048, branch2a -> branch2b 3x3, branch2b -> branch2c 1x1 with the residual-aware
                  target), N = 5000
"""
import numpy as np

# ---- VGG-16 3C-4x (lib/net.py:1309-1327, 1346-1349) ------------------------------------------------------------------
VGG16_PAIRS = [("conv1_1", "conv1_2", 64, 64), ("conv1_2", "conv2_1", 64, 128), ("conv2_1", "conv2_2", 128, 128),
               ("conv2_2", "conv3_1", 128, 256), ("conv3_1", "conv3_2", 256, 256), ("conv3_2", "conv3_3", 256, 256),
               ("conv3_3", "conv4_1", 256, 512), ("conv4_1", "conv4_2", 512, 512), ("conv4_2", "conv4_3", 512, 512),
               ("conv4_3", "conv5_1", 512, 512), ("conv5_1", "conv5_2", 512, 512), ("conv5_2", "conv5_3", 512, 512)]
VGG16_RANKDIC = {'conv1_1': 17, 'conv1_2': 17, 'conv2_1': 37, 'conv2_2': 47, 'conv3_1': 83, 'conv3_2': 89, 'conv3_3': 106,
                 'conv4_1': 175, 'conv4_2': 192, 'conv4_3': 227, 'conv5_1': 398, 'conv5_2': 390, 'conv5_3': 379}


def vgg16_4x(N=5000):
    """the reference's VGG-16 rank table (net.py:1309-1321) scaled by 4/3 (:1323-1326) never exceeds int(c / 1.15), so
    d_c = int(c / 1.15) for every pair (net.py:1346-1349): 55, 55, 111, 111, 222 x3, 445 x5"""
    specs = []
    for i, (prod, cons, c, n) in enumerate(VGG16_PAIRS):
        rank = VGG16_RANKDIC[prod] if 'conv5' in prod else int(VGG16_RANKDIC[prod] * 4. / 3.)
        specs.append(dict(layer_id=101 + i, name="V%02d_%s_%s" % (i + 1, prod, cons), N=N, c=c, n=n, k=3,
                          rank=max(int(c / 1.15), rank), residual=False, relu=True))
    return specs


# ---- VGG-16 5x (temp/channel_pruning.prototxt:57, 74, 102, 119, 147, 164, 181, 209, 226, 243: num_output of the
# pruned producers; conv5_x keep their 512 channels, :271-305) ---------------------------------------------------------
VGG16_5X_KEPT = {'conv1_1': 24, 'conv1_2': 22, 'conv2_1': 41, 'conv2_2': 51, 'conv3_1': 108, 'conv3_2': 89, 'conv3_3': 111,
                 'conv4_1': 184, 'conv4_2': 276, 'conv4_3': 228}


def vgg16_5x(N=20000):
    specs = []
    for i, (prod, cons, c, n) in enumerate(VGG16_PAIRS):
        if prod not in VGG16_5X_KEPT:
            continue
        specs.append(dict(layer_id=201 + i, name="W%02d_%s_%s" % (i + 1, prod, cons), N=N, c=c, n=n, k=3,
                          rank=VGG16_5X_KEPT[prod], residual=False, relu=True))
    return specs


# ---- ResNet-50 2x (temp/resnet-50-cp.prototxt) --------------------------------------------------------------------
# per bottleneck: (block, input channels, width, output channels, channels the sampler in front of branch2a keeps
# [the Filter layer the reference's select() inserts: lib/net.py:1627-1630, lib/builder.py:666-672], branch2a num_output,
# branch2b num_output).  A count equal to the original width means the reference left that conv alone.
RESNET50_BLOCKS = [
    ("res2a", 64, 64, 256, 35, 64, 55), ("res2b", 256, 64, 256, 101, 51, 39), ("res2c", 256, 64, 256, 97, 50, 37),
    ("res3a", 256, 128, 512, 144, 128, 106), ("res3b", 512, 128, 512, 205, 105, 72), ("res3c", 512, 128, 512, 198, 105, 72),
    ("res3d", 512, 128, 512, 288, 128, 110),
    ("res4a", 512, 256, 1024, 278, 256, 225), ("res4b", 1024, 256, 1024, 418, 209, 147),
    ("res4c", 1024, 256, 1024, 407, 204, 158), ("res4d", 1024, 256, 1024, 423, 212, 155),
    ("res4e", 1024, 256, 1024, 412, 211, 148), ("res4f", 1024, 256, 1024, 595, 256, 213),
    ("res5a", 1024, 512, 2048, 606, 512, 433), ("res5b", 2048, 512, 2048, 1222, 512, 437),
    ("res5c", 2048, 512, 2048, 1147, 512, 440),
]


def resnet50_2x(N=5000):
    """Per bottleneck up to three dictionary() problems:
       sel   the block input (c = 64 ... 2048 channels) sampled for branch2a (1x1 consumer, n = width)
       b2a   branch2a's outputs pruned against branch2b (3x3 consumer)            -- where the released model pruned them
       b2b   branch2b's outputs pruned against branch2c (1x1 consumer, n = 4 x width) with the residual-aware target
             Y + (shortcut of the original net - shortcut of the pruned net) (lib/net.py:1641-1683, 1716-1722), no ReLU on X
    """
    specs = []
    lid = 301
    for blk, cin, width, cout, sel, k2a, k2b in RESNET50_BLOCKS:
        rows = [("sel", cin, width, 1, sel, False), ("b2a", width, width, 3, k2a, False), ("b2b", width, cout, 1, k2b, True)]
        for kind, c, n, k, kept, residual in rows:
            if kept < c:
                specs.append(dict(layer_id=lid, name="R%02d_%s_%s" % (lid - 300, blk, kind), N=N, c=c, n=n, k=k, rank=kept,
                                  residual=residual, relu=not residual))
            lid += 1
    return specs


JOBS = {"vgg16": vgg16_4x, "vgg16_5x": vgg16_5x, "resnet50": resnet50_2x}


def synth(spec):
    """SURVEY.md section 8d generator (the arithmetic the reference goldens under tests/golden/ were generated from; the
    equality is pinned by tests/test_host_logic.py; restated here so that the product path imports nothing from the test
    infrastructure):
    -> X[N,c,k,k] float32, W2[n,c,k,k] float32, Y[N,n] float64, B2[n] float32"""
    N, c, n, k = spec["N"], spec["c"], spec["n"], spec["k"]
    residual = bool(spec.get("residual", False))
    rs = np.random.RandomState(1000 + spec["layer_id"])
    X = rs.randn(N, c, k, k)
    if spec.get("relu", True) and not residual:
        X = np.maximum(X, 0.)
    X = X.astype(np.float32)
    W2 = (rs.randn(n, c, k, k) * 0.05).astype(np.float32)
    B2 = np.zeros(n, dtype=np.float32)
    Y = X.reshape(N, -1).astype(np.float64) @ W2.reshape(n, -1).T.astype(np.float64) + 0.01 * rs.randn(N, n)
    if residual:
        Y = Y + 0.1 * rs.randn(N, n)
    return X, W2, Y, B2
