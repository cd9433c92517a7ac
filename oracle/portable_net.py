"""TEST INFRASTRUCTURE ONLY -- a tiny, bit-portable CNN forward pass in NumPy.

Stands in for the Caffe forward pass (``self.net.forward()``, /root/reference/lib/net.py:197) on BOTH sides of the
net.py parity tests: behind the fake ``caffe.Net`` that lets the UNMODIFIED reference ``lib/net.py`` run in the build
container (oracle/ref_net_loader.py, oracle/gen_golden_net.py) and behind the activation provider of this repository's
caffe-free ``Net`` facade on the GPU box (tests/test_net_gpu.py).  The channel masks downstream are compared
bit-exactly, so the activations must not depend on the BLAS / oneDNN kernels of the host: every layer here is a
fixed-order sequence of elementwise float64 NumPy operations (IEEE-exact), rounded to float32 once per blob -- what a
Caffe float32 blob would hold, up to the summation order nobody pins.

Layer descriptions are plain dicts:
    {"name", "type", "bottom": [..], "top": [..], + type-specific fields}
      Convolution  W float32[n, c, k, k], b float32[n] or None, pad, stride
      ReLU | Pooling (kernel, stride; max) | Eltwise (sum)
      BatchNorm    mean[c], var[c] (Caffe blobs 0 / 1, scale factor 1), eps
      Scale        k[c], b[c]
A blob named like its layer is that layer's output (non-in-place tops), as the reference arranges for the layers it
samples (lib/net.py:1106-1133 split the in-place ReLUs).
"""
import numpy as np


def conv2d(x, W, b, pad, stride):
    """x float32[B, C, H, W], W float32[n, C, k, k] -> float32[B, n, Ho, Wo]; accumulation in float64 over (c, kh, kw)
    in that fixed order, bias last."""
    B, C, H, Wd = x.shape
    n, _, k, _ = W.shape
    xp = np.zeros((B, C, H + 2 * pad, Wd + 2 * pad), dtype=np.float64)
    xp[:, :, pad:pad + H, pad:pad + Wd] = x
    Ho = (H + 2 * pad - k) // stride + 1
    Wo = (Wd + 2 * pad - k) // stride + 1
    out = np.zeros((B, n, Ho, Wo), dtype=np.float64)
    W64 = W.astype(np.float64)
    for c in range(C):
        for kh in range(k):
            for kw in range(k):
                patch = xp[:, c, kh:kh + stride * Ho:stride, kw:kw + stride * Wo:stride]      # [B, Ho, Wo]
                out += W64[None, :, c, kh, kw, None, None] * patch[:, None, :, :]
    if b is not None:
        out += b.astype(np.float64)[None, :, None, None]
    return out.astype(np.float32)


def max_pool(x, kernel, stride):
    B, C, H, W = x.shape
    Ho, Wo = (H - kernel) // stride + 1, (W - kernel) // stride + 1
    out = np.full((B, C, Ho, Wo), -np.inf, dtype=np.float32)
    for kh in range(kernel):
        for kw in range(kernel):
            out = np.maximum(out, x[:, :, kh:kh + stride * Ho:stride, kw:kw + stride * Wo:stride])
    return out


def forward(layers, data, params=None):
    """Run the graph.  params: optional {layer_name: [arrays]} overriding the arrays stored in the layer dicts
    (Convolution: [W, b]; BatchNorm: [mean, var]; Scale: [k, b]) -- the fake caffe net passes its live blobs."""
    blobs = {"data": np.ascontiguousarray(data, dtype=np.float32)}
    for L in layers:
        t = L["type"]
        name = L["name"]
        bot = [blobs[b] for b in L.get("bottom", [])]
        P = params.get(name) if params is not None and name in params else None
        if t == "Convolution":
            W = P[0] if P is not None else L["W"]
            b = (P[1] if P is not None and len(P) > 1 else L.get("b"))
            y = conv2d(bot[0], np.asarray(W, dtype=np.float32), None if b is None else np.asarray(b, dtype=np.float32),
                       L.get("pad", 0), L.get("stride", 1))
        elif t == "ReLU":
            y = np.maximum(bot[0], np.float32(0))
        elif t == "Pooling":
            y = max_pool(bot[0], L["kernel"], L["stride"])
        elif t == "Eltwise":
            y = (bot[0].astype(np.float64) + bot[1].astype(np.float64)).astype(np.float32)
        elif t == "BatchNorm":
            mean, var = (P[0], P[1]) if P is not None else (L["mean"], L["var"])
            eps = L.get("eps", 1e-5)
            y = ((bot[0].astype(np.float64) - np.asarray(mean, dtype=np.float64)[None, :, None, None])
                 / np.sqrt(np.asarray(var, dtype=np.float64) + eps)[None, :, None, None]).astype(np.float32)
        elif t == "Scale":
            k, b = (P[0], P[1]) if P is not None else (L["k"], L["b"])
            y = (bot[0].astype(np.float64) * np.asarray(k, dtype=np.float64)[None, :, None, None]
                 + np.asarray(b, dtype=np.float64)[None, :, None, None]).astype(np.float32)
        else:
            raise ValueError("layer type %r" % t)
        blobs[L["top"][0]] = y
    return blobs


# ---- the two small networks of the net.py goldens ------------------------------------------------------------
def vgg_like(seed=0, chans=((3, 12), (12, 12), (12, 16), (16, 16), (16, 24)), B=4, HW=16, nBatches=6):
    """conv1_1 relu conv1_2 relu pool1 conv2_1 relu conv2_2 relu pool2 conv3_1 relu: the VGG naming R3 keys on
    (lib/net.py:1307-1308).  -> (layers, batches)"""
    rs = np.random.RandomState(seed)
    names = ["conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1"]
    layers = []
    prev = "data"
    for name, (cin, cout) in zip(names, chans):
        W = (rs.randn(cout, cin, 3, 3) * (1.5 / np.sqrt(cin * 9))).astype(np.float32)
        b = (rs.randn(cout) * 0.1).astype(np.float32)
        layers.append(dict(name=name, type="Convolution", bottom=[prev], top=[name], W=W, b=b, pad=1, stride=1))
        layers.append(dict(name=name + "_relu", type="ReLU", bottom=[name], top=[name + "_relu"]))
        prev = name + "_relu"
        if name in ("conv1_2", "conv2_2"):
            pool = "pool" + name[4]
            layers.append(dict(name=pool, type="Pooling", bottom=[prev], top=[pool], kernel=2, stride=2))
            prev = pool
    batches = [rs.randn(B, 3, HW, HW).astype(np.float32) for _ in range(nBatches)]
    return layers, batches


def resnet_like(seed=0, B=4, HW=12, nBatches=6, width=16, mid=8):
    """conv1 -> [res2a: branch1 (1x1, BN) + branch2a/2b/2c (BN each)] -> res2a (sum) relu -> [res2b: branch2a/2b/2c]
    -> res2b (sum) relu.  BatchNorm tops carry the bn layer's name, Scale runs in place on that blob -- the lay-out
    the reference's invBN / appresb look for (lib/net.py:1200-1217, 1641-1683).  -> (layers, batches)"""
    rs = np.random.RandomState(seed)
    layers = []

    def conv(name, bottom, cin, cout, k):
        W = (rs.randn(cout, cin, k, k) * (1.2 / np.sqrt(cin * k * k))).astype(np.float32)
        b = (rs.randn(cout) * 0.05).astype(np.float32)
        layers.append(dict(name=name, type="Convolution", bottom=[bottom], top=[name], W=W, b=b, pad=k // 2, stride=1))
        return name

    def bn(tag, bottom, c):
        bname, sname = "bn" + tag, "scale" + tag
        layers.append(dict(name=bname, type="BatchNorm", bottom=[bottom], top=[bname],
                           mean=(rs.randn(c) * 0.1).astype(np.float32), var=(0.5 + rs.rand(c)).astype(np.float32), eps=1e-5))
        layers.append(dict(name=sname, type="Scale", bottom=[bname], top=[bname],
                           k=(0.8 + 0.4 * rs.rand(c)).astype(np.float32), b=(rs.randn(c) * 0.1).astype(np.float32)))
        return bname

    def relu(name, bottom, inplace=False):
        layers.append(dict(name=name, type="ReLU", bottom=[bottom], top=[bottom if inplace else name]))
        return bottom if inplace else name

    x = relu("conv1_relu", conv("conv1", "data", 3, width, 3))
    # block a (with projection shortcut)
    s = bn("2a_branch1", conv("res2a_branch1", x, width, width, 1), width)
    y = relu("res2a_branch2a_relu", bn("2a_branch2a", conv("res2a_branch2a", x, width, mid, 1), mid))
    y = relu("res2a_branch2b_relu", bn("2a_branch2b", conv("res2a_branch2b", y, mid, mid, 3), mid))
    y = bn("2a_branch2c", conv("res2a_branch2c", y, mid, width, 1), width)
    layers.append(dict(name="res2a", type="Eltwise", bottom=[s, y], top=["res2a"]))
    x = relu("res2a_relu", "res2a", inplace=True)      # in place like Caffe's ResNet: blob res2a is post-ReLU when sampled
    # block b (identity shortcut)
    y = relu("res2b_branch2a_relu", bn("2b_branch2a", conv("res2b_branch2a", x, width, mid, 1), mid))
    y = relu("res2b_branch2b_relu", bn("2b_branch2b", conv("res2b_branch2b", y, mid, mid, 3), mid))
    y = bn("2b_branch2c", conv("res2b_branch2c", y, mid, width, 1), width)
    layers.append(dict(name="res2b", type="Eltwise", bottom=[x, y], top=["res2b"]))
    relu("res2b_relu", "res2b", inplace=True)
    batches = [rs.randn(B, 3, HW, HW).astype(np.float32) for _ in range(nBatches)]
    return layers, batches
