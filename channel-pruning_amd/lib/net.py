"""Caffe-free ``Net`` facade for the channel-pruning path of the reference's lib/net.py.

The reference's ``Net`` wraps a pycaffe network (un-vendored fork, lib/net.py:9-10); here the
network is a plain ordered description of its conv layers plus an *activation provider* -- any
callable ``provider(batch) -> {blob_name: float32[B, C, H, W]}`` standing in for the Caffe forward
pass (lib/net.py:197), which is NOT on the accelerated path.  What is kept, with the reference's
names, argument meaning and return contracts:

    Net.extract_features(names, ...)   random point sampling + Y features     net.py:368-532
    Net.load_frozen(feats_dict, points_dict) / freeze_images()                 net.py:749-802, 839-876
    Net.extract_XY(X_name, Y_name)     sampled-point im2col                    net.py:534-684   [HIP: cp_patch_gather]
    Net.dictionary_kernel(X_name, None, d_prime, Y_name, None)                 net.py:1685-1735 [HIP: cp_assemble_y + dictionary]
    Net.R3()                           the VGG "3C" driver loop, pruning step  net.py:1292-1471
    Net.pruning_kernel / param accessors used by the above

``Net.R3()`` runs the whole "3C" loop of the reference (net.py:1292-1471) -- spatial decomposition (VH_decompose),
channel decomposition (ITQ_decompose), channel pruning (dictionary) per conv, in that order -- when the provider is
LIVE, i.e. takes ``(batch, net)`` and computes the blobs from the net's CURRENT weights (the reference re-runs the
Caffe forward on the modified net between the steps).  With a one-argument provider (activations of the original
network only) it runs the pruning step of every (producer, consumer) pair the reference prunes, in the same order,
with the same kept-channel request d_c = max(int(c / 1.15), rank) (net.py:1327,1346-1349).  The prototxt surgery
(insert / set_conv / save_pt) is replaced by ``Net.emit_layers()``: the decomposed network as an ordered list of
plain layer descriptions.
"""
import inspect
import pickle
from collections import OrderedDict

import numpy as np

from cpmi355 import LayerProblem, default_context, prune_layer
from cpmi355 import capi as _capi

from . import cfgs
from . import decompose as _decompose
from .cfgs import c as dcfgs
from .decompose import ITQ_decompose, VH_decompose, rel_error
from .utils import Timer, underline


class ConvSpec(object):
    """One convolution: weights W[n, c, k, k] float32, bias b[n] float32, geometry, and the name
    of the blob it reads (``bottom``).  Its own output blob carries the layer's name and is the
    PRE-ReLU response (the reference splits in-place ReLUs for exactly this, net.py:1106-1133)."""

    def __init__(self, name, W, b, bottom, pad=1, stride=1):
        self.name = name
        self.W = np.ascontiguousarray(W, dtype=np.float32)
        self.b = np.ascontiguousarray(b, dtype=np.float32)
        self.bottom = bottom
        self.pad = int(pad)
        self.stride = int(stride)

    @property
    def kernel_size(self):
        return int(self.W.shape[-1])


class Net(object):
    def __init__(self, convs, provider, nBatches=None, nPointsPerLayer=None, device=None, model='vgg'):
        """convs: iterable of ConvSpec in network order; provider: batch -> {blob: f32[B,C,H,W]}."""
        self.layers = OrderedDict((cv.name, cv) for cv in convs)
        self.convs = list(self.layers.keys())
        self.provider = provider
        self.nBatches = dcfgs.nBatches if nBatches is None else int(nBatches)
        self.nPointsPerLayer = dcfgs.nPointsPerLayer if nPointsPerLayer is None else int(nPointsPerLayer)
        self.model = model
        self._device = device
        self._mem = False
        self._feats_dict = dict()
        self._points_dict = dict()
        self.WPQ = dict()
        self.selection = dict()
        self.bottom_names = dict((n, [cv.bottom]) for n, cv in self.layers.items())
        self._blob_cache = (None, None)
        try:    # provider(batch, net): blobs follow the net's current weights (needed by the full 3C loop)
            self._live = len([p for p in inspect.signature(provider).parameters.values()
                              if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]) >= 2
        except (TypeError, ValueError):
            self._live = False

    # ---- accessors with the reference's names (net.py:46-349) --------------------------------
    def ctx(self):
        return default_context(self._device)

    def param_data(self, name):
        return self.layers[name].W

    def param_b_data(self, name):
        return self.layers[name].b

    def param_shape(self, name):
        return self.layers[name].W.shape

    def set_param_data(self, name, data):
        self.layers[name].W = np.ascontiguousarray(data, dtype=np.float32)
        self._blob_cache = (None, None)

    def set_param_b(self, name, data):
        self.layers[name].b = np.ascontiguousarray(data, dtype=np.float32)
        self._blob_cache = (None, None)

    def conv_param_pad(self, name):
        return self.layers[name].pad

    def conv_param_stride(self, name):
        return self.layers[name].stride

    def conv_param_kernel_size(self, name):
        return self.layers[name].kernel_size

    def forward(self, batch):
        """One provider call per batch (cached: extract_features and extract_XY share it)."""
        if self._blob_cache[0] != batch:
            self._blob_cache = (batch, self.provider(batch, self) if self._live else self.provider(batch))
        return self._blob_cache[1]

    # ---- feature sampling (net.py:368-532) -----------------------------------------------------
    def extract_features(self, names=[], nBatches=None, points_dict=None, save=0):
        """Sample nPointsPerLayer random (x, y) per layer per batch and collect the blob values
        there: feats_dict[name][N, C] float64, row order [batch][point][image] (net.py:505-510).
        Points come from numpy's global RNG (net.py:464-465) unless frozen in points_dict."""
        if not isinstance(names, list):
            names = [names]
        assert len(names) > 0
        frozen = points_dict is not None
        if frozen:
            nP, nB = points_dict["nPointsPerLayer"], points_dict["nBatches"]
        else:
            nP, nB = self.nPointsPerLayer, (self.nBatches if nBatches is None else nBatches)
            points_dict = {"nPointsPerLayer": nP, "nBatches": nB}
        feats_dict = dict()
        idx = 0
        for batch in range(nB):
            blobs = self.forward(batch)
            for name in names:
                feat = blobs[name]
                B, C, H, W = feat.shape
                if name not in feats_dict:
                    feats_dict[name] = np.ndarray(shape=(nP * B * nB, C))
                if (batch, name, "randx") in points_dict:
                    randx, randy = points_dict[(batch, name, "randx")], points_dict[(batch, name, "randy")]
                else:
                    randx = np.random.randint(0, H, nP)
                    randy = np.random.randint(0, W, nP)
                    points_dict[(batch, name, "randx")] = randx.copy()
                    points_dict[(batch, name, "randy")] = randy.copy()
                for point, x, y in zip(range(nP), randx, randy):
                    i_from = idx + point * B
                    feats_dict[name][i_from:(i_from + B)] = feat[:, :, x, y].reshape((B, -1))
            idx += nP * feat.shape[0]
        if save or frozen:
            return feats_dict, points_dict
        return feats_dict

    def freeze_images(self, path=None, convs=None):
        """extract_features(save=1) + pickle [feats_dict, points_dict] (net.py:749-802)."""
        feats_dict, points_dict = self.extract_features(names=convs or self.convs, save=1)
        if path is not None:
            with open(path, 'wb') as f:
                pickle.dump([feats_dict, points_dict], f, protocol=4)
        self.load_frozen(feats_dict=feats_dict, points_dict=points_dict)
        return path

    def load_frozen(self, path=None, feats_dict=None, points_dict=None):
        """net.py:839-876: adopt frozen features / points (from memory or from the pickle)."""
        if feats_dict is None:
            with open(path, 'rb') as f:
                feats_dict, points_dict = pickle.load(f)
        self._feats_dict = feats_dict
        self._points_dict = points_dict
        self._mem = True

    # ---- sampled-point im2col (net.py:534-684) -----------------------------------------------------
    def _gather_patches_device(self, X, Y, relu=0):
        """cp_patch_gather over all batches -> device buffer X[N, C, k, k] float32 (optionally ReLU'd)."""
        ctx = self.ctx()
        pad, k, stride = self.conv_param_pad(Y), self.conv_param_kernel_size(Y), self.conv_param_stride(Y)
        nP, nB = self._points_dict["nPointsPerLayer"], self._points_dict["nBatches"]
        out = None
        row0 = 0
        for batch in range(nB):
            blob = np.ascontiguousarray(self.forward(batch)[X], dtype=np.float32)
            B, C, H, W = blob.shape
            if out is None:
                N = nP * B * nB
                out = ctx.empty(N * C * k * k * 4)
            fd = ctx.to_device(blob)
            ctx.patch_gather(fd, B, C, H, W, self._points_dict[(batch, Y, "randx")],
                             self._points_dict[(batch, Y, "randy")], k, pad, stride, int(relu), out, row0)
            row0 += nP * B
            ctx.sync()
            fd.free()
        return out, N, C, k

    def extract_XY(self, X, Y, DEBUG=False, w1=None):
        """feats[N*k*k, C] float64 exactly as the reference lays it out (rows [batch][point][image]
        [kh][kw]); see dictionary_kernel for the device-resident variant used internally."""
        if w1 is not None:
            raise NotImplementedError("gw1 (two-layer window) branch is not part of the pruning path")
        Xd, N, C, k = self._gather_patches_device(X, Y)
        ctx = self.ctx()
        X4 = ctx.to_host(Xd, (N, C, k, k), np.float32)
        Xd.free()
        return np.moveaxis(X4, 1, -1).reshape((N * k * k, C)).astype(np.float64)

    # ---- residual target hook (net.py:1641-1683); VGG has none ------------------------------------
    def appresb(self, Y_name):
        return 0

    # ---- wrapper (net.py:1685-1735) ---------------------------------------------------------------
    def dictionary_kernel(self, X_name, weights, d_prime, Y_name, Y, DEBUG=0):
        """Channel-pruning wrapper: which channels of X_name to keep so that conv Y_name can still
        reproduce its sampled responses.  Returns (idxs, newW2, newB2) like the reference."""
        if not self._mem:
            feats_dict, points_dict = self.extract_features([X_name, Y_name], save=1)
            self.load_frozen(feats_dict=feats_dict, points_dict=points_dict)
        ctx = self.ctx()
        relu_x = self.model not in (cfgs.Models.xception, cfgs.Models.resnet)   # net.py:1717-1720
        Xd, N, C, k = self._gather_patches_device(X_name, Y_name, relu=relu_x)     # relu(newX) fused
        W2 = self.param_data(Y_name)
        b2 = self.param_b_data(Y_name)
        n = W2.shape[0]
        feats = self._feats_dict[Y_name]
        resY = self.appresb(Y_name)
        # Y = feats - bias (+ resY) on the device when the features are float32-representable
        # (they are: they were sampled from float32 blobs); otherwise assemble in float64 on the host.
        f32 = feats.astype(np.float32)
        Yd = ctx.empty(N * n * 8)
        if np.array_equal(f32.astype(np.float64), feats):
            fd, bd = ctx.to_device(f32), ctx.to_device(b2)
            rd = None if isinstance(resY, int) and resY == 0 else ctx.to_device(np.asarray(resY, dtype=np.float64))
            ctx.assemble_y(fd, bd, rd, N, n, Yd)
        else:
            ctx.to_device(np.ascontiguousarray(feats - b2 + resY, dtype=np.float64), Yd)
        prob = LayerProblem.from_device(ctx, Xd, _capi.CP_F32, N, C, k, W2, Yd,
                                        flags=(_capi.CP_CD_RECIPROCAL if dcfgs.cd_reciprocal else 0)
                                        | (_capi.CP_CD_DELTA if dcfgs.cd_delta else 0))
        try:
            idxs, newW2, newB2, alpha_out = prune_layer(prob, d_prime, cfgs.alpha, rank_tol=dcfgs.dic.rank_tol,
                                                        rng=np.random, ridge=float(dcfgs.fc_ridge),
                                                        mode=dcfgs.cd_mode)
            _decompose.last_call_info.clear()
            _decompose.last_call_info.update(fits=list(prob.fits), samples=prob.samples,
                                             fallback=int(prob.refit_info.fallback), p=int(prob.refit_info.p))
        finally:
            prob.free()
            Xd.free()
            Yd.free()
        cfgs.alpha = alpha_out
        return idxs, newW2, newB2

    # ---- baseline pruner (net.py:1632-1639) -------------------------------------------------------
    def pruning_kernel(self, X_name, d_prime, Y_name):
        W2 = self.param_data(Y_name)
        order = np.argsort(-np.abs(W2).sum((0, 2, 3)))
        idxs = np.zeros(W2.shape[1], dtype=bool)
        idxs[order[:d_prime]] = True
        return idxs, W2[:, idxs], self.param_b_data(Y_name)

    # ---- driver (net.py:1292-1471) ------------------------------------------------------------------
    def R3(self, alldic=None, pooldic=None, rankdic=None, decompose=None):
        """The reference's "3C" loop over VGG-16 (net.py:1292-1471).

        decompose (default: True with a live provider, else False):
          True   per conv (from the second on): spatial decomposition (VH_decompose with the ReLU-aware refit of H on
                 sampled patches), channel decomposition (ITQ_decompose on the features of the modified net), then
                 channel pruning of its outputs against the next conv -- exactly the reference's order; WPQ gets the
                 reference's keys: conv_V -> V[rank, c, k, 1]; (conv_H, 0/1) -> W'[d', rank, 1, k], zeros(d');
                 (conv_P, 0/1) -> P[n, d', 1, 1], B (rows [idxs] once the conv was pruned as a producer).
          False  the pruning step alone: WPQ maps (layer, 0) / (layer, 1) to the layer's compact weights / bias
                 (input channels pruned when it was the consumer, filters pruned when it was the producer).
        Returns (WPQ, new_pt); new_pt is the prefix string the reference derives its output prototxt name from
        ('3C4x' for dic.keep = 3, net.py:1293-1300)."""
        speed_ratio = dcfgs.dic.keep
        prefix = ('3C' if dcfgs.dic.vh else '2C') + str(int(speed_ratio) + 1) + 'x'
        convs = self.convs
        if decompose is None:
            decompose = self._live
        if decompose and not self._live:
            raise ValueError("the decomposition steps re-extract features from the MODIFIED network: the provider "
                             "has to be live, provider(batch, net)")
        self.WPQ = dict()
        self.selection = dict()
        self._decomposed = bool(decompose)
        self._mem = bool(self._feats_dict)
        if decompose and not self._mem:
            raise ValueError("R3 needs the frozen features of the original network: call freeze_images() first "
                             "(the reference asserts the same through self._mem = True, net.py:1305)")
        end = 5
        if alldic is None:
            alldic = ['conv%d_1' % i for i in range(1, end)] + ['conv%d_2' % i for i in range(3, end)]
        if pooldic is None:
            pooldic = ['conv1_2', 'conv2_2']
        if rankdic is None:
            rankdic = {'conv1_1': 17, 'conv1_2': 17, 'conv2_1': 37, 'conv2_2': 47, 'conv3_1': 83, 'conv3_2': 89,
                       'conv3_3': 106, 'conv4_1': 175, 'conv4_2': 192, 'conv4_3': 227, 'conv5_1': 398,
                       'conv5_2': 390, 'conv5_3': 379}
            rankdic = dict(rankdic)
            for i in rankdic:
                if 'conv5' in i:
                    continue
                rankdic[i] = int(rankdic[i] * 4. / speed_ratio)
        c_ratio = 1.15

        def getX(name):                                                        # net.py:1329-1331
            x = self.extract_XY(self.bottom_names[name][0], name)
            k = self.conv_param_kernel_size(name)
            return np.rollaxis(x.reshape((-1, k, k, x.shape[1])), 3, 1).copy()

        def setConv(c, d):                                                     # net.py:1333-1337
            if c in self.selection:
                Wc = self.param_data(c).copy()
                Wc[:, self.selection[c], :, :] = d
                self.set_param_data(c, Wc)
            else:
                self.set_param_data(c, d)

        t = Timer()
        for conv, convnext in zip(convs[1:], convs[2:] + ['pool5']):
            conv_V, conv_H, conv_P = underline(conv, 'V'), underline(conv, 'H'), underline(conv, 'P')
            W_shape = self.param_shape(conv)
            d_c = int(W_shape[0] / c_ratio)
            rank = rankdic.get(conv, d_c)
            d_prime = rank
            if d_c < rank:
                d_c = rank
            if decompose:
                # ---- spatial decomposition (net.py:1351-1379) ----
                t.tic()
                weights = self.param_data(conv)
                if conv in self.selection:
                    weights = weights[:, self.selection[conv], :, :]
                Y = self._feats_dict[conv] - self.param_b_data(conv)
                X = getX(conv)
                if conv in self.selection:
                    X = X[:, self.selection[conv], :, :]
                V, H, VHr, b = VH_decompose(weights, rank=rank, DEBUG=True, X=X, Y=Y)
                self.set_param_b(conv, b)
                self.WPQ[conv_V] = V
                setConv(conv, VHr)                        # the net keeps computing with the low-rank k x k weights
                self.WPQ[(conv_H, 0)] = H
                self.WPQ[(conv_H, 1)] = self.param_b_data(conv)
                t.toc('spatial_decomposition')
                # ---- channel decomposition (net.py:1383-1404) ----
                t.tic()
                feats_dict, _ = self.extract_features(names=conv, points_dict=self._points_dict, save=1)
                W1, W2, B, W12 = ITQ_decompose(feats_dict[conv], self._feats_dict[conv], H, d_prime,
                                               bias=self.param_b_data(conv), DEBUG=0, Wr=VHr)
                setConv(conv, W12.copy())
                self.set_param_b(conv, B.copy())
                self.WPQ[(conv_H, 0)] = W1.reshape([d_prime, H.shape[1], H.shape[2], H.shape[3]])
                self.WPQ[(conv_H, 1)] = np.zeros(d_prime)
                self.WPQ[(conv_P, 0)] = W2.reshape([W2.shape[0], W2.shape[1], 1, 1])
                self.WPQ[(conv_P, 1)] = B
                t.toc('channel_decomposition')
            if dcfgs.dic.vh and (conv in alldic or conv in pooldic) and (convnext in self.convs):
                t.tic()
                X_name = self.bottom_names[convnext][0] if conv in pooldic else conv
                idxs, W2, B2 = self.dictionary_kernel(X_name, None, d_c, convnext, None)
                self.selection[convnext] = idxs                                   # net.py:1445
                Wn = self.param_data(convnext).copy()
                Wn[:, ~idxs, ...] = 0
                Wn[:, idxs, ...] = W2
                self.set_param_data(convnext, Wn)
                self.set_param_b(convnext, B2)
                if decompose:                                                     # net.py:1451-1457
                    key = conv_P if (conv_P, 0) in self.WPQ else conv_H
                    self.WPQ[(key, 0)] = self.WPQ[(key, 0)][idxs]
                    self.WPQ[(key, 1)] = self.WPQ[(key, 1)][idxs]
                else:
                    # compact weights: the consumer keeps only the selected input channels ...
                    self.WPQ[(convnext, 0)] = W2.astype(np.float32)
                    self.WPQ[(convnext, 1)] = np.asarray(B2, dtype=np.float32)
                    # ... and the producer only the filters that feed them (net.py:1455-1457); a layer
                    # that was a consumer one iteration earlier is already compact on its input axis
                    Wc = self.WPQ.get((conv, 0), self.param_data(conv))
                    bc = self.WPQ.get((conv, 1), self.param_b_data(conv))
                    self.WPQ[(conv, 0)] = Wc[idxs]
                    self.WPQ[(conv, 1)] = bc[idxs]
                t.toc('channel_pruning')
        return self.WPQ, underline(prefix, 'pruned')

    # ---- what the reference writes into the new prototxt/caffemodel (net.py:1459-1470, 884-911, 967-988) -------
    def emit_layers(self):
        """The network after R3 as an ordered list of dicts
            {name, top, bottom, W float32[n, c, kh, kw], b float32[n] or None, pad (ph, pw), stride (sh, sw)}.
        A decomposed conv becomes conv_V (k x 1, no bias) -> conv_H (1 x k) -> conv_P (1 x 1) exactly as the reference
        inserts them (pad / kernel inferred from the weight shape, infer_pad_kernel; the stride of a strided conv goes
        to the axis the kernel extends along).  The last layer of each chain writes the ORIGINAL blob name (`top`), so
        ReLU / pooling / the consumers' bottoms stay as they were.  Undecomposed layers are emitted with their
        compact weights."""
        out = []
        decomposed = getattr(self, "_decomposed", False)
        for name in self.convs:
            cv = self.layers[name]
            conv_V, conv_H, conv_P = underline(name, 'V'), underline(name, 'H'), underline(name, 'P')
            if decomposed and conv_V in self.WPQ:
                def geom(W):
                    kh, kw = W.shape[2], W.shape[3]
                    return ((cv.pad if kh > 1 else 0, cv.pad if kw > 1 else 0),
                            (cv.stride if kh > 1 else 1, cv.stride if kw > 1 else 1))
                V = np.asarray(self.WPQ[conv_V], dtype=np.float32)
                Hw = np.asarray(self.WPQ[(conv_H, 0)], dtype=np.float32)
                Hb = np.asarray(self.WPQ[(conv_H, 1)], dtype=np.float32)
                chain = [(conv_V, V, None), (conv_H, Hw, Hb)]
                if (conv_P, 0) in self.WPQ:
                    chain.append((conv_P, np.asarray(self.WPQ[(conv_P, 0)], dtype=np.float32),
                                  np.asarray(self.WPQ[(conv_P, 1)], dtype=np.float32)))
                bottom = cv.bottom
                for i, (lname, W, b) in enumerate(chain):
                    pad, stride = geom(W)
                    top = name if i == len(chain) - 1 else lname
                    out.append(dict(name=lname, top=top, bottom=bottom, W=W, b=b, pad=pad, stride=stride))
                    bottom = top
            else:
                W = np.asarray(self.WPQ.get((name, 0), cv.W), dtype=np.float32)
                b = np.asarray(self.WPQ.get((name, 1), cv.b), dtype=np.float32)
                out.append(dict(name=name, top=name, bottom=cv.bottom, W=W, b=b, pad=(cv.pad, cv.pad),
                                stride=(cv.stride, cv.stride)))
        return out


__all__ = ["Net", "ConvSpec", "rel_error"]
