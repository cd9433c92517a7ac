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
    Net.appresb / invBN / getBNaff     ResNet residual-aware target             net.py:1641-1683, 1200-1217, 1106-1112
    Net.W1keep / W2keep / select / combineHP   write-back bookkeeping           net.py:1521-1630, 1473-1504
    Net.prune_resnet(keep)             the bottleneck-by-bottleneck ResNet loop those helpers serve (not in the release)
    Net.pruning_kernel / param accessors used by the above

Networks with more than a plain conv stack (BatchNorm / Scale / Eltwise shortcuts: ResNet) pass ``graph`` -- the
ordered list of ALL layers as plain dicts (lib/provider.py describes the format) -- from which the facade derives what
the reference reads off the prototxt: ``bns``, ``affines``, ``sums``, ``pools``, ``bottom_names`` and the BatchNorm /
Scale parameters.

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
    def __init__(self, convs, provider, nBatches=None, nPointsPerLayer=None, device=None, model='vgg', graph=None):
        """convs: iterable of ConvSpec in network order (None: taken from the Convolution entries of `graph`);
        provider: batch -> {blob: f32[B,C,H,W]}; graph: optional list of layer dicts (every layer of the network)."""
        self.graph = list(graph) if graph is not None else None
        if convs is None:
            convs = [ConvSpec(L["name"], L["W"], L["b"], L["bottom"][0], pad=L.get("pad", 0), stride=L.get("stride", 1))
                     for L in self.graph if L["type"] == "Convolution"]
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
        self.nonWPQ = dict()
        self.bottoms2ch = []
        self.num_output = dict()         # set_conv(num_output=...) bookkeeping (the reference edits the prototxt)
        self.removed = []                # layers combineHP() merged away
        self.aux = dict()                # BatchNorm: [mean, var]; Scale: [k, b]  (float32, like Caffe blobs)
        self.bns, self.affines, self.sums, self.pools, self.relus = [], [], [], [], []
        self._layer_bottom = dict((n, cv.bottom) for n, cv in self.layers.items())
        if self.graph is not None:
            for L in self.graph:
                t, name = L["type"], L["name"]
                self.bottom_names[name] = list(L["bottom"])
                self._layer_bottom[name] = L["bottom"][0] if len(L["bottom"]) == 1 else list(L["bottom"])
                if t == "BatchNorm":
                    self.bns.append(name)
                    self.aux[name] = [np.array(L["mean"], dtype=np.float32), np.array(L["var"], dtype=np.float32)]
                elif t == "Scale":
                    self.affines.append(name)
                    self.aux[name] = [np.array(L["k"], dtype=np.float32), np.array(L["b"], dtype=np.float32)]
                elif t == "Eltwise":
                    self.sums.append(name)
                elif t == "Pooling":
                    self.pools.append(name)
                elif t == "ReLU":
                    self.relus.append(name)
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
        if name in self.aux:
            return self.aux[name][0]
        return self.layers[name].W

    def param_b_data(self, name):
        if name in self.aux:
            return self.aux[name][1]
        return self.layers[name].b

    def layer_bottom(self, name):
        return self._layer_bottom[name]

    def set_conv(self, conv, num_output=0, **kwargs):
        """net.py:308-341 edits the prototxt; here only the output count is remembered (emit_layers reads the weights)"""
        if num_output:
            self.num_output[conv] = int(num_output)

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
        """Sample nPointsPerLayer random (x, y) per layer per batch and collect the blob values there:
        feats_dict[name][N, C] float64, row order [batch][point][image] (net.py:505-510).  Points come from numpy's
        global RNG (net.py:464-465) unless frozen in points_dict.  With save, points_dict gets the reference's keys:
        "nPointsPerLayer", "nBatches", "data" / "label" (blob shapes), (batch, 0) / (batch, 1) (the images and labels
        of every batch, net.py:431-433) and (batch, name, "randx" / "randy").  dcfgs.dic.option == resnet: a shortcut
        blob (Eltwise sum / branch1 BatchNorm) is sampled at the points of the branch2c conv it is added to
        (net.py:466-487), so that appresb() can subtract row by row."""
        if not isinstance(names, list):
            names = [names]
        assert len(names) > 0
        frozen = points_dict is not None
        if frozen:
            nP, nB = points_dict["nPointsPerLayer"], points_dict["nBatches"]
        else:
            nP, nB = self.nPointsPerLayer, (self.nBatches if nBatches is None else nBatches)
            points_dict = {"nPointsPerLayer": nP, "nBatches": nB} if save else None
        feats_dict = dict()
        idx = 0
        for batch in range(nB):
            blobs = self.forward(batch)
            if save and not frozen and blobs.get("data") is not None:
                # a provider that exposes its input blob: the images and labels of every batch are frozen with the points
                # (net.py:431-433).  One that does not (batch -> {blob: array} is the whole contract) freezes features
                # and points only; load_frozen() then leaves the provider's batches alone.
                data = np.asarray(blobs["data"])
                label = np.asarray(blobs["label"]) if blobs.get("label") is not None else \
                    np.zeros((data.shape[0], 1, 1, 1), dtype=np.float32)
                if batch == 0:
                    points_dict["data"] = tuple(data.shape)
                    points_dict["label"] = tuple(label.shape)
                points_dict[(batch, 0)] = data.copy()
                points_dict[(batch, 1)] = label.copy()
            for name in names:
                feat = blobs[name]
                B, C, H, W = feat.shape
                if name not in feats_dict:
                    feats_dict[name] = np.ndarray(shape=(nP * B * nB, C))
                if save and (batch, name, "randx") in points_dict:
                    randx, randy = points_dict[(batch, name, "randx")], points_dict[(batch, name, "randy")]
                else:
                    randx = np.random.randint(0, H, nP)
                    randy = np.random.randint(0, W, nP)
                    if save:
                        shared = self._shared_points_name(name, names)
                        if shared is not None:                       # drawn, then replaced: the RNG stream of net.py:464-487
                            randx = points_dict[(batch, shared, "randx")]
                            randy = points_dict[(batch, shared, "randy")]
                        points_dict[(batch, name, "randx")] = randx.copy()
                        points_dict[(batch, name, "randy")] = randy.copy()
                for point, x, y in zip(range(nP), randx, randy):
                    i_from = idx + point * B
                    feats_dict[name][i_from:(i_from + B)] = feat[:, :, x, y].reshape((B, -1))
            idx += nP * feat.shape[0]
        if save:
            return feats_dict, points_dict
        return feats_dict

    def _shared_points_name(self, name, names):
        """net.py:466-483: which layer's sample points a shortcut blob re-uses (ResNet option only)"""
        if dcfgs.dic.option != cfgs.pruning_options.resnet:
            return None
        if name in self.sums:
            nextblock = self.sums[self.sums.index(name) + 1]
            if nextblock + '_branch1' not in names:
                return nextblock + '_branch2c'           # the previous sum and branch2c will be identical
        elif name in self.bns:
            tag = name.split('bn')[1].split('_')[0]
            if dcfgs.model == cfgs.Models.xception:
                return 'interstellar' + tag + '_branch2c'
            if dcfgs.model == cfgs.Models.resnet:
                return 'res' + tag + '_branch2c'
        return None

    def freeze_images(self, path=None, convs=None):
        """extract_features(save=1) + pickle [feats_dict, points_dict], protocol 4 (net.py:749-802): the file is
        interchangeable with the reference's frozen<nBatches>.pickle."""
        feats_dict, points_dict = self.extract_features(names=convs or self.convs, save=1)
        if path is not None:
            with open(path, 'wb') as f:
                pickle.dump([feats_dict, points_dict], f, protocol=4)
        self.load_frozen(feats_dict=feats_dict, points_dict=points_dict)
        return path

    def load_frozen(self, path=None, feats_dict=None, points_dict=None):
        """net.py:839-876: adopt frozen features / points (from memory or from the pickle).  A provider that can be
        re-pointed (``set_batches``) is handed the frozen images (batch, 0) -- the reference feeds exactly those
        through its MemoryData layer (net.py:420-421, 622-625)."""
        if feats_dict is None:
            with open(path, 'rb') as f:
                feats_dict, points_dict = pickle.load(f)
        self._feats_dict = feats_dict
        self._points_dict = points_dict
        self._mem = True
        if hasattr(self.provider, "set_batches") and (0, 0) in points_dict:
            self.provider.set_batches([points_dict[(b, 0)] for b in range(points_dict["nBatches"])],
                                      [points_dict[(b, 1)] for b in range(points_dict["nBatches"])])
            self._blob_cache = (None, None)

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

    # ---- residual-aware target (net.py:1641-1683, 1200-1217, 1106-1112) ---------------------------------
    def getBNaff(self, bn, affine, scale=1.):
        eps = 1e-9                                                   # the reference's constant (net.py:1107)
        mean = scale * self.param_data(bn)
        variance = (scale * self.param_b_data(bn) + eps) ** .5
        return mean, variance, self.param_data(affine), self.param_b_data(affine)

    def appresb(self, Y_name):
        """What the shortcut of a residual block has drifted by since the features were frozen (the earlier layers were
        pruned): frozen - current features of the shortcut blob, sampled at Y_name's points.  0 unless
        dcfgs.res.short == 1 (net.py:1641-1683)."""
        residual_B = 0

        def extractResB(a):
            feats_dict, _ = self.extract_features([a], points_dict=self._points_dict, save=1)
            return self._feats_dict[a] - feats_dict[a]

        if dcfgs.res.short == 1:
            if dcfgs.dic.option == cfgs.pruning_options.resnet:
                b2c = '_branch2c'
                if b2c in Y_name:
                    b1sum = Y_name.partition(b2c)[0]
                    if b1sum + '_branch1' in self.convs:
                        if len(self.bns) == 0:
                            residual_B = extractResB(b1sum + '_branch1')
                        else:
                            for bn in self.bottom_names[b1sum]:
                                if bn in self.bns and 'branch1' in bn:
                                    residual_B = extractResB(bn)
                                    for k in range(self._points_dict["nBatches"]):
                                        assert np.array_equal(self._points_dict[(k, Y_name, "randx")],
                                                              self._points_dict[(k, bn, "randx")])
                                    break
                    else:
                        residual_B = extractResB(self.sums[self.sums.index(b1sum) - 1])
            elif dcfgs.dic.option == 1:
                b2c = '_conv1'
                if Y_name.endswith(b2c):
                    fsums = ['first_conv'] + self.sums
                    blockname = Y_name.partition(b2c)[0]
                    blockproj = blockname + '_proj'
                    b1sum = fsums[fsums.index(blockname + '_sum') - 1]
                    if blockproj in self.convs:
                        b1sum = blockproj
                    residual_B = extractResB(b1sum)
        return residual_B

    def invBN(self, arr, Y_name):
        """The residual lives behind branch2c's BatchNorm + Scale; the regression target is the conv's own output:
        arr * std / k (net.py:1200-1217)."""
        if isinstance(arr, int) or len(self.bns) == 0 or len(self.affines) == 0:
            return arr
        interstellar = Y_name.split('_')[0]
        bn = affine = None
        for i in self.bottom_names[interstellar]:
            if i in self.bns and 'branch2c' in i:
                bn = i
                break
        for i in self.affines:
            if self.layer_bottom(i) == bn:
                affine = i
                break
        mean, std, k, b = self.getBNaff(bn, affine)
        return arr * std / k

    # ---- wrapper (net.py:1685-1735) ---------------------------------------------------------------
    def dictionary_kernel(self, X_name, weights, d_prime, Y_name, Y, DEBUG=0):
        """Channel-pruning wrapper: which channels of X_name to keep so that conv Y_name can still
        reproduce its sampled responses.  Returns (idxs, newW2, newB2) like the reference; goes through the same
        body as lib.decompose.dictionary(), so dcfgs.nonlinear_fc / nofc / fc_ridge / dic.rank_tol apply."""
        if not self._mem:
            feats_dict, points_dict = self.extract_features([X_name, Y_name], save=1)
            self.load_frozen(feats_dict=feats_dict, points_dict=points_dict)
        ctx = self.ctx()
        resnet_like = dcfgs.model in (cfgs.Models.xception, cfgs.Models.resnet) or \
            self.model in (cfgs.Models.xception, cfgs.Models.resnet)
        W2 = self.param_data(Y_name)
        b2 = self.param_b_data(Y_name)
        n = W2.shape[0]
        feats = self._feats_dict[Y_name]
        resY = self.appresb(Y_name)                                                 # net.py:1715
        if resnet_like:
            resY = self.invBN(resY, Y_name)                                         # net.py:1716-1718
        Xd, N, C, k = self._gather_patches_device(X_name, Y_name, relu=not resnet_like)   # relu(newX) fused (net.py:1720)
        # Y = feats - bias (+ resY) on the device when the features are float32-representable
        # (they are: they were sampled from float32 blobs); otherwise assemble in float64 on the host.
        f32 = feats.astype(np.float32)
        Yd = ctx.empty(N * n * 8)
        if np.array_equal(f32.astype(np.float64), feats):
            fd, bd = ctx.to_device(f32), ctx.to_device(b2)
            rd = None if isinstance(resY, int) and resY == 0 else ctx.to_device(np.ascontiguousarray(resY, dtype=np.float64))
            ctx.assemble_y(fd, bd, rd, N, n, Yd)
        else:
            ctx.to_device(np.ascontiguousarray(feats - b2 + resY, dtype=np.float64), Yd)
        prob = LayerProblem.from_device(ctx, Xd, _capi.CP_F32, N, C, k, W2, Yd, flags=_decompose._flags())
        try:
            return _decompose.prune_resident(prob, d_prime, W2)
        finally:
            prob.free()
            Xd.free()
            Yd.free()

    # ---- write-back bookkeeping of the layer-by-layer drivers (net.py:1521-1630) ------------------------------
    def W1keep(self, conv, idxs):
        """Keep only the filters `idxs` of the PRODUCER behind blob `conv` (and the matching BatchNorm / Scale entries):
        WPQ gets the compact arrays, the live parameters are zeroed outside idxs (net.py:1521-1608)."""
        idxs = np.asarray(idxs, dtype=bool)
        if conv in (self.sums + self.pools):
            if dcfgs.model in [cfgs.Models.resnet] or self.model == cfgs.Models.resnet:
                sconv = None
                for i in self.convs:
                    if self.layer_bottom(i) == conv:
                        sconv = i
                self.bottoms2ch.append([conv, sconv, idxs])
                return
            conv = self.layer_bottom(conv)                       # vgg: the pooling layer's bottom
        bn = affine = None
        if conv in self.bns:
            bn = conv
            for i in self.affines:
                if self.layer_bottom(i) == bn:
                    affine = i
                    break
        if conv not in self.convs:
            conv = self.bottom_names[conv][0]
            if conv not in self.convs and conv in self._layer_bottom:      # e.g. a ReLU blob: its conv is one further up
                conv = self.layer_bottom(conv)
        else:
            for i in self.affines:
                if self.layer_bottom(i) == conv:
                    affine = i
                    break
            for i in self.bns:
                if self.layer_bottom(i) == conv:
                    bn = i
                    break
            if affine is None and bn is not None:      # Scale reading the BatchNorm's own top (non-in-place lay-out)
                for i in self.affines:
                    if self.layer_bottom(i) == bn:
                        affine = i
                        break
        W1 = self.param_data(conv)[idxs, ...]
        b = self.param_b_data(conv)[idxs]
        self.WPQ[(conv, 0)] = self.WPQ[(conv, 0)][idxs, ...].copy() if (conv, 0) in self.WPQ else W1.copy()
        self.WPQ[(conv, 1)] = self.WPQ[(conv, 1)][idxs].copy() if (conv, 1) in self.WPQ else b.copy()
        for extra in (bn, affine):
            if extra is not None:
                self.WPQ[(extra, 0)] = self.param_data(extra)[idxs].copy()
                self.WPQ[(extra, 1)] = self.param_b_data(extra)[idxs].copy()
        Wl, bl = self.param_data(conv), self.param_b_data(conv)
        Wl[~idxs, ...] = 0.
        bl[~idxs] = 0.
        for extra in (bn, affine):
            if extra is not None:
                self.param_data(extra)[~idxs] = 0.
                self.param_b_data(extra)[~idxs] = 0.
        self._blob_cache = (None, None)
        self.set_conv(conv, num_output=len(self.WPQ[(conv, 1)]))

    def W2keep(self, top, idxs, W2, B2=None, layerbylayer=False):
        """The CONSUMER `top` keeps the input channels idxs with the refitted weights W2 (net.py:1610-1625); its bias
        becomes B2 + its old bias unless layerbylayer."""
        idxs = np.asarray(idxs, dtype=bool)
        Wt = self.param_data(top)
        Wt[:, ~idxs, ...] = 0
        Wt[:, idxs, ...] = W2
        self.WPQ[(top, 0)] = np.array(W2, copy=True)
        if B2 is not None:
            newB2 = np.array(B2, dtype=np.float64, copy=True)
            if (not layerbylayer) and (dcfgs.ls != cfgs.solvers.gd or not self._mem):
                newB2 += self.param_b_data(top)
            self.set_param_b(top, newB2)
            self.WPQ[(top, 1)] = newB2.copy()
        self._blob_cache = (None, None)

    def select(self, name, nextname, idxs):
        """A channel-selection ("Filter") layer between blob `name` and its consumer `nextname` (net.py:1627-1630; the
        reference inserts it into the prototxt, lib/builder.py:666-672): remembered in nonWPQ under the filter's name."""
        fname = name + '_Filter'                          # builder.py:315-319, 659-661: <bottom>_Filter (e.g. res2a_Filter)
        self.nonWPQ[fname] = np.asarray(idxs).astype(int)
        self._layer_bottom[nextname] = fname
        return fname

    def combineHP(self):
        """Fold the 1x1 conv_P back into conv_H where that is cheaper, 3 m >= 2 o (net.py:1473-1504): operates on the
        decomposed layers held in WPQ after R3()."""
        H = [k[0] for k in self.WPQ if isinstance(k, tuple) and k[1] == 0 and k[0].endswith('_H')]
        merged = []
        for h in H:
            pname = h[:-2] + '_P'
            if (pname, 0) not in self.WPQ:
                continue
            Hw_full, Pw_full = np.asarray(self.WPQ[(h, 0)]), np.asarray(self.WPQ[(pname, 0)])
            m, o = Hw_full.shape[0], Pw_full.shape[0]
            if 3 * m >= 2 * o:
                newshape = list(Hw_full.shape)
                newshape[0] = o
                Hw, Pw = Hw_full.reshape((m, -1)), Pw_full.reshape((o, -1))
                Hb, pb = np.asarray(self.WPQ[(h, 1)]), np.asarray(self.WPQ[(pname, 1)])
                self.WPQ[(h, 0)] = Pw.dot(Hw).reshape(newshape)
                self.WPQ[(h, 1)] = pb + Pw.dot(Hb)
                self.set_conv(h, num_output=o)
                del self.WPQ[(pname, 0)], self.WPQ[(pname, 1)]
                self.removed.append(pname)
                merged.append(h)
        return merged

    # ---- ResNet driver (reconstructed: the reference ships the helpers, not the loop) ------------------------------
    def resnet_blocks(self):
        """[(block, branch2a, branch2b, branch2c)] of the bottlenecks in network order (blocks = the Eltwise sums)"""
        out = []
        for blk in self.sums:
            names = [blk + '_branch2' + t for t in 'abc']
            if all(n in self.convs for n in names):
                out.append((blk,) + tuple(names))
        return out

    def _producer_handle(self, blob):
        """What W1keep() is called with for the producer of `blob`: the BatchNorm behind it when there is one (W1keep
        then finds the conv and the Scale: net.py:1547-1569), else the conv itself; ReLU layers are looked through."""
        name = blob
        while name in self.relus or (name not in self.convs and name not in self.bns and name in self._layer_bottom
                                     and not isinstance(self._layer_bottom[name], list)
                                     and name not in self.sums and name not in self.pools):
            name = self._layer_bottom[name]
        return name

    def prune_resnet(self, keep, layerbylayer=False):
        """Layer-by-layer channel pruning of a bottleneck ResNet -- the loop the reference's helpers W1keep / W2keep /
        select / appresb / invBN (net.py:1521-1683, 1200-1217) were written for and its release leaves out (SURVEY.md
        section 2, component 12), reconstructed from them and from the released ResNet-50 2x model
        (temp/resnet-50-cp.prototxt: a Filter layer in front of every branch2a, pruned branch2a / branch2b outputs).

        keep: {consumer conv: number of its INPUT channels to keep}.  Per bottleneck, in network order:
          branch2a  reads the block input, which the shortcut shares: its producer keeps all filters and a channel
                    sampler is put in front of branch2a (select), whose weights are refitted on the sampled channels;
          branch2b  prunes branch2a's filters (W1keep on its BatchNorm / Scale / conv) and refits branch2b (W2keep);
          branch2c  prunes branch2b's filters against the residual-aware target Y + (frozen - current shortcut)
                    (appresb + invBN inside dictionary_kernel; dcfgs.res.short = 1, dic.option = resnet).
        Needs frozen features (freeze_images / load_frozen) and, for the residual term to see the pruned earlier layers,
        a live provider.  Returns (WPQ, nonWPQ): WPQ as W1keep / W2keep fill it, nonWPQ the samplers' index masks."""
        if not self._mem:
            raise ValueError("prune_resnet needs the frozen features of the original network: call freeze_images() first")
        saved = (dcfgs.model, dcfgs.res.short, dcfgs.dic.option)
        dcfgs.model, dcfgs.res.short, dcfgs.dic.option = cfgs.Models.resnet, 1, cfgs.pruning_options.resnet
        t = Timer()
        try:
            for blk, b2a, b2b, b2c in self.resnet_blocks():
                for consumer in (b2a, b2b, b2c):
                    if consumer not in keep:
                        continue
                    t.tic()
                    X_name = self.bottom_names[consumer][0]
                    idxs, W2, B2 = self.dictionary_kernel(X_name, None, int(keep[consumer]), consumer, None)
                    if consumer == b2a:
                        self.select(X_name, consumer, idxs)
                    else:
                        self.W1keep(self._producer_handle(X_name), idxs)
                    self.W2keep(consumer, idxs, W2, B2, layerbylayer=layerbylayer)
                    self.selection[consumer] = idxs
                    t.toc('channel_pruning')
        finally:
            dcfgs.model, dcfgs.res.short, dcfgs.dic.option = saved
        return self.WPQ, self.nonWPQ

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
