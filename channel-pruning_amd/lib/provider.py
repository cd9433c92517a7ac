"""Activation providers for the caffe-free ``Net`` (lib/net.py): what stands in for the reference's Caffe forward pass
(``self.net.forward()`` in lib/net.py:197, 395-401, 627) when features are extracted.

A provider is a callable ``provider(batch, net) -> {blob_name: float32[B, C, H, W]}``; taking ``net`` makes it LIVE (it
computes with the weights the net holds NOW, which the full 3C loop of ``Net.R3`` needs because it re-extracts
features after every decomposition step, exactly as the reference re-runs its modified Caffe net).

``TorchSequentialProvider`` covers plain VGG-style stacks: convolutions in network order, an in-place ReLU after each
(its output is exposed as ``<conv>_relu``; the conv blob itself is the pre-ReLU response, as the reference arranges
by splitting the in-place ReLUs, lib/net.py:1106-1133) and optional max-pooling after named convs.  It runs on torch
(CPU by default; pass ``device="cuda"`` only in a process that imported torch BEFORE the first cpmi355 Context, see
cpmi355/capi.py::load).  torch is plumbing here: the forward pass is not on the accelerated path.
"""
import numpy as np

"""Layer-description format shared by ``Net(graph=...)`` and ``TorchGraphProvider`` (one dict per layer, network order):
    {"name", "type", "bottom": [blob, ..], "top": [blob], ...}
      Convolution  W float32[n, c, k, k], b float32[n], pad, stride
      ReLU | Pooling (kernel, stride; max) | Eltwise (sum of two bottoms)
      BatchNorm    mean[c], var[c], eps          (Caffe blobs 0 and 1 with scale factor 1)
      Scale        k[c], b[c]
A top equal to the bottom means "in place" (Caffe's in-place ReLU / Scale): the blob is overwritten."""


class TorchSequentialProvider(object):
    def __init__(self, batches, pools=None, device="cpu", num_threads=None):
        """batches: list of float32 arrays [B, C, H, W] (the frozen images, lib/net.py:749-802);
        pools: {conv_name: (blob_name, kernel, stride)} max-pooling applied to relu(conv)."""
        import torch
        self.torch = torch
        self.batches = [np.ascontiguousarray(b, dtype=np.float32) for b in batches]
        self.pools = dict(pools or {})
        self.device = torch.device(device)
        if num_threads is not None:
            torch.set_num_threads(int(num_threads))

    def __len__(self):
        return len(self.batches)

    def __call__(self, batch, net):
        torch = self.torch
        F = torch.nn.functional
        blobs = {"data": torch.from_numpy(self.batches[batch]).to(self.device)}
        for name in net.convs:
            cv = net.layers[name]
            if cv.bottom not in blobs:
                raise KeyError("conv %r reads blob %r, which no earlier layer produced" % (name, cv.bottom))
            W = torch.from_numpy(net.param_data(name)).to(self.device)
            b = torch.from_numpy(net.param_b_data(name)).to(self.device)
            y = F.conv2d(blobs[cv.bottom], W, b, stride=cv.stride, padding=cv.pad)
            blobs[name] = y
            r = F.relu(y)
            blobs[name + "_relu"] = r
            if name in self.pools:
                pname, kernel, stride = self.pools[name]
                blobs[pname] = F.max_pool2d(r, kernel, stride)
        return dict((k, v.detach().cpu().numpy()) for k, v in blobs.items())


class TorchGraphProvider(object):
    """Live provider for arbitrary layer graphs (ResNet blocks: BatchNorm / Scale / Eltwise shortcuts) on torch --
    ``device="cuda"`` runs the forward pass on the MI355X through torch-ROCm (import torch BEFORE the first cpmi355
    Context, see cpmi355/capi.py::load), ``"cpu"`` anywhere.  Parameters are read from the net on every call
    (``net.param_data`` / ``param_b_data``: conv weights, BatchNorm mean / variance, Scale k / b), so the blobs follow
    whatever the pruning steps wrote back.  ``set_batches`` re-points it at the images of a frozen pickle."""

    def __init__(self, layers, batches, labels=None, device="cpu", num_threads=None):
        import torch
        self.torch = torch
        self.layers = list(layers)
        self.device = torch.device(device)
        if num_threads is not None:
            torch.set_num_threads(int(num_threads))
        self.set_batches(batches, labels)

    def set_batches(self, batches, labels=None):
        self.batches = [np.ascontiguousarray(b, dtype=np.float32) for b in batches]
        self.labels = None if labels is None else [np.asarray(v, dtype=np.float32) for v in labels]

    def __len__(self):
        return len(self.batches)

    def __call__(self, batch, net):
        torch = self.torch
        F = torch.nn.functional
        dev = self.device

        def t(a):
            return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)

        blobs = {"data": t(self.batches[batch])}
        for L in self.layers:
            kind, name = L["type"], L["name"]
            x = [blobs[b] for b in L["bottom"]]
            if kind == "Convolution":
                y = F.conv2d(x[0], t(net.param_data(name)), t(net.param_b_data(name)), stride=L.get("stride", 1),
                             padding=L.get("pad", 0))
            elif kind == "ReLU":
                y = F.relu(x[0])
            elif kind == "Pooling":
                y = F.max_pool2d(x[0], L["kernel"], L["stride"])
            elif kind == "Eltwise":
                y = x[0] + x[1]
            elif kind == "BatchNorm":
                mean, var = t(net.param_data(name)), t(net.param_b_data(name))
                y = (x[0] - mean[None, :, None, None]) / torch.sqrt(var + L.get("eps", 1e-5))[None, :, None, None]
            elif kind == "Scale":
                k, b = t(net.param_data(name)), t(net.param_b_data(name))
                y = x[0] * k[None, :, None, None] + b[None, :, None, None]
            else:
                raise ValueError("layer type %r" % kind)
            blobs[L["top"][0]] = y
        out = dict((k, v.detach().cpu().numpy()) for k, v in blobs.items())
        B = self.batches[batch].shape[0]
        out["label"] = self.labels[batch] if self.labels is not None else np.zeros((B, 1, 1, 1), dtype=np.float32)
        return out
