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
