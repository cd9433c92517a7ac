"""Test helper: activation provider for the caffe-free Net facade over oracle/portable_net.py -- the same bit-portable
forward pass the reference's net.py ran on when the n0x goldens were generated (oracle/gen_golden_net.py)."""
import numpy as np
import portable_net


class PortableProvider(object):
    def __init__(self, layers, batches):
        self.layers = layers
        self.set_batches(batches)

    def set_batches(self, batches, labels=None):
        self.batches = [np.ascontiguousarray(b, dtype=np.float32) for b in batches]

    def __call__(self, batch, net):
        params = {}
        for L in self.layers:
            if L["type"] in ("Convolution", "BatchNorm", "Scale"):
                params[L["name"]] = [net.param_data(L["name"]), net.param_b_data(L["name"])]
        blobs = portable_net.forward(self.layers, self.batches[batch], params)
        B = self.batches[batch].shape[0]
        blobs["label"] = np.full((B, 1, 1, 1), float(batch), dtype=np.float32)   # what the fake Data layer produced
        return blobs
