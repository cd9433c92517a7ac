"""TEST INFRASTRUCTURE ONLY -- runs the UNMODIFIED reference ``lib/net.py`` without Caffe (build container only).

``/root/reference/lib/net.py`` reaches Caffe only through thin accessors (``self.net.blobs[..].data``,
``self.net.params[..]``, ``self.net.forward()``, ``self.net.set_input_arrays``) and the prototxt through
``self.net_param.layer[name][0]``.  This loader registers stand-in modules for what the file imports but the image
lacks (``caffe``, ``caffe.proto.caffe_pb2`` -- pycaffe is an un-vendored fork, .gitmodules:3), imports the file as it
is (next to the already loaded ``decompose`` / ``utils`` / ``worker`` / ``cfgs`` of oracle/ref_loader.py) and builds a
reference ``Net`` object around a FakeCaffeNet whose forward pass is oracle/portable_net.py.  ``Net.__init__`` (prototxt
parsing, GPU set-up) is bypassed with ``object.__new__``; every METHOD under test runs unmodified:

    extract_features   lib/net.py:368-532      extract_XY        lib/net.py:534-684
    dictionary_kernel  lib/net.py:1685-1735    appresb / invBN   lib/net.py:1641-1683, 1200-1217
    R3                 lib/net.py:1292-1471    (insert / set_conv / save_pt -- prototxt surgery -- are stubbed)
"""
import importlib.util
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import portable_net  # noqa: E402
import ref_loader  # noqa: E402


class _Anything(object):
    """permissive stand-in for protobuf message classes referenced at import time (lib/builder.py:15)"""

    def __init__(self, *a, **k):
        pass

    def __getattr__(self, name):
        return _Anything()

    def __call__(self, *a, **k):
        return _Anything()


def _install_caffe_stubs():
    if "caffe" in sys.modules:
        return
    caffe = types.ModuleType("caffe")
    caffe.TEST, caffe.TRAIN = 1, 0
    caffe.set_mode_gpu = caffe.set_mode_cpu = lambda *a, **k: None
    caffe.set_device = lambda *a, **k: None
    caffe.Net = _Anything
    proto = types.ModuleType("caffe.proto")
    pb2 = types.ModuleType("caffe.proto.caffe_pb2")
    pb2.__getattr__ = lambda name: _Anything            # module-level __getattr__ (PEP 562)
    proto.caffe_pb2 = pb2
    caffe.proto = proto
    sys.modules.update({"caffe": caffe, "caffe.proto": proto, "caffe.proto.caffe_pb2": pb2})
    if "matplotlib.pyplot" not in sys.modules:
        try:
            import matplotlib
            matplotlib.use("Agg")
        except Exception:
            m = types.ModuleType("matplotlib")
            mp = types.ModuleType("matplotlib.pyplot")
            m.pyplot = mp
            sys.modules.update({"matplotlib": m, "matplotlib.pyplot": mp})


_net_mod = None


def load():
    """-> (reference net module, decompose module, cfgs module)"""
    global _net_mod
    D, cfgs = ref_loader.load()
    if _net_mod is not None:
        return _net_mod, D, cfgs
    _install_caffe_stubs()
    libdir = os.path.join(ref_loader.REF_ROOT, "lib")
    shim = types.ModuleType("lib")
    shim.__path__ = []
    shim.cfgs = cfgs
    prev = {k: sys.modules.get(k) for k in ("lib", "lib.cfgs")}
    sys.modules["lib"], sys.modules["lib.cfgs"] = shim, cfgs
    try:
        import contextlib
        import io
        for name in ("builder", "net"):
            spec = importlib.util.spec_from_file_location("_cpref." + name, os.path.join(libdir, name + ".py"))
            mod = importlib.util.module_from_spec(spec)
            sys.modules["_cpref." + name] = mod
            with contextlib.redirect_stdout(io.StringIO()):
                spec.loader.exec_module(mod)
    finally:
        for k, v in prev.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    _net_mod = sys.modules["_cpref.net"]
    return _net_mod, D, cfgs


# ---- a pycaffe-shaped network over portable_net ---------------------------------------------------------------
class FakeBlob(object):
    def __init__(self, data):
        self.data = data

    @property
    def num(self):
        return self.data.shape[0]

    @property
    def channels(self):
        return self.data.shape[1]

    @property
    def height(self):
        return self.data.shape[2]

    @property
    def width(self):
        return self.data.shape[3]

    @property
    def count(self):
        return self.data.size

    def reshape(self, *shape):
        self.data = np.zeros(shape, dtype=self.data.dtype)


class FakeCaffeNet(object):
    """blobs / params / forward / set_input_arrays / top_names / bottom_names of a pycaffe Net.  Without input
    arrays a forward consumes the next image batch of `batches` (a Data layer); set_input_arrays makes the next
    forward use the given arrays (a MemoryData layer, lib/net.py:420-421, 622-625)."""

    def __init__(self, layers, batches):
        self.layers = layers
        self.batches = batches
        self._next = 0
        self._pending = None
        self.params = {}
        for L in layers:
            if L["type"] == "Convolution":
                self.params[L["name"]] = [FakeBlob(L["W"].copy()), FakeBlob(L["b"].copy())]
            elif L["type"] == "BatchNorm":
                self.params[L["name"]] = [FakeBlob(L["mean"].copy()), FakeBlob(L["var"].copy())]
            elif L["type"] == "Scale":
                self.params[L["name"]] = [FakeBlob(L["k"].copy()), FakeBlob(L["b"].copy())]
        self.top_names = {L["name"]: list(L["top"]) for L in layers}
        self.bottom_names = {L["name"]: list(L["bottom"]) for L in layers}
        self.blobs = {}
        out = portable_net.forward(layers, batches[0], self._live_params())
        for k, v in out.items():
            self.blobs[k] = FakeBlob(v.copy())
        self.blobs["label"] = FakeBlob(np.zeros((batches[0].shape[0], 1, 1, 1), dtype=np.float32))

    def _live_params(self):
        return {k: [b.data for b in v] for k, v in self.params.items()}

    def set_input_arrays(self, data, labels):
        self._pending = (np.asarray(data, dtype=np.float32), np.asarray(labels, dtype=np.float32))

    def forward(self):
        if self._pending is not None:
            data, labels = self._pending
            self._pending = None
        else:
            data = self.batches[self._next % len(self.batches)]
            labels = np.full((data.shape[0], 1, 1, 1), float(self._next % len(self.batches)), dtype=np.float32)
            self._next += 1
        out = portable_net.forward(self.layers, data, self._live_params())
        for k, v in out.items():
            self.blobs[k].data = v
        self.blobs["label"].data = labels
        return {"accuracy@5": 0.0}


class _ConvParam(object):
    def __init__(self, L):
        self.pad = [L.get("pad", 0)]
        self.kernel_size = [int(L["W"].shape[-1])]
        self.stride = [L["stride"]] if L.get("stride", 1) != 1 else []
        self.num_output = int(L["W"].shape[0])
        self.group = 1


class _LayerMsg(object):
    def __init__(self, L):
        self.name, self.type = L["name"], L["type"]
        self.bottom, self.top = list(L["bottom"]), list(L["top"])
        if L["type"] == "Convolution":
            self.convolution_param = _ConvParam(L)


class _Bag(object):
    """attribute bag for protobuf messages that are only written to (the MemoryData edit of freeze_images)"""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        b = _Bag()
        object.__setattr__(self, name, b)
        return b

    def ClearField(self, name):
        pass

    def extend(self, *a):
        pass


class FakeNetParam(object):
    """the slice of lib/builder.py::Net the methods under test touch: .layer[name][0], type2names, layer_bottom"""

    def __init__(self, layers):
        from collections import OrderedDict
        self.layer = OrderedDict((L["name"], [_LayerMsg(L)]) for L in layers)
        data = _Bag()
        data.name, data.type = "data", "Data"
        self.layer["data"] = [data]
        self.net = _Bag()

    def type2names(self, layer_type):
        return [n for n, m in self.layer.items() if m[0].type == layer_type]

    def layer_bottom(self, name):
        b = self.layer[name][0].bottom
        return b[0] if len(b) == 1 else b

    def selector(self, bottom, top, shape, num_output):
        """lib/builder.py:666-672 inserts a Filter layer named <bottom>_Filter (builder.py:315-319, 659-661) between `bottom`
        and its consumer `top`; here only the name and the re-wiring of `top` are kept (Net.select stores the mask)."""
        fname = bottom + "_Filter"
        self.layer[top][0].bottom = [fname]
        self.filters = getattr(self, "filters", []) + [(fname, bottom, top, int(num_output))]
        return fname

    def rm_layer(self, name, inplace=False):
        """lib/builder.py: drop a layer from the prototxt (Net.remove -> combineHP); remembered only"""
        self.removed = getattr(self, "removed", []) + [name]


def make_reference_net(layers, batches):
    """A reference ``Net`` (unmodified class) around the fake pycaffe net; __init__ is bypassed (it parses a prototxt)."""
    R, D, cfgs = load()
    net = object.__new__(R.Net)
    net.net = FakeCaffeNet(layers, batches)
    net.net_param = FakeNetParam(layers)
    net.pt_dir = "temp/fake.prototxt"
    net.caffemodel_dir = "temp/fake.caffemodel"
    net.num = None
    net.prunedweights = 0
    net._layers = dict()
    net._bottom_names = None
    net._top_names = None
    net.data_layer = "data"
    net._mem = False
    net._accname = "accuracy@5"
    net.kernel = net.dictionary_kernel
    net.acc = []
    net._protocol = 4
    net._points_dict_name = cfgs._points_dict_name
    net.WPQ, net.nonWPQ, net.bottoms2ch, net.bnidx = {}, {}, [], []
    net.convs = net.type2names()
    net.spation_convs, net.nonsconvs = [], list(net.convs)
    net.relus = net.type2names(layer_type="ReLU")
    net.bns = net.type2names(layer_type="BatchNorm")
    net.affines = net.type2names(layer_type="Scale")
    net.pools = net.type2names(layer_type="Pooling")
    net.sums = net.type2names("Eltwise")
    net.innerproduct = net.type2names("InnerProduct")
    net._feats_dict, net._points_dict = dict(), dict()
    # prototxt surgery has no meaning without a prototxt: stubbed (these calls do not touch the numbers under test)
    net.insert = lambda *a, **k: None
    net.set_conv = lambda *a, **k: None
    net.save_pt = lambda *a, **k: "temp/3C4x_fake.prototxt"
    return net
