"""TEST INFRASTRUCTURE ONLY -- loader for the *real* reference solver library.

Imports ``/root/reference/lib/decompose.py`` UNMODIFIED, under the private package
name ``_cpref`` so it can never shadow (or be shadowed by) this repository's own
drop-in ``lib`` package.  The reference needs three modules that are not installed in
this image (``easydict``, ``IPython``, ``termcolor``) and one scikit-learn symbol that
no longer exists (``RandomizedLasso``); tiny stand-ins are registered for those before
the import.  Nothing here is shipped: only ``oracle/gen_golden.py`` and
``oracle/validate_oracle.py`` call it, and only in the build container --
``/root/reference`` does not exist on the GPU box.

Reference entry points exposed (file:line in /root/reference):
  lib/decompose.py:386  dictionary()
  lib/decompose.py:636  fc_kernel()
  lib/cfgs.py:18        alpha   (module global, carried across layers)
  lib/cfgs.py:75        c.dic.rank_tol
"""
import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get("CP_REFERENCE_ROOT", "/root/reference")


def available():
    return os.path.isfile(os.path.join(REF_ROOT, "lib", "decompose.py"))


def _install_stubs():
    if "easydict" not in sys.modules:
        m = types.ModuleType("easydict")

        class EasyDict(dict):
            """attribute-access dict (enough of easydict for lib/cfgs.py)."""

            def __getattr__(self, k):
                try:
                    return self[k]
                except KeyError:
                    raise AttributeError(k)

            def __setattr__(self, k, v):
                self[k] = v

        m.EasyDict = EasyDict
        sys.modules["easydict"] = m
    if "IPython" not in sys.modules:
        m = types.ModuleType("IPython")
        m.embed = lambda *a, **k: None
        sys.modules["IPython"] = m
    if "termcolor" not in sys.modules:
        m = types.ModuleType("termcolor")
        m.colored = lambda s, *a, **k: s
        sys.modules["termcolor"] = m
    # The reference calls scipy.linalg.pinv(x, 1e-6) (decompose.py:151) -- the second positional argument was
    # `cond` (singular values below cond * largest are dropped) in the SciPy it was written for and is gone from the
    # installed one; `rtol` has exactly that meaning.  Only ITQ_decompose uses it.
    import scipy.linalg as sl
    if not getattr(sl.pinv, "_cp_shim", False):
        _orig_pinv = sl.pinv

        def pinv(a, cond=None, rcond=None, **kw):
            c = cond if cond is not None else rcond
            return _orig_pinv(a, rtol=c, **kw) if c is not None else _orig_pinv(a, **kw)

        pinv._cp_shim = True
        sl.pinv = pinv
    import sklearn.linear_model as lm

    if not hasattr(lm, "RandomizedLasso"):
        lm.RandomizedLasso = None  # imported by name at decompose.py:7, never used


_cached = None


def load():
    """Return (decompose_module, cfgs_module) of the real reference."""
    global _cached
    if _cached is not None:
        return _cached
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    _install_stubs()
    libdir = os.path.join(REF_ROOT, "lib")
    pkg = types.ModuleType("_cpref")
    pkg.__path__ = [libdir]
    sys.modules["_cpref"] = pkg
    # lib/utils.py does `import lib.cfgs as cfgs` (absolute): alias it to the same objects.
    mods = {}
    for name in ("cfgs", "worker", "utils", "decompose"):
        spec = importlib.util.spec_from_file_location(
            "_cpref." + name, os.path.join(libdir, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules["_cpref." + name] = mod
        if name == "cfgs":
            # satisfy utils.py's absolute import without importing any other `lib`
            prev_lib = sys.modules.get("lib")
            prev_cfgs = sys.modules.get("lib.cfgs")
            spec.loader.exec_module(mod)
            shim = types.ModuleType("lib")
            shim.__path__ = []
            shim.cfgs = mod
            mods["_shim"] = (shim, prev_lib, prev_cfgs)
        elif name == "utils":
            shim, prev_lib, prev_cfgs = mods["_shim"]
            sys.modules["lib"] = shim
            sys.modules["lib.cfgs"] = mods["cfgs"]
            try:
                spec.loader.exec_module(mod)
            finally:
                for key, prev in (("lib", prev_lib), ("lib.cfgs", prev_cfgs)):
                    if prev is None:
                        sys.modules.pop(key, None)
                    else:
                        sys.modules[key] = prev
        else:
            import contextlib
            import io
            with contextlib.redirect_stdout(io.StringIO()):  # "no lighting pack" print
                spec.loader.exec_module(mod)
        mods[name] = mod
    _cached = (mods["decompose"], mods["cfgs"])
    return _cached
