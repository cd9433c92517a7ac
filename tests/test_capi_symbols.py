"""CPU: the C-ABI library loads and exports every symbol include/cpmi355.h declares
(no compute call is made here)."""
import os
import re

from conftest import ROOT


def header_symbols():
    text = open(os.path.join(ROOT, "include", "cpmi355.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cp_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_hot_path():
    syms = header_symbols()
    for needed in ("cp_patch_gather", "cp_assemble_y", "cp_lasso_gram", "cp_enet_cd_gram",
                   "cp_lasso_alpha_search", "cp_lstsq_refit", "cp_ctx_create"):
        assert needed in syms


def test_library_exports_every_declared_symbol():
    from cpmi355 import capi
    lib = capi.load()
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), "libcpmi355.so does not export %s" % s
    # and the ctypes table binds exactly the header's functions
    assert sorted(capi.SIGNATURES) == syms
    assert lib.cp_version() >= 100
    assert lib.cp_strerror(-5).decode() == "numerical breakdown"


def test_missing_gpu_fails_loudly():
    """No silent CPU path: without a gfx950 device, context creation raises."""
    import pytest
    from cpmi355 import capi
    if capi.device_count() > 0:
        pytest.skip("a GPU is visible here")
    with pytest.raises(capi.CpError):
        capi.Context(0)
