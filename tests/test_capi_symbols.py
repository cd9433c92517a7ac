"""CPU: the C-ABI library loads and exports every symbol include/cpmi355.h declares
(no compute call is made here)."""
import os
import re

import pytest

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


def test_header_is_valid_c99_and_links_from_plain_c(tmp_path):
    """include/cpmi355.h compiles as C99 and a plain C program links against libcpmi355.so and calls the
    entry points that need no GPU (what a cgo / JNI style binding would do)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib_dir = os.path.join(root, "channel-pruning_amd", "cpmi355")
    if not os.path.isfile(os.path.join(lib_dir, "libcpmi355.so")):
        pytest.skip("library not built")
    src = tmp_path / "t.c"
    src.write_text('#include <stdio.h>\n#include <string.h>\n#include "cpmi355.h"\n'
                   'int main(void) {\n'
                   '  if (cp_version() != CP_VERSION) return 1;\n'
                   '  if (strlen(cp_strerror(CP_ERR_NODEVICE)) == 0) return 2;\n'
                   '  if (cp_ctx_destroy(0) != CP_OK) return 3;\n'
                   '  printf("ok %d\\n", cp_version());\n  return 0;\n}\n')
    exe = tmp_path / "t"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(root, "include"), str(src),
                           "-L", lib_dir, "-lcpmi355", "-Wl,-rpath," + lib_dir, "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.startswith("ok"), (out.returncode, out.stdout, out.stderr)


def test_cd_kernel_form_table():
    """cp_cd_kernel_form (no GPU needed): which coordinate-descent kernel a width runs -- the multi-CU team takes
    512 < c <= 2048 (c % 8 == 0) in sklearn's operation order (flags 0) and in the fast variant (flags 3), the one-workgroup
    team c % 8 == 0 up to 512, the one- / two-wave kernels everything else; 2048 channels is the library's limit."""
    from cpmi355 import capi
    lib = capi.load()
    form = lambda c, f=0: int(lib.cp_cd_kernel_form(c, f))
    assert [form(c) for c in (64, 256, 512)] == [2, 2, 2]
    assert [form(c) for c in (520, 1024, 1536, 2048)] == [3, 3, 3, 3] and form(2048, 3) == 3
    assert form(2040) == 3 and form(2044) == 0          # c % 8 != 0: one wavefront, 32 doubles per lane
    assert form(55) == 0 and form(256, 1) == 1 and form(1024, 1) == 0
    assert form(0) == -1 and form(2056) == -1 and form(4096) == -1
