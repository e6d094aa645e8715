"""CPU tests of the drop-in boundary: libfbg_hip.so loads, exports every symbol include/fbg_hip.h
declares, and refuses to compute without a GPU (no silent fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "fbg_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(fbg_[a-z_0-9]+)\s*\(", src)))


def test_header_declares_the_seam():
    names = header_functions()
    for must in ("fbg_ctx_create", "fbg_ctx_destroy", "fbg_last_error", "fbg_elastic_f", "fbg_minmax_dp",
                 "fbg_repeatfree_v", "fbg_repeatfree_dp", "fbg_index_build", "fbg_scan_f", "fbg_scan_v"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from founderblockgraphs_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_functions():
        assert hasattr(L, name), f"{name} declared in include/fbg_hip.h but not exported"
    # and the Python mirror binds exactly the declared set
    assert sorted(_lib.SIGNATURES) == header_functions()


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import founderblockgraphs_amd as F
    with pytest.raises(F.FbgError) as ei:
        F.Engine(0)
    assert ei.value.code == 6 and "no CPU fallback" in str(ei.value)


def test_product_does_not_import_the_oracle():
    """The oracle is test infrastructure: nothing under founderblockgraphs_amd/ may reference it."""
    pkg = os.path.join(ROOT, "founderblockgraphs_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp", ".hpp", "Makefile")):
                text = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "pyoracle" not in text and "liboracle" not in text and "fbg_oracle" not in text, (dirpath, fn)


def test_every_option_key_is_documented_in_the_header():
    """The behaviour switches of fbg_set_option (the table g_opt_keys in csrc/ctx.hip) are part of the seam: a key the
    header does not explain is a key a maintainer of the reference cannot use."""
    import re
    src = open(os.path.join(ROOT, "founderblockgraphs_amd", "csrc", "ctx.hip")).read()
    table = src[src.index("static const OptKey g_opt_keys[] = {"):]
    table = table[:table.index("};")]
    keys = re.findall(r'\{"([a-z_0-9]+)", &FbgOptions::', table)
    assert len(keys) >= 20
    header = open(os.path.join(ROOT, "include", "fbg_hip.h")).read()
    missing = [k for k in keys if not re.search(r"\b%s\b" % re.escape(k), header)]
    assert not missing, missing
