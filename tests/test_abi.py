"""CPU checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/muzero_amd.h declares; host-only entry points validate arguments; no compute is launched."""
import ctypes as C
import os
import re

import pytest


def header_symbols():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    txt = open(os.path.join(root, "include", "muzero_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mz_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from muzero_amd import _lib as L
    names = header_symbols()
    assert len(names) >= 17
    for n in names:
        assert hasattr(L.lib, n), n
    assert sorted(names) == L.EXPORTS  # the python binding covers exactly the header
    assert L.lib.mz_version() >= 1


def test_layout_is_host_only_and_validates():
    from muzero_amd import _lib as L
    lay = L.PoolLayout()
    L.call("mz_pool_layout_compute", 4096, 51, L.MZ_F32, C.byref(lay))
    nodes = 4096 * 51
    assert lay.off_stats == 0 and lay.off_hidden >= nodes * 65 * 16
    assert lay.off_meta - lay.off_hidden >= nodes * 512 * 4
    assert lay.total_bytes % 256 == 0 and lay.hidden_elem_bytes == 4
    L.call("mz_pool_layout_compute", 8, 101, L.MZ_BF16, C.byref(lay))
    assert lay.hidden_elem_bytes == 2
    for bad in ((0, 51, 0), (4, 1, 0), (4, 70000, 0), (4, 51, 9)):
        with pytest.raises(L.MzError):
            L.call("mz_pool_layout_compute", *bad, C.byref(lay))
    assert b"dtype" in L.lib.mz_last_error()


def test_null_and_range_arguments_are_rejected_without_launching():
    from muzero_amd import _lib as L
    with pytest.raises(L.MzError):
        L.call("mz_env_step", *([None] * 9), 4, None)
    with pytest.raises(L.MzError):
        L.call("mz_env_reset", *([None] * 6), -1, None)
    L.call("mz_env_reset", *([None] * 6), 0, None)  # empty batch is a no-op even with null pointers
    with pytest.raises(L.MzError):
        L.call("mz_env_random_playout", 1, 0, 4, 300, 0, *([None] * 9), None)


def test_product_never_touches_the_oracle():
    """the product package must not import, link or shell out to oracle/ (prompt rule 3)"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dp, _, files in os.walk(os.path.join(root, "muzero_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(dp, f)).read()
                assert "oracle" not in txt.lower(), os.path.join(dp, f)


def test_host_tensors_are_refused():
    import torch
    from muzero_amd import _lib as L
    with pytest.raises(L.MzError):
        L.dptr(torch.zeros(4))
