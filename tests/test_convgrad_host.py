"""Host logic of muzero_amd/convgrad.py without a GPU: the wrappers refuse what the kernels cannot take (no CPU fallback --
host tensors raise), and the size constants agree with include/muzero_amd.h."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_size_constants_match_the_header():
    from muzero_amd import _lib as L
    from muzero_amd import convgrad as CG
    h = open(os.path.join(ROOT, "include", "muzero_amd.h")).read()

    def macro(name):
        expr = re.search(r"#define %s \((.*)\)" % name, h).group(1)
        assert re.fullmatch(r"[0-9 +*()]+", expr)
        return eval(expr)   # noqa: S307  (digits and + * only)
    assert macro("MZ_CONV3X3_PACKED_BYTES") == 2 * CG.PACKED_ELEMS
    assert macro("MZ_RESTOWER_X3_PACKED_BYTES") == L.MZ_RESTOWER_X3_PACKED_BYTES == 2 * CG.RESTOWER_X3_ELEMS
    assert macro("MZ_TOWER_X3_ENDS_BYTES") == L.MZ_TOWER_X3_ENDS_BYTES == 2 * CG.TOWER_X3_ENDS_ELEMS
    assert CG.STREAM_ELEMS == 76 * 4096 and CG.PACKED_ELEMS == 40 * 4096   # hi view = stages 0..39, lo view = stages 36..75


def test_host_tensors_are_refused():
    from muzero_amd import _lib as L
    from muzero_amd import convgrad as CG
    x = torch.zeros(2, 128, 8, 8)
    w = torch.zeros(128, 128, 3, 3)
    with pytest.raises(L.MzError):
        CG.conv3x3(x, torch.zeros(CG.PACKED_ELEMS, dtype=torch.bfloat16))
    with pytest.raises(L.MzError):
        CG.pack_conv(w)
    with pytest.raises(L.MzError):
        CG.pack_convs([w])
    with pytest.raises(L.MzError):
        CG.conv3x3_wgrad([x], [x], w)
    with pytest.raises(L.MzError):
        CG.restower_x3_pack([w] * 6)
    with pytest.raises(L.MzError):
        CG.restower_x3(x, torch.zeros(CG.RESTOWER_X3_ELEMS, dtype=torch.bfloat16), torch.zeros(6, 128))
    with pytest.raises(L.MzError):
        CG.tower_x3_pack_ends(torch.zeros(128, 10, 3, 3), torch.zeros(8, 128, 3, 3))
    with pytest.raises(L.MzError):
        CG.tower_x3(torch.zeros(2, 10, 8, 8), torch.zeros(CG.RESTOWER_X3_ELEMS, dtype=torch.bfloat16),
                    torch.zeros(CG.TOWER_X3_ENDS_ELEMS, dtype=torch.bfloat16), torch.zeros(6, 128))


def test_split_precision_stream_views():
    """pack_conv(split=True)'s four images are views of two [hi 36 | lo 36 | pad 4]-stage streams: the lo image starts 36 stages
    behind the hi image, which is what lets one launch walk both (mz_conv3x3_fused x_part = 2)"""
    from muzero_amd import convgrad as CG
    buf = torch.zeros((2, CG.STREAM_ELEMS), dtype=torch.bfloat16)
    fh, bh, fl, bl = CG._stream_views(buf)
    for hi, lo in ((fh, fl), (bh, bl)):
        assert hi.numel() == lo.numel() == CG.PACKED_ELEMS and hi.is_contiguous() and lo.is_contiguous()
        assert lo.data_ptr() == hi.data_ptr() + 36 * 4096 * 2
    assert bh.data_ptr() == fh.data_ptr() + CG.STREAM_ELEMS * 2
