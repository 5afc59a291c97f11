"""Host side of the training step's hand-written convolution kernels (muzero_amd/csrc/train_kernels.hip).

What autograd computes in loss.backward() (main.py:265-266) for the 3x3, 128 -> 128 convolutions of the two towers
(network.py:35-47), on the matrix cores: operands rounded to bf16, sums in fp32.  No CPU fallback: host tensors raise."""
import ctypes as C

import torch

from . import _lib as L

_workspaces = {}


def _workspace(device, nbytes):
    """the weight gradient's partial-sum scratch: ONE buffer per device, allocated once at the size the largest call needs
    (the slice count is capped, so that is 75.6 MB) and never replaced -- a caller's captured hipGraph may hold its
    address.  Calls on one device share it: weight-gradient calls that overlap in time on two streams would race."""
    ws = _workspaces.get(device)
    if ws is None:
        ws = torch.empty(max(int(nbytes), int(L.lib.mz_conv3x3_wgrad_workspace(1 << 24))), dtype=torch.uint8, device=device)
        _workspaces[device] = ws
    if ws.numel() < nbytes:
        raise L.MzError("conv3x3_wgrad: workspace of %d bytes, %d needed" % (ws.numel(), nbytes))
    return ws


def _check(t, name):
    if not t.is_cuda or t.dtype != torch.float32 or not t.is_contiguous() or t.dim() != 4 or tuple(t.shape[1:]) != (128, 8, 8):
        raise L.MzError("%s: need contiguous cuda float32 tensors of shape (n, 128, 8, 8), got %s %s on %s" % (name, tuple(t.shape), t.dtype, t.device))


def conv3x3_wgrad(xs, gs, dw, db=None, accumulate=False, x3=False):
    """dw (128,128,3,3) (+)= weight gradient, db (128) (+)= bias gradient of one 3x3 / padding 1 convolution summed over its uses:
    xs[k] = the k-th use's input, gs[k] = the gradient of its output (mz_conv3x3_wgrad).  x3: split precision (every product as
    g_hi x_hi + g_lo x_hi + g_hi x_lo with hi = bf16(v), lo = bf16(v - hi): close to fp32; mz_conv3x3_wgrad_x3)."""
    if len(xs) != len(gs) or not 1 <= len(xs) <= 8:
        raise L.MzError("conv3x3_wgrad: 1..8 (input, grad_output) pairs, got %d / %d" % (len(xs), len(gs)))
    for x, g in zip(xs, gs):
        _check(x, "conv3x3_wgrad input")
        _check(g, "conv3x3_wgrad grad_output")
        if x.shape[0] != g.shape[0] or x.device != dw.device:
            raise L.MzError("conv3x3_wgrad: mismatched use")
    if dw.dtype != torch.float32 or tuple(dw.shape) != (128, 128, 3, 3) or not dw.is_contiguous():
        raise L.MzError("conv3x3_wgrad: dw must be a contiguous float32 (128,128,3,3) tensor")
    if db is not None and (db.dtype != torch.float32 or tuple(db.shape) != (128,) or not db.is_contiguous()):
        raise L.MzError("conv3x3_wgrad: db must be a contiguous float32 (128,) tensor")
    k = len(xs)
    total = sum(int(x.shape[0]) for x in xs)
    need = int(L.lib.mz_conv3x3_wgrad_workspace(total))
    ws = _workspace(dw.device, need)
    xp = (C.c_void_p * k)(*[x.data_ptr() for x in xs])
    gp = (C.c_void_p * k)(*[g.data_ptr() for g in gs])
    nb = (C.c_int64 * k)(*[int(x.shape[0]) for x in xs])
    L.call("mz_conv3x3_wgrad_x3" if x3 else "mz_conv3x3_wgrad", xp, gp, nb, k, L.dptr(dw), L.dptr(db), 1 if accumulate else 0, L.dptr(ws), ws.numel(), L.current_stream())
    return dw, db


def conv3x3_wgrad_unchecked(st, xs, gs, dw, db, x3=False):
    """conv3x3_wgrad(accumulate=True) for callers that have already checked their tensors (train.DeferredWeightGrads.flush:
    contiguous (n,128,8,8) f32 pairs from _TowerHip), on the stream handle st"""
    k = len(xs)
    nb = [x.shape[0] for x in xs]
    ws = _workspace(dw.device, int(L.lib.mz_conv3x3_wgrad_workspace(sum(nb))))
    rc = (L.lib.mz_conv3x3_wgrad_x3 if x3 else L.lib.mz_conv3x3_wgrad)(
        (C.c_void_p * k)(*[x.data_ptr() for x in xs]), (C.c_void_p * k)(*[g.data_ptr() for g in gs]), (C.c_int64 * k)(*nb), k,
        dw.data_ptr(), None if db is None else db.data_ptr(), 1, ws.data_ptr(), ws.numel(), st)
    if rc != 0:
        raise L.MzError("mz_conv3x3_wgrad failed (%d): %s" % (rc, L.lib.mz_last_error().decode()))


PACKED_ELEMS = 40 * 4096   # MZ_CONV3X3_PACKED_BYTES / 2


def pack_conv(weight, out=None, split=False):
    """conv weight (128,128,3,3) f32 on the GPU -> (forward image, data-gradient image), bf16 (mz_conv3x3_pack); split=True:
    (forward, data-gradient, forward lo, data-gradient lo) -- the images of bf16(w) and of the remainder bf16(w - bf16(w))"""
    w = weight.detach()
    if not w.is_cuda or w.dtype != torch.float32 or tuple(w.shape) != (128, 128, 3, 3) or not w.is_contiguous():
        raise L.MzError("pack_conv: need a contiguous cuda float32 (128,128,3,3) weight")
    want = (2, STREAM_ELEMS if split else PACKED_ELEMS)
    if out is not None and (tuple(out.shape) != want or out.dtype != torch.bfloat16 or not out.is_contiguous() or out.device != w.device):
        raise L.MzError("pack_conv: out must be a contiguous bfloat16 %s tensor on the weight's device" % (want,))
    if not split:
        if out is None:
            out = torch.empty((2, PACKED_ELEMS), dtype=torch.bfloat16, device=w.device)
        L.call("mz_conv3x3_pack", L.dptr(w), L.dptr(out[0]), L.dptr(out[1]), L.current_stream())
        return out[0], out[1]
    if out is None:
        out = torch.empty((2, STREAM_ELEMS), dtype=torch.bfloat16, device=w.device)
    views = _stream_views(out)
    L.call("mz_conv3x3_pack", L.dptr(w), L.dptr(views[0]), L.dptr(views[1]), L.current_stream())          # hi, + 4 zero stages ...
    L.call("mz_conv3x3_pack_part", L.dptr(w), L.dptr(views[2]), L.dptr(views[3]), 1, L.current_stream())   # ... that the lo image overwrites
    return views


STREAM_ELEMS = 76 * 4096   # split precision: [36 stages hi][36 stages lo][4 stages of padding] per direction


def _stream_views(buf):
    """(2, STREAM_ELEMS) -> (forward hi, data-gradient hi, forward lo, data-gradient lo), 40-stage views: a hi view's padding
    stages are the lo image's first stages (prefetched, never used), and hi + lo together are the stream one split-precision
    launch walks (mz_conv3x3_fused x_part = 2)"""
    return buf[0, :PACKED_ELEMS], buf[1, :PACKED_ELEMS], buf[0, 36 * 4096:], buf[1, 36 * 4096:]


def pack_convs(weights, split=False):
    """pack_conv for up to 16 weights in one launch (two with split=True): a list of pack_conv's result tuples"""
    ws = [w.detach() for w in weights]
    if not 1 <= len(ws) <= 16:
        raise L.MzError("pack_convs: 1..16 weights")
    for w in ws:
        if not w.is_cuda or w.dtype != torch.float32 or tuple(w.shape) != (128, 128, 3, 3) or not w.is_contiguous():
            raise L.MzError("pack_convs: need contiguous cuda float32 (128,128,3,3) weights")
    k = len(ws)
    wp = (C.c_void_p * k)(*[w.data_ptr() for w in ws])
    if split:
        out = torch.empty((k, 2, STREAM_ELEMS), dtype=torch.bfloat16, device=ws[0].device)
        res = [_stream_views(out[i]) for i in range(k)]
    else:
        out = torch.empty((k, 2, PACKED_ELEMS), dtype=torch.bfloat16, device=ws[0].device)
        res = [(out[i, 0], out[i, 1]) for i in range(k)]
    for part in ((0, 1) if split else (0,)):   # hi first: its zero padding stages are where the lo image starts
        fp = (C.c_void_p * k)(*[r[2 * part].data_ptr() for r in res])
        bp = (C.c_void_p * k)(*[r[2 * part + 1].data_ptr() for r in res])
        L.call("mz_conv3x3_pack_many", wp, fp, bp, k, part, L.current_stream())
    return res


def conv3x3(x, packed, bias=None, relu=False, out=None, skip=None, mask=None, packed_lo=None, three_launches=False):
    """y = [relu]( [mask > 0 ?] (conv3x3(x) + bias [+ skip]) : 0 ) with the packed weight image (forward image: the convolution;
    data-gradient image applied to a grad_output: the input gradient).  x, skip, mask (n,128,8,8) f32 NCHW (mz_conv3x3_fused).
    packed_lo (the matching lo image of pack_conv(split=True)): split precision -- ONE launch that forms the three products
    W_hi x_hi + W_hi x_lo + W_lo x_hi from a hi and a lo image of x in LDS when the two images are pack_conv's stream (lo
    36 stages behind hi), else three launches accumulating in y."""
    _check(x, "conv3x3 input")
    for t, name in ((skip, "skip"), (mask, "mask")):
        if t is not None:
            _check(t, "conv3x3 " + name)
            if t.shape != x.shape:
                raise L.MzError("conv3x3: %s has another batch size" % name)
    for pk in (packed, packed_lo):
        if pk is not None and (pk.dtype != torch.bfloat16 or pk.numel() != PACKED_ELEMS or not pk.is_cuda or not pk.is_contiguous()):
            raise L.MzError("conv3x3: packed must be one of pack_conv's images")
    if bias is not None and (bias.dtype != torch.float32 or tuple(bias.shape) != (128,) or not bias.is_cuda):
        raise L.MzError("conv3x3: bias must be a cuda float32 (128,) tensor")
    y = torch.empty_like(x) if out is None else out
    n, st, b = int(x.shape[0]), L.current_stream(), L.dptr(bias.detach() if bias is not None else None)
    if packed_lo is None:
        L.call("mz_conv3x3_fused", L.dptr(packed), b, L.dptr(x), 0, L.dptr(skip), L.dptr(mask), L.dptr(y), n, 1 if relu else 0, st)
    elif packed_lo.data_ptr() == packed.data_ptr() + 36 * 4096 * 2 and not three_launches:
        L.call("mz_conv3x3_fused", L.dptr(packed), b, L.dptr(x), 2, L.dptr(skip), L.dptr(mask), L.dptr(y), n, 1 if relu else 0, st)
    else:
        L.call("mz_conv3x3_fused", L.dptr(packed), b, L.dptr(x), 0, L.dptr(skip), None, L.dptr(y), n, 0, st)
        L.call("mz_conv3x3_fused", L.dptr(packed), None, L.dptr(x), 1, L.dptr(y), None, L.dptr(y), n, 0, st)
        L.call("mz_conv3x3_fused", L.dptr(packed_lo), None, L.dptr(x), 0, L.dptr(y), L.dptr(mask), L.dptr(y), n, 1 if relu else 0, st)
    return y


def conv3x3_unchecked(st, x, packed, bias=None, relu=False, skip=None, mask=None, x3=False):
    """conv3x3 for callers that have already checked their tensors (train._TowerHip: 66 calls per training step, where the
    checks and the stream lookup of conv3x3 cost more host time than the launch): st = L.current_stream() taken once by the
    caller, packed = a forward or data-gradient image of pack_conv, x3 = split precision in one launch (packed is then the hi
    image of pack_conv(split=True), whose lo image follows 36 stages behind it)."""
    y = torch.empty_like(x)
    rc = L.lib.mz_conv3x3_fused(packed.data_ptr(), None if bias is None else bias.data_ptr(), x.data_ptr(), 2 if x3 else 0,
                                None if skip is None else skip.data_ptr(), None if mask is None else mask.data_ptr(), y.data_ptr(),
                                x.shape[0], 1 if relu else 0, st)
    if rc != 0:
        raise L.MzError("mz_conv3x3_fused failed (%d): %s" % (rc, L.lib.mz_last_error().decode()))
    return y


RESTOWER_X3_ELEMS = L.MZ_RESTOWER_X3_PACKED_BYTES // 2


def restower_x3_pack(weights, out=None):
    """six (128,128,3,3) f32 convolution weights (a tower's three residual blocks, in order) -> the weight stream of
    restower_x3 (mz_restower_x3_pack); out: an earlier result to refill in place"""
    ws = [w.detach().contiguous() for w in weights]
    if len(ws) != 6 or any((not w.is_cuda) or w.dtype != torch.float32 or tuple(w.shape) != (128, 128, 3, 3) for w in ws):
        raise L.MzError("restower_x3_pack: six cuda float32 (128,128,3,3) weights")
    if out is None:
        out = torch.empty(RESTOWER_X3_ELEMS, dtype=torch.bfloat16, device=ws[0].device)
    elif out.dtype != torch.bfloat16 or out.numel() != RESTOWER_X3_ELEMS or not out.is_contiguous() or out.device != ws[0].device:
        raise L.MzError("restower_x3_pack: out must come from an earlier call")
    L.call("mz_restower_x3_pack", (C.c_void_p * 6)(*[w.data_ptr() for w in ws]), L.dptr(out), L.current_stream())
    return out


def restower_x3(x, packed, bias, relu_in=False, out=None):
    """y = three residual blocks relu(h + conv(relu(conv(h)))) (network.py:35-47) applied to x (n,128,8,8) f32, in split
    precision, as ONE LDS-resident launch (mz_restower_x3_forward).  bias: (6,128) f32; relu_in: x = relu(x) first; out may be x."""
    _check(x, "restower_x3 input")
    if out is None:
        out = torch.empty_like(x)
    _check(out, "restower_x3 output")
    if out.shape != x.shape or out.device != x.device or packed.device != x.device or bias.device != x.device:
        raise L.MzError("restower_x3: mismatched tensors")
    if packed.dtype != torch.bfloat16 or packed.numel() != RESTOWER_X3_ELEMS or not packed.is_contiguous():
        raise L.MzError("restower_x3: packed must come from restower_x3_pack")
    if bias.dtype != torch.float32 or tuple(bias.shape) != (6, 128) or not bias.is_contiguous():
        raise L.MzError("restower_x3: bias must be a contiguous float32 (6,128) tensor")
    L.call("mz_restower_x3_forward", L.dptr(packed), L.dptr(bias), L.dptr(x), L.dptr(out), x.shape[0], 1 if relu_in else 0, L.current_stream())
    return out


TOWER_X3_ENDS_ELEMS = L.MZ_TOWER_X3_ENDS_BYTES // 2


def tower_x3_pack_ends(w_first, w_last, out=None):
    """first (128,cin,3,3) and last (8,128,3,3) convolution weights of a tower -> the fragment arrays of tower_x3
    (mz_tower_x3_pack_ends); out: an earlier result to refill in place"""
    wf, wl = w_first.detach().contiguous(), w_last.detach().contiguous()
    cin = wf.shape[1] if wf.dim() == 4 else 0
    if (not wf.is_cuda or wf.dtype != torch.float32 or wf.dim() != 4 or tuple(wf.shape) != (128, cin, 3, 3) or not 1 <= cin <= 16
            or wl.device != wf.device or wl.dtype != torch.float32 or tuple(wl.shape) != (8, 128, 3, 3)):
        raise L.MzError("tower_x3_pack_ends: cuda float32 weights of shape (128,cin<=16,3,3) and (8,128,3,3)")
    if out is None:
        out = torch.empty(TOWER_X3_ENDS_ELEMS, dtype=torch.bfloat16, device=wf.device)
    elif out.dtype != torch.bfloat16 or out.numel() != TOWER_X3_ENDS_ELEMS or not out.is_contiguous() or out.device != wf.device:
        raise L.MzError("tower_x3_pack_ends: out must come from an earlier call")
    L.call("mz_tower_x3_pack_ends", L.dptr(wf), cin, L.dptr(wl), L.dptr(out), L.current_stream())
    return out


def tower_x3(x, packed, ends, bias, b_first=None, b_last=None, out=None):
    """hidden (n,8,8,8) = relu(conv_last(blocks(relu(conv_first(x))))) for x (n,cin,8,8) f32: a whole tower (network.py:49-77 /
    :113-135) in split precision as ONE LDS-resident launch (mz_tower_x3_forward)"""
    if not x.is_cuda or x.dtype != torch.float32 or not x.is_contiguous() or x.dim() != 4 or tuple(x.shape[2:]) != (8, 8) or not 1 <= x.shape[1] <= 16:
        raise L.MzError("tower_x3: need a contiguous cuda float32 (n, cin <= 16, 8, 8) input, got %s %s on %s" % (tuple(x.shape), x.dtype, x.device))
    n = x.shape[0]
    if out is None:
        out = torch.empty((n, 8, 8, 8), dtype=torch.float32, device=x.device)
    if out.dtype != torch.float32 or tuple(out.shape) != (n, 8, 8, 8) or not out.is_contiguous() or out.device != x.device:
        raise L.MzError("tower_x3: out must be a contiguous float32 (n,8,8,8) tensor on the input's device")
    if packed.dtype != torch.bfloat16 or packed.numel() != RESTOWER_X3_ELEMS or ends.dtype != torch.bfloat16 or ends.numel() != TOWER_X3_ENDS_ELEMS:
        raise L.MzError("tower_x3: packed / ends must come from restower_x3_pack / tower_x3_pack_ends")
    if bias.dtype != torch.float32 or tuple(bias.shape) != (6, 128) or not bias.is_contiguous():
        raise L.MzError("tower_x3: bias must be a contiguous float32 (6,128) tensor")
    for t, k in ((b_first, 128), (b_last, 8)):
        if t is not None and (t.dtype != torch.float32 or tuple(t.shape) != (k,) or not t.is_contiguous() or t.device != x.device):
            raise L.MzError("tower_x3: b_first / b_last must be contiguous float32 (128,) / (8,) tensors")
    if packed.device != x.device or ends.device != x.device or bias.device != x.device:
        raise L.MzError("tower_x3: tensors on different devices")
    L.call("mz_tower_x3_forward", L.dptr(packed), L.dptr(ends), L.dptr(bias), L.dptr(b_first), L.dptr(b_last), x.shape[1], L.dptr(x), L.dptr(out), n, L.current_stream())
    return out
