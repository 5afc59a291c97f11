"""The training step's hand-written convolution kernels against PyTorch's fp32 convolution gradients (autograd is what the
reference runs: main.py:265-266).  Operands are rounded to bf16 inside the kernels, sums are fp32: on inputs that ARE bf16
values the only difference is the summation order."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref_wgrad(xs, gs):
    X, G = torch.cat(xs).double(), torch.cat(gs).double()
    w = torch.zeros(128, 128, 3, 3, dtype=torch.float64, device=X.device)
    _, gw, gb = torch.ops.aten.convolution_backward(G, X, w, [128], [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, True])
    return gw, gb


def _bf16_values(t):
    return t.to(torch.bfloat16).float()


@pytest.mark.parametrize("boards", [[1], [3], [7, 2], [512], [5, 130, 64], [512] * 5])
def test_wgrad_matches_fp64_convolution_backward_on_bf16_values(boards):
    from muzero_amd.convgrad import conv3x3_wgrad
    g = torch.Generator(device="cuda").manual_seed(sum(boards))
    xs = [_bf16_values(torch.randn(n, 128, 8, 8, device="cuda", generator=g)).relu() for n in boards]
    gs = [_bf16_values(torch.randn(n, 128, 8, 8, device="cuda", generator=g) * 0.1) for n in boards]
    dw = torch.full((128, 128, 3, 3), 7.0, device="cuda")
    db = torch.full((128,), 7.0, device="cuda")
    conv3x3_wgrad(xs, gs, dw, db)
    gw, gb = _ref_wgrad(xs, gs)
    # products of bf16 values are exact in fp32; what differs from the fp64 sum is fp32 accumulation over n * 64 terms
    assert (dw.double() - gw).abs().max().item() <= 2e-5 * gw.abs().max().item() + 1e-6
    assert (db.double() - gb).abs().max().item() <= 2e-5 * gb.abs().max().item() + 1e-6
    dw0 = torch.empty_like(dw)
    conv3x3_wgrad(xs, gs, dw0)                      # a convolution without bias: no db
    assert torch.equal(dw0, dw)
    # accumulate: adds to what the outputs hold; deterministic: the same bits again
    dw2, db2 = dw.clone(), db.clone()
    conv3x3_wgrad(xs, gs, dw2, db2, accumulate=True)
    assert torch.equal(dw2, dw + dw) and torch.equal(db2, db + db)


def test_wgrad_on_fp32_values_is_within_bf16_rounding():
    from muzero_amd.convgrad import conv3x3_wgrad
    g = torch.Generator(device="cuda").manual_seed(5)
    xs = [torch.randn(256, 128, 8, 8, device="cuda", generator=g).relu() for _ in range(2)]
    gs = [torch.randn(256, 128, 8, 8, device="cuda", generator=g) for _ in range(2)]
    dw, db = torch.empty(128, 128, 3, 3, device="cuda"), torch.empty(128, device="cuda")
    conv3x3_wgrad(xs, gs, dw, db)
    gw, gb = _ref_wgrad(xs, gs)
    # each product carries two roundings of 2^-9 relative, independent over 32,768 terms: a few 1e-3 of the gradient's scale
    assert (dw.double() - gw).abs().max().item() <= 4e-3 * gw.abs().max().item()
    assert (dw.double() - gw).norm().item() <= 3e-3 * gw.norm().item()
    assert (db.double() - gb).abs().max().item() <= 1e-5 * gb.abs().max().item() + 1e-4   # summed from the fp32 values


def test_wgrad_refuses_what_it_cannot_do():
    from muzero_amd import _lib as L
    from muzero_amd.convgrad import conv3x3_wgrad
    x = torch.zeros(2, 128, 8, 8, device="cuda")
    dw = torch.zeros(128, 128, 3, 3, device="cuda")
    with pytest.raises(L.MzError):
        conv3x3_wgrad([x.cpu()], [x.cpu()], dw)
    with pytest.raises(L.MzError):
        conv3x3_wgrad([x] * 9, [x] * 9, dw)
    with pytest.raises(L.MzError):
        conv3x3_wgrad([x.half()], [x.half()], dw)


def _grads(net, batch, hip):
    from muzero_amd import train as T
    for p in net.parameters():
        p.grad = None
    dw = T.DeferredWeightGrads(hip=hip)
    pl, vl = T.muzero_loss_batched(net, batch, dw)
    (pl + vl).backward()
    dw.flush()
    return float(pl.detach()), float(vl.detach()), {k: p.grad.detach().clone() for k, p in net.named_parameters()}


def _autocast_grads(net, batch):
    from muzero_amd import train as T
    for p in net.parameters():
        p.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16):
        pl, vl = T.muzero_loss(net, batch)
    (pl.float() + vl.float()).backward()
    return {k: p.grad.detach().clone() for k, p in net.named_parameters()}


def test_training_gradients_with_hip_convolutions_match_the_fp32_step():
    """the whole K = 5 unrolled loss: every parameter's gradient with the towers' convolutions (forward, data and weight
    gradient) on the HIP kernels against the fp32 PyTorch/MIOpen step on the same weights and batch.  The difference is bf16
    operand rounding through 36 convolutions in a row; measured on MI355X (tools/train_grad_noise.py), per-tensor relative L2:
    B = 64: worst 5.3e-2, median 1.6e-2 (torch.autocast(bfloat16) on the same step: 7.1e-2 / 2.3e-2); B = 512: worst 3.4e-2,
    median 6.0e-3 (autocast: 1.7e-1 / 1.1e-2).  Bounds: absolute, and not worse than PyTorch's own bf16 path."""
    from muzero_amd.network import Network
    torch.manual_seed(1261)
    net = Network().to("cuda")
    net.train()
    B, K = 64, 5
    g = torch.Generator(device="cpu").manual_seed(3)
    batch = (torch.rand(B, 9, 8, 8, generator=g).round().cuda(), torch.rand(K, B, 2, 8, 8, generator=g).round().cuda(),
             (torch.rand(K + 1, B, generator=g) * 2 - 1).cuda(), torch.softmax(torch.rand(K + 1, B, 65, generator=g), -1).cuda())
    pl0, vl0, ref = _grads(net, batch, hip=False)
    pl1, vl1, got = _grads(net, batch, hip=True)
    amp = _autocast_grads(net, batch)
    assert abs(pl0 - pl1) <= 2e-3 * abs(pl0) and abs(vl0 - vl1) <= 2e-3 * abs(vl0)
    rel = lambda d: sorted(float((d[k].double() - ref[k].double()).norm() / ref[k].double().norm().clamp_min(1e-12)) for k in ref)  # noqa: E731
    mine, theirs = rel(got), rel(amp)
    print("per-tensor relative L2 gradient difference to fp32: HIP worst %.3g median %.3g; autocast worst %.3g median %.3g" %
          (mine[-1], mine[len(mine) // 2], theirs[-1], theirs[len(theirs) // 2]))
    assert mine[-1] <= 1e-1 and mine[len(mine) // 2] <= 3e-2
    assert mine[-1] <= 1.25 * theirs[-1] and mine[len(mine) // 2] <= 1.25 * theirs[len(theirs) // 2]
    assert all(torch.isfinite(v).all() for v in got.values())


@pytest.mark.parametrize("n", [1, 8, 13, 512, 1023, 1501, 2051])   # 2-board tiles (fewer than one 4-board tile per CU), 4-board tiles, 8-board tiles
def test_conv3x3_forward_and_data_gradient_match_fp64_on_bf16_values(n):
    from muzero_amd.convgrad import conv3x3, pack_conv
    g = torch.Generator(device="cuda").manual_seed(n)
    w = _bf16_values(torch.randn(128, 128, 3, 3, device="cuda", generator=g) * 0.03)
    bias = torch.randn(128, device="cuda", generator=g)
    x = _bf16_values(torch.randn(n, 128, 8, 8, device="cuda", generator=g)).relu()
    gy = _bf16_values(torch.randn(n, 128, 8, 8, device="cuda", generator=g))
    pf, pb = pack_conv(w)
    y = conv3x3(x, pf, bias)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), bias.double(), padding=1)
    assert (y.double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    yr = conv3x3(x, pf, bias, relu=True)
    assert torch.equal(yr, y.clamp_min(0))
    gx = conv3x3(gy, pb)
    refg = torch.ops.aten.convolution_backward(gy.double(), x.double(), w.double(), None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0]
    assert (gx.double() - refg).abs().max().item() <= 1e-5 * refg.abs().max().item()


def test_conv3x3_on_fp32_values_is_within_bf16_rounding():
    from muzero_amd.convgrad import conv3x3, pack_conv
    g = torch.Generator(device="cuda").manual_seed(11)
    w = torch.randn(128, 128, 3, 3, device="cuda", generator=g) * 0.03
    x = torch.randn(64, 128, 8, 8, device="cuda", generator=g)
    y = conv3x3(x, pack_conv(w)[0])
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    assert (y.double() - ref).norm().item() <= 4e-3 * ref.norm().item()


def test_split_precision_kernels_are_close_to_fp32():
    """bf16x3: hi = bf16(v), lo = bf16(v - hi), a*b = a_hi b_hi + a_lo b_hi + a_hi b_lo -- on arbitrary fp32 values the convolution,
    its data gradient and its weight gradient are two orders of magnitude closer to the fp64 result than the one-product form"""
    from muzero_amd.convgrad import conv3x3, conv3x3_wgrad, pack_conv
    g = torch.Generator(device="cuda").manual_seed(21)
    w = torch.randn(128, 128, 3, 3, device="cuda", generator=g) * 0.03
    bias = torch.randn(128, device="cuda", generator=g)
    x = torch.randn(37, 128, 8, 8, device="cuda", generator=g)
    gy = torch.randn(37, 128, 8, 8, device="cuda", generator=g)
    skip = torch.randn(37, 128, 8, 8, device="cuda", generator=g)
    pf, pb, pfl, pbl = pack_conv(w, split=True)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), bias.double(), padding=1)
    e1 = (conv3x3(x, pf, bias).double() - ref).norm().item() / ref.norm().item()
    e3 = (conv3x3(x, pf, bias, packed_lo=pfl).double() - ref).norm().item() / ref.norm().item()
    print("conv: relative L2 error bf16 %.3g, bf16x3 %.3g" % (e1, e3))
    assert e3 <= 2e-5 and e3 <= e1 / 50
    # one launch (hi / lo images of x in LDS, the weights' hi / lo stream) against the three accumulating launches: the same products
    y1, y3 = conv3x3(x, pf, bias, packed_lo=pfl), conv3x3(x, pf, bias, packed_lo=pfl, three_launches=True)
    assert (y1 - y3).norm().item() <= 1e-6 * y3.norm().item() and not torch.equal(y1, torch.zeros_like(y1))
    e3b = (y3.double() - ref).norm().item() / ref.norm().item()
    assert e3b <= 2e-5
    # the fused epilogue runs once, on the sum of the three products
    got = conv3x3(x, pf, bias, relu=True, skip=skip, mask=skip.relu(), packed_lo=pfl)
    want = torch.where(skip > 0, (ref + skip.double()).clamp_min(0), torch.zeros_like(ref))
    assert (got.double() - want).norm().item() <= 2e-5 * want.norm().item()
    gx, gw, gb = torch.ops.aten.convolution_backward(gy.double(), x.double(), w.double(), [128], [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, True, True])
    assert (conv3x3(gy, pb, packed_lo=pbl).double() - gx).norm().item() <= 2e-5 * gx.norm().item()
    dw, db = torch.empty_like(w), torch.empty(128, device="cuda")
    conv3x3_wgrad([x[:20], x[20:]], [gy[:20], gy[20:]], dw, db, x3=True)
    ew = (dw.double() - gw).norm().item() / gw.norm().item()
    print("weight gradient: relative L2 error bf16x3 %.3g" % ew)
    assert ew <= 2e-5 and (db.double() - gb).abs().max().item() <= 1e-5 * gb.abs().max().item() + 1e-4


@pytest.mark.parametrize("n", [1, 6, 64, 1030])
def test_split_precision_residual_tower_matches_fp64(n):
    """mz_restower_x3_forward: three residual blocks (network.py:35-47) in ONE LDS-resident launch, activations as hi / lo bf16
    images, three products per convolution, the skip path in fp32 registers -- against the fp64 modules and against the chain
    of single split-precision convolutions (three launches each)"""
    from muzero_amd.convgrad import conv3x3, pack_conv, restower_x3, restower_x3_pack
    g = torch.Generator(device="cuda").manual_seed(100 + n)
    ws = [torch.randn(128, 128, 3, 3, device="cuda", generator=g) * 0.03 for _ in range(6)]
    bias = torch.randn(6, 128, device="cuda", generator=g) * 0.2
    x = torch.randn(n, 128, 8, 8, device="cuda", generator=g)
    packed = restower_x3_pack(ws)
    F = torch.nn.functional

    def ref(h):
        h = h.double()
        for blk in range(3):
            t = F.conv2d(h, ws[2 * blk].double(), bias[2 * blk].double(), padding=1).relu()
            h = (h + F.conv2d(t, ws[2 * blk + 1].double(), bias[2 * blk + 1].double(), padding=1)).relu()
        return h
    want = ref(x)
    got = restower_x3(x, packed, bias)
    err = (got.double() - want).norm().item() / want.norm().item()
    print("residual tower, %d boards: relative L2 error %.3g, max %.3g of max %.3g" % (n, err, (got.double() - want).abs().max().item(), want.abs().max().item()))
    assert err <= 2e-5
    assert (got.double() - want).abs().max().item() <= 1e-4 * want.abs().max().item()
    # the chain of single convolutions computes the same products; it sums them in another order and rounds the skip path
    # through fp32 tensors in HBM, not through registers
    h = x
    for blk in range(3):
        p1, p2 = pack_conv(ws[2 * blk], split=True), pack_conv(ws[2 * blk + 1], split=True)
        t = conv3x3(h, p1[0], bias[2 * blk], relu=True, packed_lo=p1[2])
        h = conv3x3(t, p2[0], bias[2 * blk + 1], relu=True, skip=h, packed_lo=p2[2])
    assert (got - h).norm().item() <= 2e-5 * h.norm().item()
    # ReLU on load, in place, and the same bits again
    want2 = ref(x.relu())
    y = x.clone()
    assert restower_x3(y, packed, bias, relu_in=True, out=y) is y
    assert (y.double() - want2).norm().item() <= 2e-5 * want2.norm().item()
    assert torch.equal(restower_x3(x.relu(), packed, bias), y)
    # repacking in place serves the new weights
    ws2 = [w.flip(0) for w in ws]
    restower_x3_pack(ws2, out=packed)
    assert torch.equal(restower_x3(x, packed, bias), restower_x3(x, restower_x3_pack(ws2), bias))


@pytest.mark.parametrize("n,cin,with_bias", [(1, 9, True), (7, 10, False), (64, 10, False), (1030, 9, True), (33, 16, True), (5, 1, False)])
def test_split_precision_whole_tower_matches_fp64(n, cin, with_bias):
    """mz_tower_x3_forward: first convolution (cin -> 128), three residual blocks, last convolution (128 -> 8) -- Representation
    (network.py:49-77, 9 planes, biases) and Dynamics (network.py:113-135, 10 planes, no bias on the ends) -- in ONE launch,
    against the fp64 modules"""
    from muzero_amd.convgrad import restower_x3_pack, tower_x3, tower_x3_pack_ends
    g = torch.Generator(device="cuda").manual_seed(7 * n + cin)
    ws = [torch.randn(128, 128, 3, 3, device="cuda", generator=g) * 0.03 for _ in range(6)]
    bias = torch.randn(6, 128, device="cuda", generator=g) * 0.2
    wf = torch.randn(128, cin, 3, 3, device="cuda", generator=g) * 0.2
    wl = torch.randn(8, 128, 3, 3, device="cuda", generator=g) * 0.03
    bf = torch.randn(128, device="cuda", generator=g) * 0.2 if with_bias else None
    bl = torch.randn(8, device="cuda", generator=g) * 0.2 if with_bias else None
    x = torch.rand(n, cin, 8, 8, device="cuda", generator=g)
    F = torch.nn.functional
    d = lambda t: None if t is None else t.double()  # noqa: E731
    h = F.conv2d(x.double(), wf.double(), d(bf), padding=1).relu()
    for blk in range(3):
        t = F.conv2d(h, ws[2 * blk].double(), bias[2 * blk].double(), padding=1).relu()
        h = (h + F.conv2d(t, ws[2 * blk + 1].double(), bias[2 * blk + 1].double(), padding=1)).relu()
    want = F.conv2d(h, wl.double(), d(bl), padding=1).relu()
    packed, ends = restower_x3_pack(ws), tower_x3_pack_ends(wf, wl)
    got = tower_x3(x, packed, ends, bias, bf, bl)
    assert got.shape == (n, 8, 8, 8) and got.dtype == torch.float32
    err = (got.double() - want).norm().item() / want.norm().item()
    print("whole tower, %d boards, %d planes: relative L2 error %.3g, max %.3g of max %.3g" % (n, cin, err, (got.double() - want).abs().max().item(), want.abs().max().item()))
    assert err <= 2e-5
    assert (got.double() - want).abs().max().item() <= 1e-4 * want.abs().max().item()
    assert torch.equal(tower_x3(x, packed, ends, bias, bf, bl), got)


def test_training_gradients_in_split_precision_match_the_fp32_step():
    """the K = 5 unrolled loss with the towers in split precision against the fp32 PyTorch/MIOpen step (itself Winograd
    arithmetic, not exact sums): measured on MI355X worst tensor 1.0e-3, median 1.7e-4 -- 50-90 x closer than the one-product
    bf16 form (5.3e-2 / 1.5e-2)"""
    from muzero_amd import train as T
    from muzero_amd.network import Network
    torch.manual_seed(1261)
    net = Network().to("cuda")
    net.train()
    B, K = 64, 5
    g = torch.Generator(device="cpu").manual_seed(3)
    batch = (torch.rand(B, 9, 8, 8, generator=g).round().cuda(), torch.rand(K, B, 2, 8, 8, generator=g).round().cuda(),
             (torch.rand(K + 1, B, generator=g) * 2 - 1).cuda(), torch.softmax(torch.rand(K + 1, B, 65, generator=g), -1).cuda())
    pl0, vl0, ref = _grads(net, batch, hip=False)
    pl1, vl1, got = _grads(net, batch, hip="x3")
    assert abs(pl0 - pl1) <= 1e-5 * abs(pl0) and abs(vl0 - vl1) <= 1e-5 * abs(vl0)
    rel = sorted(float((got[k].double() - ref[k].double()).norm() / ref[k].double().norm().clamp_min(1e-12)) for k in ref)
    print("split precision vs fp32: per-tensor relative L2 worst %.3g median %.3g" % (rel[-1], rel[len(rel) // 2]))
    assert rel[-1] <= 5e-3 and rel[len(rel) // 2] <= 5e-4


def test_pack_convs_equals_pack_conv():
    from muzero_amd.convgrad import pack_conv, pack_convs
    g = torch.Generator(device="cuda").manual_seed(2)
    ws = [torch.randn(128, 128, 3, 3, device="cuda", generator=g) * 0.03 for _ in range(12)]
    for split in (False, True):
        many = pack_convs(ws, split=split)
        assert len(many) == 12
        for w, got in zip(ws, many):
            want = pack_conv(w, split=split)
            assert len(got) == len(want) and all(torch.equal(a, b) for a, b in zip(got, want))
