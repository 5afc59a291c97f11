"""robustness probe of the split-precision kernels on extreme inputs"""
import sys; import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from muzero_amd.convgrad import conv3x3, conv3x3_wgrad, pack_conv
g = torch.Generator(device="cuda").manual_seed(1)
w = torch.randn(128, 128, 3, 3, device="cuda", generator=g) * 0.03
pf, pb, pfl, pbl = pack_conv(w, split=True)
for n in (512, 37):
    x = torch.randn(n, 128, 8, 8, device="cuda", generator=g)
    for k in (-38, -30, -20, -10, 0, 10, 20, 30, 36):
        xs = x * (10.0 ** k)
        y = conv3x3(xs, pf, None, packed_lo=pfl)
        y2 = conv3x3(xs, pf, None, packed_lo=pfl)
        ref = torch.nn.functional.conv2d(xs.double(), w.double(), padding=1)
        err = ((y.double() - ref).norm() / ref.norm().clamp_min(1e-300)).item()
        gs = torch.randn(n, 128, 8, 8, device="cuda", generator=g) * (10.0 ** k)
        dw = torch.empty_like(w); conv3x3_wgrad([x], [gs], dw, None, x3=True)
        print(n, "scale 1e%d" % k, "finite", bool(torch.isfinite(y).all()), "repeatable", bool(torch.equal(y, y2)), "rel err %.3g" % err, "wgrad finite", bool(torch.isfinite(dw).all()), flush=True)
# sparse inputs (post-ReLU zeros), masks and skips
x = torch.randn(512, 128, 8, 8, device="cuda", generator=g).relu()
sk = torch.randn(512, 128, 8, 8, device="cuda", generator=g)
for i in range(20):
    a = conv3x3(x, pf, None, relu=True, skip=sk, mask=x, packed_lo=pfl)
    b = conv3x3(x, pf, None, relu=True, skip=sk, mask=x, packed_lo=pfl)
    assert torch.equal(a, b) and torch.isfinite(a).all(), i
print("in-place accumulation repeatable over 20 runs")
