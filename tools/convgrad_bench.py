"""the hand-written training convolution kernels against PyTorch/MIOpen on the same shapes: python tools/convgrad_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from muzero_amd import convgrad as CG


def timeit(fn, it=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / it * 1e3


torch.manual_seed(0)
w = torch.randn(128, 128, 3, 3, device="cuda") * 0.03
for uses, n in ((1, 512), (5, 512)):
    xs = [torch.randn(n, 128, 8, 8, device="cuda").relu() for _ in range(uses)]
    gs = [torch.randn(n, 128, 8, 8, device="cuda") for _ in range(uses)]
    dw, db = torch.empty_like(w), torch.empty(128, device="cuda")
    flop = 2.0 * uses * n * 64 * 128 * 128 * 9
    t_hip = timeit(lambda: CG.conv3x3_wgrad(xs, gs, dw, db))

    def miopen():
        X, G = (torch.cat(xs), torch.cat(gs)) if uses > 1 else (xs[0], gs[0])
        return torch.ops.aten.convolution_backward(G, X, w, [128], [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, True])
    t_mi = timeit(miopen)
    print("wgrad %d x %d boards, split precision: hip %.1f us" % (uses, n, timeit(lambda: CG.conv3x3_wgrad(xs, gs, dw, db, x3=True))), flush=True)
    print("wgrad %d x %d boards: hip %.1f us (%.0f TFLOP/s, %.2f TB/s of fp32 operands)   miopen (cat + wrw + bias) %.1f us" %
          (uses, n, t_hip, flop / t_hip / 1e6, uses * n * 65536 / t_hip / 1e6, t_mi), flush=True)

for n in (512, 2560):
    x = torch.randn(n, 128, 8, 8, device="cuda").relu()
    y = torch.empty_like(x)
    b = torch.randn(128, device="cuda")
    pf, pb = CG.pack_conv(w)
    flop = 2.0 * n * 64 * 128 * 128 * 9
    t_hip = timeit(lambda: CG.conv3x3(x, pf, b, out=y))
    t_mi = timeit(lambda: torch.nn.functional.conv2d(x, w, b, padding=1))
    pf4 = CG.pack_conv(w, split=True)
    print("conv %d boards, split precision (3 launches): hip %.1f us" % (n, timeit(lambda: CG.conv3x3(x, pf4[0], b, out=y, packed_lo=pf4[2]))), flush=True)
    print("conv %d boards: hip %.1f us (%.0f TFLOP/s)   miopen fp32 %.1f us" % (n, t_hip, flop / t_hip / 1e6, t_mi), flush=True)
print("pack: %.1f us" % timeit(lambda: CG.pack_conv(w)))
