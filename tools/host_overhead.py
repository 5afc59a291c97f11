"""host-side cost of the convolution wrappers against the raw ctypes call (why train._TowerHip uses conv3x3_unchecked): python tools/host_overhead.py"""
import sys, time, torch
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from muzero_amd.convgrad import conv3x3, pack_conv, conv3x3_wgrad
from muzero_amd import _lib as L
w = torch.randn(128, 128, 3, 3, device="cuda") * 0.03
pf, pb = pack_conv(w)
x = torch.randn(4, 128, 8, 8, device="cuda"); y = torch.empty_like(x); b = torch.zeros(128, device="cuda")
def t(f, n=3000):
    for _ in range(100): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    dt = time.perf_counter() - t0; torch.cuda.synchronize()
    return dt / n * 1e6
print("conv3x3 wrapper        %.1f us per call (host issue)" % t(lambda: conv3x3(x, pf, b, relu=True, out=y, skip=x)))
st = L.current_stream(); px, py, pp, pb_ = L.dptr(x), L.dptr(y), L.dptr(pf), L.dptr(b)
print("raw ctypes call        %.1f us" % t(lambda: L.lib.mz_conv3x3_fused(pp, pb_, px, 0, px, None, py, 4, 1, st)))
print("L.current_stream()     %.1f us" % t(lambda: L.current_stream()))
print("L.dptr(x)              %.1f us" % t(lambda: L.dptr(x)))
print("torch.empty_like(x)    %.1f us" % t(lambda: torch.empty_like(x)))
print("torch relu_ (eager op) %.1f us" % t(lambda: y.relu_()))
dw = torch.zeros_like(w); db = torch.zeros(128, device="cuda")
print("conv3x3_wgrad wrapper  %.1f us" % t(lambda: conv3x3_wgrad([x], [y], dw, db, accumulate=True), 500))
