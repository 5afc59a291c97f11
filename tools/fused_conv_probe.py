"""does PyTorch-ROCm expose a fused conv+bias(+add)+relu that MIOpen runs as one kernel for our shapes?"""
import torch, time
import torch.nn.functional as F
B = 4096
def timeit(fn, it=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / it * 1e3
for dt in (torch.bfloat16, torch.float16):
    x = torch.randn(B, 128, 8, 8, device="cuda", dtype=dt).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(128, 128, 3, 3, device="cuda", dtype=dt) * 0.03).contiguous(memory_format=torch.channels_last)
    b = torch.randn(128, device="cuda", dtype=dt)
    z = torch.randn_like(x)
    print(dt)
    print("  conv only            %.1f us" % timeit(lambda: F.conv2d(x, w, None, 1, 1)))
    print("  conv+bias            %.1f us" % timeit(lambda: F.conv2d(x, w, b, 1, 1)))
    print("  conv+bias+relu       %.1f us" % timeit(lambda: F.relu(F.conv2d(x, w, b, 1, 1))))
    print("  relu(z+conv+bias)    %.1f us" % timeit(lambda: F.relu(z + F.conv2d(x, w, b, 1, 1))))
    try:
        f = torch.ops.aten.miopen_convolution_relu
        ref = F.relu(F.conv2d(x, w, b, 1, 1)); got = f(x, w, b, [1, 1], [1, 1], [1, 1], 1)
        print("  miopen_convolution_relu      %.1f us  maxdiff %.4f" % (timeit(lambda: f(x, w, b, [1, 1], [1, 1], [1, 1], 1)), (ref.float() - got.float()).abs().max().item()))
    except Exception as ex:
        print("  miopen_convolution_relu failed:", str(ex)[:200])
    try:
        f2 = torch.ops.aten.miopen_convolution_add_relu
        ref = F.relu(z + F.conv2d(x, w, b, 1, 1)); got = f2(x, w, z, 1.0, b, [1, 1], [1, 1], [1, 1], 1)
        print("  miopen_convolution_add_relu  %.1f us  maxdiff %.4f" % (timeit(lambda: f2(x, w, z, 1.0, b, [1, 1], [1, 1], [1, 1], 1)), (ref.float() - got.float()).abs().max().item()))
    except Exception as ex:
        print("  miopen_convolution_add_relu failed:", str(ex)[:200])
    # plain GEMM of the same FLOPs for reference: (B*64, 1152) x (1152, 128)
    a = torch.randn(B * 64, 1152, device="cuda", dtype=dt); wm = torch.randn(1152, 128, device="cuda", dtype=dt)
    t = timeit(lambda: a @ wm)
    print("  GEMM (B*64,1152)x(1152,128)  %.1f us = %.0f TFLOP/s" % (t, 2 * B * 64 * 1152 * 128 / t / 1e6))
    a2 = torch.randn(B * 64, 128, device="cuda", dtype=dt); wm2 = torch.randn(128, 128, device="cuda", dtype=dt)
    t = timeit(lambda: a2 @ wm2)
    print("  GEMM (B*64,128)x(128,128)    %.1f us" % t)
