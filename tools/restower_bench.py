"""python tools/restower_bench.py [boards]: the split-precision residual tower as one LDS-resident launch (mz_restower_x3_forward)
against the chain of single split-precision convolutions (18 launches) and PyTorch/MIOpen fp32, on the same box."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))

from muzero_amd.convgrad import conv3x3, pack_conv, restower_x3, restower_x3_pack

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
g = torch.Generator(device="cuda").manual_seed(1)
ws = [torch.randn(128, 128, 3, 3, device="cuda", generator=g) * 0.03 for _ in range(6)]
bias = torch.randn(6, 128, device="cuda", generator=g) * 0.2
x = torch.randn(n, 128, 8, 8, device="cuda", generator=g)
packed = restower_x3_pack(ws)
packs = [pack_conv(w, split=True) for w in ws]
F = torch.nn.functional


def fused():
    return restower_x3(x, packed, bias)


def chain():
    h = x
    for blk in range(3):
        p1, p2 = packs[2 * blk], packs[2 * blk + 1]
        t = conv3x3(h, p1[0], bias[2 * blk], relu=True, packed_lo=p1[2])
        h = conv3x3(t, p2[0], bias[2 * blk + 1], relu=True, skip=h, packed_lo=p2[2])
    return h


def miopen():
    h = x
    for blk in range(3):
        t = F.conv2d(h, ws[2 * blk], bias[2 * blk], padding=1).relu_()
        h = (h + F.conv2d(t, ws[2 * blk + 1], bias[2 * blk + 1], padding=1)).relu_()
    return h


def timed(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


flops = 2 * 6 * 128 * 128 * 9 * 64 * n
ref = miopen().double()
for name, f in (("fused x3 (1 launch)", fused), ("chain x3 (18 launches)", chain), ("PyTorch/MIOpen fp32", miopen)):
    us = timed(f)
    err = (f().double() - ref).norm().item() / ref.norm().item()
    print("%-24s %9.1f us per %d boards  = %7.1f TFLOP/s of fp32-grade work (x3 = %7.1f TFLOP/s of bf16 MFMA work)   vs MIOpen fp32: rel L2 %.2g"
          % (name, us, n, flops / us * 1e-6, 3 * flops / us * 1e-6, err))
