"""python tools/restower_stamps.py [boards]: cycles per phase of one 4-board tile of k_restower_x3 (mz_restower_x3_stamps)."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from muzero_amd import _lib as L  # noqa: E402
from muzero_amd.convgrad import restower_x3_pack  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
g = torch.Generator(device="cuda").manual_seed(1)
ws = [torch.randn(128, 128, 3, 3, device="cuda", generator=g) * 0.03 for _ in range(6)]
bias = torch.randn(6, 128, device="cuda", generator=g) * 0.2
x = torch.randn(n, 128, 8, 8, device="cuda", generator=g)
y = torch.empty_like(x)
packed = restower_x3_pack(ws)
tiles = (n + 3) // 4
st = torch.zeros((tiles, 32), dtype=torch.int64, device="cuda")
for _ in range(20):   # sustained load: the clock the kernel holds
    L.call("mz_restower_x3_stamps", L.dptr(packed), L.dptr(bias), L.dptr(x), L.dptr(y), n, L.dptr(st), L.current_stream())
torch.cuda.synchronize()
s = st.cpu().double()
names = ["load + images"]
for l in range(6):
    names += ["layer %d  W_hi (x_hi | x_lo)" % l, "layer %d  W_lo x_hi" % l, "layer %d  epilogue" % l]
names += ["store"]
idx = [0, 1] + [k for l in range(6) for k in (2 + 3 * l, 3 + 3 * l, 4 + 3 * l)] + [20]
total = (s[:, 20] - s[:, 0]).mean().item()
for k, name in enumerate(names):
    d = s[:, idx[k + 1]] - s[:, idx[k]]
    print("%-28s mean %9.0f  min %9.0f  max %9.0f   %5.1f %%" % (name, d.mean().item(), d.min().item(), d.max().item(), 100 * d.mean().item() / total))
rt = (s[:, 31] - s[:, 30])   # 100 MHz ticks
print("total per tile: %.0f cycles; in-kernel clock %.3f GHz; tile %.1f us; tiles per CU %.2f" % (
    total, ((s[:, 20] - s[:, 0]) / rt).median().item() * 0.1, rt.mean().item() * 0.01, tiles / 256))
print("ideal MFMA cycles per layer: 18432 x 3 = 55296")
