"""per-parameter gradient difference to the fp32 step: HIP bf16 convolutions vs torch.autocast(bf16).  python tools/train_grad_noise.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from muzero_amd import train as T
from muzero_amd.network import Network

B, K = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 5
torch.manual_seed(1261)
net = Network().cuda()
net.train()
g = torch.Generator(device="cpu").manual_seed(3)
batch = (torch.rand(B, 9, 8, 8, generator=g).round().cuda(), torch.rand(K, B, 2, 8, 8, generator=g).round().cuda(),
         (torch.rand(K + 1, B, generator=g) * 2 - 1).cuda(), torch.softmax(torch.rand(K + 1, B, 65, generator=g), -1).cuda())


def grads(mode):
    for p in net.parameters():
        p.grad = None
    if mode == "autocast":
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pl, vl = T.muzero_loss(net, batch)
    else:
        dw = T.DeferredWeightGrads(hip=(mode == "hip"))
        pl, vl = T.muzero_loss_batched(net, batch, dw)
    (pl.float() + vl.float()).backward()
    if mode != "autocast":
        dw.flush()
    return float(pl.detach()), float(vl.detach()), {k: p.grad.detach().double().clone() for k, p in net.named_parameters()}


ref = grads("fp32")
for mode in ("fp32", "hip", "autocast"):
    pl, vl, got = grads(mode)
    d = {k: float((got[k] - ref[2][k]).norm() / ref[2][k].norm().clamp_min(1e-30)) for k in got}
    worst = sorted(d.items(), key=lambda kv: -kv[1])[:4]
    print("%-9s p_loss %.5f v_loss %.5f  worst relative L2: %s   median %.3g" % (mode, pl, vl, ", ".join("%s %.3g" % kv for kv in worst), sorted(d.values())[len(d) // 2]), flush=True)
