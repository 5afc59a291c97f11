"""Is update_weights bit-reproducible run to run on this GPU?  python tools/train_determinism.py  (plain and batched loss, B = 512 and 32)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from muzero_amd import train as T
from muzero_amd.network import Network
T.LOG = None
dev = torch.device("cuda")


def batch(B, K):
    g = torch.Generator(device="cpu").manual_seed(1)
    return (torch.rand(B, 9, 8, 8, generator=g).round().to(dev), torch.rand(K, B, 2, 8, 8, generator=g).round().to(dev),
            (torch.rand(K + 1, B, generator=g) * 2 - 1).to(dev), torch.softmax(torch.rand(K + 1, B, 65, generator=g), -1).to(dev))


def run(B, K, batched, steps=3):
    torch.manual_seed(1261)
    net = Network().to(dev)
    net.train()
    opt = T.make_optimizer(net)
    bucket = T.FlatGradBucket(net.parameters())
    b = batch(B, K)
    for _ in range(steps):
        T.update_weights(opt, net, b, bucket, batched=batched)
    torch.cuda.synchronize()
    return torch.cat([p.detach().reshape(-1) for p in net.parameters()]).clone()


for B, K in ((512, 5), (32, 3), (16, 3)):
    for batched in (False, True):
        ws = [run(B, K, batched) for _ in range(4)]
        diffs = [int((ws[0] != w).sum()) for w in ws[1:]]
        rel = max(float((ws[0] - w).abs().max()) for w in ws[1:])
        print("B %d K %d %-8s differing weights vs first run: %s   max |diff| %.3g" % (B, K, "batched" if batched else "plain", diffs, rel), flush=True)
