"""Soak: the graphed training step (train.GraphedUpdate) against the eager step on the same stream of random batches, two
networks from one initialisation: python tools/graph_soak.py [steps] [bf16|x3|fp32] [lr]
Prints the running difference of the weights; a corruption in the replayed graph shows as a jump."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from muzero_amd import config, train as T
from muzero_amd.network import Network
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from graphed_update import GraphedUpdate

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
mode = sys.argv[2] if len(sys.argv) > 2 else "bf16"
lr = float(sys.argv[3]) if len(sys.argv) > 3 else 0.005
if len(sys.argv) > 4:
    torch.backends.cuda.preferred_blas_library(sys.argv[4])
    print("preferred BLAS library:", torch.backends.cuda.preferred_blas_library())
hip = {"bf16": True, "x3": "x3", "fp32": False}[mode]
T.LOG = None
B, K = 512, 5
config.num_unroll_steps = K
dev = torch.device("cuda")
gen = torch.Generator(device="cuda").manual_seed(5)


def batch():
    return ((torch.rand(B, 9, 8, 8, device=dev, generator=gen) < 0.3).float(), (torch.rand(K, B, 2, 8, 8, device=dev, generator=gen) < 0.05).float(),
            torch.rand(K + 1, B, device=dev, generator=gen) * 2 - 1, torch.softmax(torch.randn(K + 1, B, 65, device=dev, generator=gen) * 2, -1))


nets, opts, buckets = [], [], []
for _ in range(2):
    torch.manual_seed(1261)
    n = Network().to(dev)
    n.train()
    nets.append(n)
    opts.append(torch.optim.SGD(n.parameters(), lr=lr, momentum=0.9, weight_decay=1e-4))
    buckets.append(T.FlatGradBucket(n.parameters()))
gu = GraphedUpdate(opts[1], nets[1], buckets[1], batch(), hip_convs=hip)
worst = 0.0
for s in range(steps):
    b = batch()
    le = T.update_weights(opts[0], nets[0], b, buckets[0], hip_convs=hip)
    lg = gu(b)
    if s < 12 or s % 50 == 0 or s == steps - 1:
        ds = {k: float((p.detach() - q.detach()).norm() / q.detach().norm().clamp_min(1e-12)) for (k, p), q in zip(nets[0].named_parameters(), nets[1].parameters())}
        d = max(ds.values())
        print("   worst:", ", ".join("%s %.2g" % kv for kv in sorted(ds.items(), key=lambda kv: -kv[1])[:3]))
        gd = float((buckets[0].flat - buckets[1].flat).norm() / buckets[1].flat.norm().clamp_min(1e-12))
        worst = max(worst, d)
        print("step %5d: eager loss %.4f %.4f | graphed %.4f %.4f | worst tensor rel diff %.3g | gradient rel diff %.3g" %
              (s, float(le[0]), float(le[1]), float(lg[0]), float(lg[1]), d, gd), flush=True)
        if not (d == d) or d > 0.5:
            print("DIVERGED")
            break
