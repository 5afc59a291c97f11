"""graph-only replays vs eager-only steps on the same batch sequence, run one after the other (no interleaving):
python tools/graph_soak2.py [steps] [bf16|x3|fp32]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from muzero_amd import config, train as T
from muzero_amd.network import Network
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from graphed_update import GraphedUpdate

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
mode = sys.argv[2] if len(sys.argv) > 2 else "bf16"
hip = {"bf16": True, "x3": "x3", "fp32": False}[mode]
T.LOG = None
B, K = 512, 5
config.num_unroll_steps = K
dev = torch.device("cuda")


def batches(n):
    gen = torch.Generator(device="cuda").manual_seed(5)
    for _ in range(n):
        yield ((torch.rand(B, 9, 8, 8, device=dev, generator=gen) < 0.3).float(), (torch.rand(K, B, 2, 8, 8, device=dev, generator=gen) < 0.05).float(),
               torch.rand(K + 1, B, device=dev, generator=gen) * 2 - 1, torch.softmax(torch.randn(K + 1, B, 65, device=dev, generator=gen) * 2, -1))


def run(graphed, check_every=10):
    torch.manual_seed(1261)
    n = Network().to(dev)
    n.train()
    opt = torch.optim.SGD(n.parameters(), lr=0.005, momentum=0.9, weight_decay=1e-4)
    bucket = T.FlatGradBucket(n.parameters())
    it = batches(steps + 1)
    step = GraphedUpdate(opt, n, bucket, next(it), hip_convs=hip) if graphed else (lambda b, _first=next(it): T.update_weights(opt, n, b, bucket, hip_convs=hip))
    snaps = []
    for s, b in enumerate(it):
        step(b)
        if s % check_every == 0 or s == steps - 1:
            snaps.append((s, [p.detach().clone() for p in n.parameters()], bucket.flat.clone()))
    return snaps, [k for k, _ in n.named_parameters()]


for a_name, a_g, b_name, b_g in (("eager", False, "eager again", False), ("graph", True, "graph again", True), ("eager", False, "graph", True)):
    A, names = run(a_g)
    Bn, _ = run(b_g)
    out = []
    for (s, wa, ga), (_, wb, gb) in zip(A, Bn):
        d = {k: float((p - q).norm() / q.norm().clamp_min(1e-12)) for k, p, q in zip(names, wa, wb)}
        k = max(d, key=d.get)
        out.append("step %d: weights %.2g (%s) grads %.2g" % (s, d[k], k.replace("prediction.", "pred."), float((ga - gb).norm() / gb.norm().clamp_min(1e-12))))
    print("%s vs %s:\n   " % (a_name, b_name) + "\n   ".join(out[:4] + out[-3:]), flush=True)
