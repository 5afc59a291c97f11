"""which side is wrong when eager steps and graph replays interleave?  After every step the gradient each side used is recomputed
eagerly on a clone of its pre-step weights: python tools/graph_soak3.py [steps] [fp32|bf16]"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from muzero_amd import config, train as T
from muzero_amd.network import Network
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from graphed_update import GraphedUpdate

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
hip = {"bf16": True, "fp32": False}[sys.argv[2] if len(sys.argv) > 2 else "fp32"]
interleave = not (len(sys.argv) > 3 and sys.argv[3] == "solo")
T.LOG = None
B, K = 512, 5
config.num_unroll_steps = K
dev = torch.device("cuda")
gen = torch.Generator(device="cuda").manual_seed(5)


def batch():
    return ((torch.rand(B, 9, 8, 8, device=dev, generator=gen) < 0.3).float(), (torch.rand(K, B, 2, 8, 8, device=dev, generator=gen) < 0.05).float(),
            torch.rand(K + 1, B, device=dev, generator=gen) * 2 - 1, torch.softmax(torch.randn(K + 1, B, 65, device=dev, generator=gen) * 2, -1))


def fresh_grads(net, b):
    c = copy.deepcopy(net)
    for p in c.parameters():
        p.grad = None
    dw = T.DeferredWeightGrads(hip=hip)
    pl, vl = T.muzero_loss_batched(c, b, dw)
    (pl + vl).backward()
    dw.flush()
    return torch.cat([p.grad.reshape(-1) for p in c.parameters()]), {k: p.grad for k, p in c.named_parameters()}


nets, opts, buckets = [], [], []
for _ in range(2):
    torch.manual_seed(1261)
    n = Network().to(dev)
    n.train()
    nets.append(n)
    opts.append(torch.optim.SGD(n.parameters(), lr=0.005, momentum=0.9, weight_decay=1e-4))
    buckets.append(T.FlatGradBucket(n.parameters()))
gu = GraphedUpdate(opts[1], nets[1], buckets[1], batch(), hip_convs=hip)
manual = [[torch.zeros_like(p) for p in n.parameters()] for n in nets]   # momentum buffers of a hand-made SGD, per side
for s in range(steps):
    b = batch()
    pre = [copy.deepcopy(n) for n in nets]
    if interleave:
        T.update_weights(opts[0], nets[0], b, buckets[0], hip_convs=hip)
    gu(b)
    torch.cuda.synchronize()
    msg = []
    for name, i in (("eager", 0), ("graph", 1)):
        if i == 0 and not interleave:
            continue
        # the step added weight_decay * w to the gradient in place? (torch SGD does not: foreach add returns new tensors)
        ref, named = fresh_grads(pre[i], b)
        got = buckets[i].flat
        worst = max(((float((named[k] - p.grad).norm() / named[k].norm().clamp_min(1e-20)), k) for k, p in nets[i].named_parameters()))
        # the optimiser: SGD(lr 0.005, momentum 0.9, weight_decay 1e-4) by hand from the recomputed gradient and the pre-step weights
        dev_w = []
        for (k, p_new), p_old, buf in zip(nets[i].named_parameters(), pre[i].parameters(), manual[i]):
            g = named[k] + 1e-4 * p_old.detach()
            buf.mul_(0.9).add_(g)
            want = p_old.detach() - 0.005 * buf
            dev_w.append((float((p_new.detach() - want).norm() / (0.005 * buf).norm().clamp_min(1e-20)), k))
        msg.append("%s: grad %.2g, update vs hand-made SGD %.2g (%s)" % (name, float((got - ref).norm() / ref.norm()), max(dev_w)[0], max(dev_w)[1]))
        if i == 1 and s in (1, 2):
            errs = sorted(((float((named[k] - p.grad).norm() / named[k].norm().clamp_min(1e-20)), k) for k, p in nets[i].named_parameters()), reverse=True)
            print("     graph-side gradient errors by tensor:", ", ".join("%s %.2g" % (k, e) for e, k in errs[:8]))
    print("step %2d  " % s + " | ".join(msg), flush=True)
