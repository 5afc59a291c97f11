"""what makes a replayed training-step graph go wrong?  python tools/graph_dbg.py <variant>"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from muzero_amd import config, train as T
from muzero_amd.network import Network
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from graphed_update import GraphedUpdate
variant = sys.argv[1]
T.LOG = None
B, K = 512, 5
config.num_unroll_steps = K
dev = torch.device("cuda")
gen = torch.Generator(device="cuda").manual_seed(5)
def batch():
    return ((torch.rand(B, 9, 8, 8, device=dev, generator=gen) < 0.3).float(), (torch.rand(K, B, 2, 8, 8, device=dev, generator=gen) < 0.05).float(),
            torch.rand(K + 1, B, device=dev, generator=gen) * 2 - 1, torch.softmax(torch.randn(K + 1, B, 65, device=dev, generator=gen) * 2, -1))
def fresh(net, b):
    c = copy.deepcopy(net)
    for p in c.parameters(): p.grad = None
    pl, vl = T.muzero_loss(c, b)
    (pl + vl).backward()
    return torch.cat([p.grad.reshape(-1) for p in c.parameters()])
torch.manual_seed(1261)
n = Network().to(dev); n.train()
opt = torch.optim.SGD(n.parameters(), lr=0.005, momentum=0.9, weight_decay=1e-4)
bk = T.FlatGradBucket(n.parameters())
if variant == "junk_before":
    junk = [torch.zeros(p.shape, device=dev) for p in n.parameters() for _ in range(2)]
gu = GraphedUpdate(opt, n, bk, batch(), hip_convs=False)
if variant in ("junk_after", "junk_ops"):
    junk = [torch.zeros(p.shape, device=dev) for p in n.parameters() for _ in range(2)]
res = []
for s in range(int(sys.argv[2]) if len(sys.argv) > 2 else 6):
    b = batch()
    pre = copy.deepcopy(n)
    if variant == "junk_ops":
        for j in junk: j.mul_(0.9).add_(1.0)
    if variant == "small_allocs":
        tmp = [torch.empty(128, device=dev) for _ in range(50)]
    gu(b); torch.cuda.synchronize()
    ref = fresh(pre, b)
    res.append(float((bk.flat - ref).norm() / ref.norm()))
bad = [(i, round(r, 3)) for i, r in enumerate(res) if r > 1e-3]
print("%-12s sync=%s: %d steps, wrong replays: %s" % (variant, os.environ.get("MZ_GRAPH_SYNC", "0"), len(res), bad), flush=True)
