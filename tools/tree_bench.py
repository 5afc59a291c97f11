"""k_backup_select alone, as bench.py's roofline_tree leg measures it: one real search records the network outputs of every
simulation, then the same search is replayed with the tree kernel launched back to back (HIP events on the launch stream).
   python tools/tree_bench.py [trees] [sims]      (A/B of library builds: MUZERO_AMD_LIB=muzero_amd/lib/exp/<name>.so)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from muzero_amd import mcts as MC
from muzero_amd.env import make_image
from muzero_amd.network import InferenceEngine, Network
from muzero_amd.selfplay import SelfPlayEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
S = int(sys.argv[2]) if len(sys.argv) > 2 else 50
device = torch.device("cuda", 0)
torch.manual_seed(1261)
net = InferenceEngine(Network(), n, torch.bfloat16)
POOL_S = int(os.environ.get("MZ_TREE_BENCH_POOL_SIMS", S))   # size the pool for more simulations than are run (page-spread experiment)
eng = SelfPlayEngine(n, net, num_simulations=POOL_S, device=device, use_graph=True, seed=1261)
roots, hw, hb, idx = B.make_roots(n, 1261, 0, device)
eng.env = roots
image = make_image(hw, hb, idx, n, torch.bfloat16)
noise = eng.dirichlet_noise(roots.legal)
v0, p0, h0 = net.initial(image)
h0 = h0.reshape(n, 8, 8, 8).contiguous()
eng.mcts.root_init(roots.to_play, roots.legal, p0, h0, noise)
eng.mcts.select()
rec, depth = [], 0.0
for s in range(S - 1):
    v, p, h = net.recurrent(eng.mcts.net_in)
    rec.append((v.clone(), p.clone(), h.reshape(n, 8, 8, 8).clone()))
    depth += float(eng.mcts.meta()[:, MC.META_PATH_LEN].float().mean())
    eng.mcts.backup_select(*rec[-1])
vis = eng.mcts.root_stats()[0].clone()
start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tot, launches = 0.0, 0
for rep in range(8):
    eng.mcts.root_init(roots.to_play, roots.legal, p0, h0, noise)
    eng.mcts.select()
    torch.cuda.synchronize()
    start.record()
    for s in range(S - 1):
        eng.mcts.backup_select(*rec[s])
    end.record()
    end.synchronize()
    if rep >= 2:
        tot += start.elapsed_time(end)
        launches += S - 1
assert os.environ.get("MZ_TREE_BENCH_NOCHECK") or torch.equal(eng.mcts.root_stats()[0], vis)
# the whole search as the bench times it (graph replay: tower + tree per simulation)
if POOL_S != S:
    print("k_backup_select %.2f us per launch (%d trees, %d simulations in a pool sized for %d, mean depth %.2f)" % (tot / launches * 1e3, n, S, POOL_S, depth / (S - 1)))
    sys.exit(0)
for _ in range(3):
    eng.search(image)
torch.cuda.synchronize()
start.record()
for _ in range(10):
    eng.search(image)
end.record()
end.synchronize()
print("k_backup_select %.2f us per launch (%d trees, mean depth %.2f) | search %.3f ms | visits digest %d" %
      (tot / launches * 1e3, n, depth / (S - 1), start.elapsed_time(end) / 10, int(vis.long().sum() * 7 + (vis.long() * torch.arange(65, device=device)).sum())))
