"""time the HIP tower kernel against the MIOpen path: python tools/tower_bench.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from muzero_amd.network import Network, InferenceEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096


def timeit(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / it


torch.manual_seed(0)
net = Network()
x = torch.rand(B, 10, 8, 8, device="cuda").to(torch.bfloat16)
for tower in (sys.argv[2].split(",") if len(sys.argv) > 2 else ("hip", "miopen")):
    eng = InferenceEngine(net, B, torch.bfloat16, tower=tower)
    ms_t = timeit(lambda: eng._hidden_recurrent(x))
    ms = timeit(lambda: eng.recurrent(x))
    print("%-7s tower %.3f ms (%.0f TFLOP/s)   tower+heads %.3f ms" % (tower, ms_t, B * 115.6e6 / ms_t / 1e9, ms), flush=True)
