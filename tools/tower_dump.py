"""sha256 of the inference launch's outputs on a seeded input: python tools/tower_dump.py [n]   (A/B of library builds selected
with MUZERO_AMD_LIB: equal digests = bit-identical kernels)"""
import hashlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from muzero_amd.network import InferenceEngine, Network

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
torch.manual_seed(1261)
net = Network()
eng = InferenceEngine(net, n, torch.bfloat16)
g = torch.Generator(device="cuda").manual_seed(3)
x10 = torch.randn((n, 10, 8, 8), device="cuda", generator=g).to(torch.bfloat16)
img = (torch.rand((n, 9, 8, 8), device="cuda", generator=g) < 0.3).to(torch.bfloat16)
h = hashlib.sha256()
for out in (eng.recurrent(x10), eng.initial(img)):
    for t in out:
        h.update(t.contiguous().view(torch.uint8).cpu().numpy().tobytes())
print("digest", n, h.hexdigest())
