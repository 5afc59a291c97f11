import sys, time, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from muzero_amd.network import Network, InferenceEngine
eng = InferenceEngine(Network(), 64, torch.bfloat16)
sd = Network().state_dict()
for _ in range(3): eng.load_weights(sd)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): eng.load_weights(sd)
torch.cuda.synchronize(); print("load_weights: %.2f ms" % ((time.perf_counter() - t0) / 20 * 1e3))
