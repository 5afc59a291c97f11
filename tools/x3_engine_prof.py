"""python tools/x3_engine_prof.py [boards] [reps]: InferenceEngine(tower="hip_x3").recurrent in a loop (run under rocprofv3 --kernel-trace --stats
to see what one fp32-grade inference step is made of)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from muzero_amd.network import InferenceEngine, Network  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
torch.manual_seed(1)
eng = InferenceEngine(Network(), n, torch.float32, channels_last=False, tower="hip_x3")
x10 = torch.rand(n, 10, 8, 8, device="cuda")
for _ in range(5):
    eng.recurrent(x10)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    eng.recurrent(x10)
torch.cuda.synchronize()
print("recurrent inference, %d boards: %.1f us per call" % (n, (time.perf_counter() - t0) / reps * 1e6))
