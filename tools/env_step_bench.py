"""k_env_step alone: us per launch over n games resident in HBM (60 real steps of random play).  python tools/env_step_bench.py [n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from muzero_amd.env import BatchedEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
e = BatchedEnv(n, "cuda")
acts = torch.empty((60, n), dtype=torch.uint8, device="cuda")
for t in range(60):
    e.random_action(1261, 0, t, out=acts[t])
    e.step(acts[t])
torch.cuda.synchronize()
start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tot, reps = 0.0, 5
for _ in range(reps):
    e.reset()
    start.record()
    for t in range(60):
        e.step(acts[t])
    end.record()
    end.synchronize()
    tot += start.elapsed_time(end)
us = tot / (reps * 60) * 1e3
print("k_env_step: %d games, %.2f us per launch, %.2f TB/s algorithmic (46 B/step) = %.1f %% of 8 TB/s" % (n, us, n * 46 / us / 1e6, n * 46 / us / 1e6 / 8 * 100))
