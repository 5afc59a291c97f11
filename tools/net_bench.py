"""micro-benchmark of the PyTorch-ROCm inference step (recurrent_inference) under different
memory formats / dtypes / MIOpen modes.  python tools/net_bench.py [batch]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from muzero_amd.network import Network, InferenceEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096

def timeit(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / it

for bench in (False, True):
    torch.backends.cudnn.benchmark = bench
    for dt in (torch.bfloat16, torch.float16, torch.float32):
        for cl in (True, False):
            torch.manual_seed(0)
            eng = InferenceEngine(Network(), B, dt, channels_last=cl)
            x = torch.rand(B, 10, 8, 8, device="cuda").to(dt)
            try:
                ms = timeit(lambda: eng.recurrent(x))
            except Exception as ex:
                print("fail", bench, dt, cl, ex); continue
            print("benchmark=%s dtype=%s channels_last=%s: %.3f ms  %.1f TFLOP/s" % (bench, dt, cl, ms, B * 116e6 / ms / 1e9), flush=True)
