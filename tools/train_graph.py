"""The whole training step (zero grads, K+1 network passes, backward, optimiser) as ONE hipGraph against the eager step:
   python tools/train_graph.py [B] [K]
The eager step is bound by the host (about 900 launches per step through PyTorch's dispatcher and autograd); MIOpen picks its
kernels during the eager warm-up, and the captured graph replays exactly those."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from muzero_amd import train as T
from muzero_amd.network import Network

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = int(sys.argv[2]) if len(sys.argv) > 2 else 5
T.LOG = None
dev = torch.device("cuda")


def batch(seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.rand(B, 9, 8, 8, generator=g).round().to(dev), torch.rand(K, B, 2, 8, 8, generator=g).round().to(dev),
            (torch.rand(K + 1, B, generator=g) * 2 - 1).to(dev), torch.softmax(torch.rand(K + 1, B, 65, generator=g), -1).to(dev))


def run(name, benchmark, amp, graph, hip=False):
    torch.backends.cudnn.benchmark = benchmark
    torch.manual_seed(1261)
    net = Network().to(dev)
    net.train()
    opt = T.make_optimizer(net)
    bucket = T.FlatGradBucket(net.parameters())
    net.steps = 1
    static = batch(1)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(4):
            T.update_weights(opt, net, static, bucket, amp_dtype=amp, hip_convs=hip)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    if graph:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = T.update_weights(opt, net, static, bucket, amp_dtype=amp, hip_convs=hip)
        step = g.replay
    else:
        out = None

        def step():
            nonlocal out
            out = T.update_weights(opt, net, static, bucket, amp_dtype=amp, hip_convs=hip)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for i in range(n):
        new = batch(2 + i % 3)                      # fresh data every step, copied into the captured inputs
        for d, t in zip(static, new):
            d.copy_(t)
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for i in range(n):
        step()
    torch.cuda.synchronize()
    dt2 = (time.perf_counter() - t0) / n
    print("%-46s %7.2f ms/step (with a host-made batch copied in: %7.2f)   p_loss %.4f v_loss %.4f" %
          (name, dt2 * 1e3, dt * 1e3, float(out[0]), float(out[1])), flush=True)


for graph in (False, True):
    run("fp32 + HIP bf16 tower convolutions, %s" % ("hipGraph" if graph else "eager"), False, None, graph, hip=True)
for graph in (False, True):
    run("fp32 + HIP split-precision (bf16x3) tower convolutions, %s" % ("hipGraph" if graph else "eager"), False, None, graph, hip="x3")
if len(sys.argv) > 3 and sys.argv[3] == "hip":
    sys.exit(0)
for amp in (None, torch.bfloat16):
    for bench in (False, True):
        for graph in (False, True):
            run("%s, MIOpen find %s, %s" % ("fp32" if amp is None else "bf16 autocast", "on" if bench else "off", "hipGraph" if graph else "eager"), bench, amp, graph)
