"""Training-step variants on the GPU: memory format / MIOpen find mode / autocast.  python tools/train_bench.py [B] [K]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from muzero_amd import config, train as T
from muzero_amd.network import Network

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = int(sys.argv[2]) if len(sys.argv) > 2 else 5
T.LOG = None
dev = torch.device("cuda")


def batch():
    g = torch.Generator(device="cpu").manual_seed(1)
    image = torch.rand(B, 9, 8, 8, generator=g).round().to(dev)
    actions = torch.rand(K, B, 2, 8, 8, generator=g).round().to(dev)
    tv = (torch.rand(K + 1, B, generator=g) * 2 - 1).to(dev)
    tp = torch.softmax(torch.rand(K + 1, B, 65, generator=g), -1).to(dev)
    return image, actions, tv, tp


def run(name, channels_last=False, benchmark=False, amp=None, batched=None, hip=False):
    torch.backends.cudnn.benchmark = benchmark
    torch.manual_seed(1261)
    net = Network().to(dev)
    if channels_last:
        net = net.to(memory_format=torch.channels_last)
    net.train()
    opt = T.make_optimizer(net)
    bucket = T.FlatGradBucket(net.parameters())
    net.steps = 1
    b = batch()
    for _ in range(4):
        T.update_weights(opt, net, b, bucket, amp_dtype=amp, batched=batched, hip_convs=hip)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        pl, vl = T.update_weights(opt, net, b, bucket, amp_dtype=amp, batched=batched, hip_convs=hip)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("%-44s %7.2f ms/step   p_loss %.4f v_loss %.4f" % (name, dt * 1e3, float(pl), float(vl)), flush=True)


run("fp32 NCHW, plain loss (r02)", batched=False)
run("fp32 NCHW, batched heads + deferred weight gradients (as shipped)")
run("fp32 NCHW, batched, HIP bf16 convolution gradients", hip=True)
run("fp32 NCHW, plain loss (r02), again", batched=False)
run("fp32 NCHW, batched, again")
run("fp32 NCHW, batched, HIP bf16 convolution gradients, again", hip=True)
run("fp32 NCHW, batched, HIP split-precision (bf16x3) convolutions", hip="x3")
if len(sys.argv) > 3 and sys.argv[3] == "short":
    sys.exit(0)
run("fp32 NCHW, MIOpen find (benchmark=True)", benchmark=True)
run("fp32 channels_last", channels_last=True)
run("fp32 channels_last, find", channels_last=True, benchmark=True)
run("bf16 autocast NCHW", amp=torch.bfloat16)
run("bf16 autocast channels_last", channels_last=True, amp=torch.bfloat16)
run("bf16 autocast channels_last, find", channels_last=True, benchmark=True, amp=torch.bfloat16)
