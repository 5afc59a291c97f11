"""one training configuration under rocprofv3: python3 tools/train_prof.py [fp32|bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from muzero_amd import train as T
from muzero_amd.network import Network
mode = sys.argv[1] if len(sys.argv) > 1 else "fp32"
T.LOG = None
dev = torch.device("cuda")
B, K = 512, 5
g = torch.Generator(device="cpu").manual_seed(1)
b = (torch.rand(B, 9, 8, 8, generator=g).round().to(dev), torch.rand(K, B, 2, 8, 8, generator=g).round().to(dev),
     (torch.rand(K + 1, B, generator=g) * 2 - 1).to(dev), torch.softmax(torch.rand(K + 1, B, 65, generator=g), -1).to(dev))
torch.manual_seed(1261)
net = Network().to(dev)
amp = None
if mode == "bf16":
    net = net.to(memory_format=torch.channels_last)
    amp = torch.bfloat16
net.train()
opt = T.make_optimizer(net)
bucket = T.FlatGradBucket(net.parameters())
net.steps = 1
for _ in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    T.update_weights(opt, net, b, bucket, amp_dtype=amp, hip_convs=(mode == "hip"))
torch.cuda.synchronize()
