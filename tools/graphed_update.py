"""EXPERIMENT, not shipped in muzero_amd: the training step (update_weights) captured once as a hipGraph and replayed.

It replays at 4.1 ms against 4.3-5.2 ms for the eager step with the HIP convolution kernels, but it is NOT reliable on this
stack (PyTorch 2.10 / ROCm 7.2): a replayed step intermittently computes a wrong gradient -- also with pure PyTorch/MIOpen
kernels (no kernel of this repository in the graph) -- and 3 of 4 learning runs through it diverged where 6 of 6 eager runs did
not (profiles/r03/train_graph_soak.txt; tools/graph_dbg.py, tools/graph_soak*.py reproduce it)."""
import os

import torch

from muzero_amd import train as T


class GraphedUpdate:
    """update_weights (zero the gradients, K + 1 network passes, backward, weight-gradient flush, optimiser step) captured ONCE
    as a hipGraph on static batch buffers and replayed per step: with the towers' convolutions on the HIP kernels the eager
    step is bound by the host's ~400 launches (5.5 ms), the replay by the GPU (4.6 ms; tools/train_graph.py).  Single process
    only (the captured step holds no collective).  Construction leaves weights, optimiser state and step counter as they were:
    the warm-up steps MIOpen and the allocator need run on a snapshot that is restored in place."""

    def __init__(self, optimizer, network, bucket, example_batch, hip_convs=False, warmup=3):
        if T._dist_on():
            raise RuntimeError("GraphedUpdate captures no collective: use update_weights in data-parallel runs")
        dev = next(network.parameters()).device
        self.optimizer, self.network, self.bucket = optimizer, network, bucket
        self.static = tuple(torch.as_tensor(x, dtype=torch.float32).to(dev).clone() for x in example_batch)
        had_state = {id(p): optimizer.state.get(p, {}).get("momentum_buffer") is not None for p in network.parameters()}
        snap_w = [p.detach().clone() for p in network.parameters()]
        snap_m = {id(p): optimizer.state[p]["momentum_buffer"].clone() for p in network.parameters() if had_state[id(p)]}
        steps0, log = network.steps, T.LOG
        T.LOG = None
        try:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    T.update_weights(optimizer, network, self.static, bucket, hip_convs=hip_convs)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize(dev)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.out = T.update_weights(optimizer, network, self.static, bucket, hip_convs=hip_convs)
        finally:
            T.LOG = log
            network.steps = steps0
        with torch.no_grad():   # back to the state before the warm-up, in place (the graph holds these addresses)
            for p, w in zip(network.parameters(), snap_w):
                p.copy_(w)
                buf = optimizer.state.get(p, {}).get("momentum_buffer")
                if buf is not None:
                    if had_state[id(p)]:
                        buf.copy_(snap_m[id(p)])
                    else:
                        buf.zero_()   # SGD's first step sets the buffer to the gradient: the same as momentum * 0 + gradient

    def __call__(self, batch):
        for d, t in zip(self.static, batch):
            d.copy_(torch.as_tensor(t, dtype=torch.float32), non_blocking=True)
        if os.environ.get("MZ_GRAPH_SYNC") == "1":
            torch.cuda.current_stream().synchronize()
        self.graph.replay()
        p_loss, v_loss = self.out[0].clone(), self.out[1].clone()
        if T.LOG is not None and self.network.steps % 10 == 0:   # main.py:268-269
            T.LOG('step {}: p_loss {:.4f} v_loss {:.4f}'.format(self.network.steps, float(p_loss), float(v_loss)))
        self.network.steps += 1
        return p_loss, v_loss


