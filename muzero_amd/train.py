"""Training side of the loop: reference main.py:201-272 (train_network / update_weights), the
checkpoint format of main.py:223-224 plus a resume the reference lacks, and the one collective of
the whole system -- a single flat-bucket gradient all-reduce (RCCL over xGMI on the GPUs, gloo in
the CPU tests).

The networks and the optimiser stay PyTorch (north_star); what is build-specific:

  * ``FlatGradBucket``: every parameter's ``.grad`` is a view into ONE contiguous fp32 buffer
    (1,836,526 elements = 7.35 MB), so data-parallel training costs exactly one all-reduce per step.
    On an 8-GPU xGMI ring that is latency-dominated, hence one bucket, not many.
  * ``update_weights`` keeps the reference's signature and loss, accepts the device tensors
    ``muzero_amd.replay.ReplayBuffer.sample_batch`` returns (or the reference's numpy arrays), and
    returns the two loss terms instead of only printing them.
  * every rank applies the same averaged gradient to the same weights, so the replicas stay
    identical without a weight broadcast; ``broadcast_weights`` exists for (re)joining ranks / resume.
"""
import os
import random
import time

import torch
import torch.distributed as dist
import torch.optim as optim

from . import config


def _dist_on(group=None):
    return dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1


class FlatGradBucket:
    """all gradients of `params` as views of one contiguous buffer -> one all-reduce per step"""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        p0 = self.params[0]
        self.flat = torch.zeros(sum(p.numel() for p in self.params), dtype=p0.dtype, device=p0.device)
        self.attach()

    def attach(self):
        o = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[o:o + n].view_as(p)
            o += n

    def zero(self):
        """optimizer.zero_grad() that keeps the views (set_to_none would detach them)"""
        self.flat.zero_()
        if any(p.grad is None or p.grad.data_ptr() < self.flat.data_ptr() or
               p.grad.data_ptr() >= self.flat.data_ptr() + self.flat.numel() * self.flat.element_size() for p in self.params):
            self.attach()

    def all_reduce_mean(self, group=None):
        """average over the data-parallel ranks: each rank's loss is a mean over ITS batch, so the
        mean of the gradients is the gradient of the mean over the global batch (equal shard sizes)"""
        if not _dist_on(group):
            return
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        self.flat.div_(dist.get_world_size(group))


def make_optimizer(network):
    """train_network's optimiser (main.py:206): SGD lr 1e-3, momentum config.momentum and -- a quirk of
    the reference kept on purpose -- weight_decay = config.lr_decay_rate (0.1)"""
    return optim.SGD(network.parameters(), lr=0.001, weight_decay=config.lr_decay_rate, momentum=config.momentum)


LOG = print   # progress lines as the reference prints them; set to None to silence (bench.py does: its stdout is one JSON line)


def muzero_loss(network, batch):
    """main.py:239-265: K-step unrolled loss.  Returns (p_loss, v_loss) tensors.
    p_loss = mean over (B, K+1) of -sum_a pi log p ; v_loss = mean over B of sum_k (z - v)^2."""
    dev = next(network.parameters()).device
    image, actions, target_values, target_policies = (torch.as_tensor(x, dtype=torch.float32).to(dev) for x in batch)
    value, policy, hidden_state = network.initial_inference(image)
    p_value, p_policy = [value], [policy]
    for action in actions:
        value, policy, hidden_state = network.recurrent_inference(hidden_state, action)
        p_value.append(value)
        p_policy.append(policy)
    p_value = torch.stack(p_value).squeeze(-1)      # (K+1, B); the reference's bare .squeeze() also drops B == 1
    p_policy = torch.stack(p_policy)                # (K+1, B, 65)
    target_policies = target_policies.transpose(0, 1)
    p_policy = p_policy.transpose(0, 1)
    target_values = target_values.transpose(0, 1)
    p_value = p_value.transpose(0, 1)
    p_loss = torch.mean(torch.sum(-target_policies * torch.log(p_policy), dim=2))
    v_loss = torch.mean(torch.sum((target_values - p_value) ** 2, dim=1))
    return p_loss, v_loss


# ---- the same loss, organised for the GPU ---------------------------------------------------------------------------------
# main.py:247-252 calls the Dynamics and Prediction networks K (+1) times per step, so autograd computes every one of their
# weight gradients K (+1) times on a batch of B and adds the pieces up: per step ~40 weight-gradient convolutions with their
# layout transposes, ~120 accumulation adds and ~44 bias reductions, most of them short kernels with idle gaps between them
# (profiles/r02/train_step_breakdown.txt).  muzero_loss_batched computes the SAME function:
#   * the prediction heads run ONCE on the (K+1) B hidden states (the heads see each sample independently: no batch norm);
#   * a Dynamics convolution's weight (and bias) gradient is computed in ONE call over the concatenation of its K uses: the
#     backward of each use returns only the input gradient and parks (input, grad_output); flush_weight_grads() then runs one
#     weight-gradient convolution per layer.
# Same sums in another order: fp32 rounding differences only (tests/test_gpu_train.py holds it to the reference's recorded
# gradients with the tolerances of the plain form).
class _ConvDeferW(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, conv, sink, packed):
        ctx.sink = sink
        ctx.packed = packed
        if packed is not None:   # the HIP kernels (muzero_amd/convgrad.py): operands rounded to bf16, fp32 sums
            from .convgrad import conv3x3
            x = x.contiguous()
            ctx.save_for_backward(x)
            return conv3x3(x, packed[0], conv.bias, packed_lo=packed[2] if len(packed) > 2 else None)
        ctx.save_for_backward(x, conv.weight)
        return torch.ops.aten.convolution(x, conv.weight, conv.bias, [1, 1], [1, 1], [1, 1], False, [0, 0], 1)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        gx = None
        if ctx.packed is not None:
            from .convgrad import conv3x3
            x, = ctx.saved_tensors
            if ctx.needs_input_grad[0]:
                gx = conv3x3(g, ctx.packed[1], packed_lo=ctx.packed[3] if len(ctx.packed) > 2 else None)
        else:
            x, w = ctx.saved_tensors
            if ctx.needs_input_grad[0]:
                gx = torch.ops.aten.convolution_backward(g, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0]
        ctx.sink.append((x, g))
        return gx, None, None, None


class DeferredWeightGrads:
    """parks the (input, grad_output) pairs of every use of a convolution during backward; flush() = one weight-gradient call
    per convolution over its uses, accumulated into .grad (views of a FlatGradBucket stay views).
    hip=True: the 3x3, 128 -> 128 convolutions run on the hand-written matrix-core kernels -- forward and data gradient through
    mz_conv3x3_forward, weight and bias gradients through mz_conv3x3_wgrad (operands rounded to bf16, fp32 sums, no
    concatenation) -- instead of MIOpen's fp32 kernels."""

    def __init__(self, hip=False):
        if hip not in (False, True, "bf16", "x3"):
            raise ValueError("hip must be False, True / 'bf16' or 'x3'")
        self.sinks = {}
        self.hip = bool(hip)
        self.x3 = hip == "x3"      # split precision: three bf16 MFMA products per fp32 product (muzero_amd/convgrad.py)
        self.packed = {}

    def _is_tower_conv(self, conv, x):
        return self.hip and conv.in_channels == 128 and conv.out_channels == 128 and x.is_cuda and x.dtype == torch.float32

    def pack(self, conv):
        packed = self.packed.get(id(conv))
        if packed is None:   # once per step: the weights change with every optimiser step
            from .convgrad import pack_conv
            packed = self.packed[id(conv)] = pack_conv(conv.weight, split=self.x3)
        return packed

    def prepack(self, convs):
        """the packed images of all the step's tower convolutions in ONE launch (two in split precision)"""
        convs = [c for c in convs if id(c) not in self.packed]
        if self.hip and convs and convs[0].weight.is_cuda:
            from .convgrad import pack_convs
            for c, p in zip(convs, pack_convs([c.weight for c in convs], split=self.x3)):
                self.packed[id(c)] = p

    def park(self, conv, x, g):
        self.sinks.setdefault(id(conv), (conv, []))[1].append((x, g))

    def conv(self, x, conv):
        sink = self.sinks.setdefault(id(conv), (conv, []))[1]
        return _ConvDeferW.apply(x, conv, sink, self.pack(conv) if self._is_tower_conv(conv, x) else None)

    @torch.no_grad()
    def flush(self):
        st = None
        for conv, sink in self.sinks.values():
            if not sink:
                continue
            has_b = conv.bias is not None
            if self._is_tower_conv(conv, sink[0][0]):
                from .convgrad import conv3x3_wgrad, conv3x3_wgrad_unchecked
                for prm in (conv.weight, conv.bias):
                    if prm is not None and prm.grad is None:
                        prm.grad = torch.zeros_like(prm)
                gw, gb = conv.weight.grad, conv.bias.grad if has_b else None
                if (len(sink) <= 8 and gw.is_contiguous() and gw.dtype == torch.float32 and (gb is None or gb.is_contiguous())
                        and all(x.is_contiguous() and g.is_contiguous() and x.shape == g.shape and x.dtype == g.dtype == torch.float32 for x, g in sink)):
                    if st is None:
                        from . import _lib as L
                        st = L.current_stream()
                    conv3x3_wgrad_unchecked(st, [x for x, _ in sink], [g for _, g in sink], gw, gb, x3=self.x3)
                else:   # the checked call says what is wrong
                    conv3x3_wgrad([x.contiguous() for x, _ in sink], [g for _, g in sink], gw, gb, accumulate=True, x3=self.x3)
                sink.clear()
                continue
            X = torch.cat([x for x, _ in sink]) if len(sink) > 1 else sink[0][0]
            G = torch.cat([g for _, g in sink]) if len(sink) > 1 else sink[0][1]
            _, gw, gb = torch.ops.aten.convolution_backward(G, X, conv.weight, [conv.out_channels] if has_b else None, [1, 1], [1, 1], [1, 1],
                                                            False, [0, 0], 1, [False, True, has_b])
            for prm, gr in ((conv.weight, gw), (conv.bias, gb)):
                if prm is None or gr is None:
                    continue
                if prm.grad is None:
                    prm.grad = gr
                else:
                    prm.grad.add_(gr)
            sink.clear()


class _TowerHip(torch.autograd.Function):
    """a tower's residual blocks relu(x + conv(relu(conv(x)))) (network.py:35-47) on the HIP kernels with every elementwise
    operation in a convolution's store phase (mz_conv3x3_fused): forward  t = relu(conv1(y)), y' = relu(y + conv2(t));
    backward  g_t = dgrad2(g_pre) * (t > 0),  g_pre' = (g_pre + dgrad1(g_t)) * (y_prev > 0)  -- the skip path's gradient and the
    previous block's ReLU gate in one epilogue.  Weight gradients are parked in dw as for _ConvDeferW."""

    @staticmethod
    def forward(ctx, h0, blocks, dw):
        from . import _lib as L
        from .convgrad import conv3x3_unchecked as conv
        y = h0.contiguous()        # (n,128,8,8) f32 on the GPU: _tower_deferred has checked (dw._is_tower_conv)
        st, x3 = L.current_stream(), dw.x3
        saved, packs = [y], []
        for blk in blocks:
            c1, c2 = blk.block[0], blk.block[2]
            p1, p2 = dw.pack(c1), dw.pack(c2)
            t = conv(st, y, p1[0], c1.bias, relu=True, x3=x3)
            y = conv(st, t, p2[0], c2.bias, relu=True, skip=y, x3=x3)
            saved += [t, y]
            packs.append((p1, p2))
        ctx.save_for_backward(*saved)
        ctx.blocks, ctx.dw, ctx.packs = blocks, dw, packs
        return y

    @staticmethod
    def backward(ctx, g):
        from . import _lib as L
        from .convgrad import conv3x3_unchecked as conv
        st, x3 = L.current_stream(), ctx.dw.x3
        saved = ctx.saved_tensors                                  # h0, t1, y1, t2, y2, t3, y3
        g_pre = torch.ops.aten.threshold_backward(g.contiguous(), saved[-1], 0)   # the last block's ReLU gate
        for k in range(len(ctx.blocks) - 1, -1, -1):
            c1, c2 = ctx.blocks[k].block[0], ctx.blocks[k].block[2]
            p1, p2 = ctx.packs[k]
            y_prev, t = saved[2 * k], saved[2 * k + 1]
            g_t = conv(st, g_pre, p2[1], mask=t, x3=x3)
            ctx.dw.park(c2, t, g_pre)
            g_prev = conv(st, g_t, p1[1], skip=g_pre, mask=y_prev if k > 0 else None, x3=x3)   # h0's own gate belongs to the caller's ReLU
            ctx.dw.park(c1, y_prev, g_t)
            g_pre = g_prev
        return g_pre, None, None


def _tower_deferred(blocks, h, dw):
    blocks = tuple(blocks)
    if dw.hip and h.requires_grad and all(dw._is_tower_conv(c, h) for blk in blocks for c in (blk.block[0], blk.block[2])):
        return _TowerHip.apply(h, blocks, dw)
    for blk in blocks:
        t = torch.relu(dw.conv(h, blk.block[0]))
        h = torch.relu(h + dw.conv(t, blk.block[2]))
    return h


def _representation_deferred(rep, image, dw):
    """Representation (network.py:49-77) with every convolution's weight gradient deferred"""
    h = rep.conv1(image)                      # its input carries no gradient: plain autograd (a deferred convolution hangs its weight gradient on the input's)
    h = _tower_deferred(rep.conv2, h, dw)
    return torch.relu(dw.conv(h, rep.conv3[0]))


def _dynamics_deferred(dyn, x, dw):
    """Dynamics.conv (network.py:113-135) with every convolution's weight gradient deferred"""
    seq = dyn.conv
    h = torch.relu(dw.conv(x, seq[0]))
    h = _tower_deferred((seq[2], seq[3], seq[4]), h, dw)
    return torch.relu(dw.conv(h, seq[5]))


def muzero_loss_batched(network, batch, dw):
    """muzero_loss with the heads batched over the unroll and the Dynamics weight gradients deferred into dw
    (call dw.flush() after backward).  Returns (p_loss, v_loss)."""
    dev = next(network.parameters()).device
    image, actions, target_values, target_policies = (torch.as_tensor(x, dtype=torch.float32).to(dev) for x in batch)
    if dw.hip and image.is_cuda:
        dw.prepack([c for blk in list(network.representation.conv2) + [network.dynamics.conv[i] for i in (2, 3, 4)] for c in (blk.block[0], blk.block[2])
                    if c.in_channels == 128 and c.out_channels == 128])
    hidden = _representation_deferred(network.representation, image, dw) if dw.hip else network.representation(image)
    hiddens = [hidden]
    for action in actions:
        hidden = _dynamics_deferred(network.dynamics, torch.cat((hidden, action), dim=1), dw)
        hiddens.append(hidden)
    K1, B = len(hiddens), image.shape[0]
    p_policy, p_value = network.prediction(torch.cat(hiddens))        # ((K+1) B, 65), ((K+1) B, 1): rows k * B + b
    p_policy = p_policy.view(K1, B, -1)
    p_value = p_value.view(K1, B)
    p_loss = torch.mean(torch.sum(-target_policies * torch.log(p_policy), dim=2))   # mean over (K+1, B) = mean over (B, K+1)
    v_loss = torch.sum((target_values - p_value) ** 2) / B                          # mean over B of the sum over k
    return p_loss, v_loss


def update_weights(optimizer, network, batch, bucket=None, group=None, amp_dtype=None, loss_scale=1.0, batched=None, hip_convs=False):
    """main.py:230-272.  With `bucket` (FlatGradBucket over network.parameters()) the gradients are
    averaged over the ranks of `group` by ONE all-reduce between backward and optimizer.step.
    amp_dtype (e.g. torch.bfloat16): run the forward/backward convolutions under torch.autocast (fp32 master
    weights, gradients and optimiser state stay fp32); None = the reference's fp32 arithmetic.
    loss_scale: data-parallel ranks with UNEQUAL local batches (sharded replay window) pass b_local * world / B_global, so the
    mean over ranks of the scaled gradients is the gradient of the mean over the global batch; batch=None = this rank drew
    no sample this step (it contributes a zero gradient and still joins the all-reduce).  The returned losses are unscaled.
    batched: use muzero_loss_batched (heads once over the unroll, one weight-gradient call per Dynamics convolution); default:
    on a GPU in fp32, the plain reference-shaped muzero_loss elsewhere.
    hip_convs: the gradients of the towers' 3x3, 128 -> 128 convolutions through the hand-written matrix-core kernels
    (muzero_amd/convgrad.py: operands rounded to bf16, fp32 sums; fp32 master weights and optimiser) instead of MIOpen's fp32
    kernels; implies batched.  "x3": the same kernels in split precision (every fp32 product as three bf16 MFMA products: close to
    fp32, tests/test_gpu_train.py holds it to the reference's recorded gradients).  False = the reference's fp32 arithmetic."""
    if bucket is not None:
        bucket.zero()
    else:
        optimizer.zero_grad()
    if batch is None:
        if bucket is None:
            raise RuntimeError("an empty local batch needs a FlatGradBucket (the zero gradient must exist to be all-reduced)")
        dev = next(network.parameters()).device
        p_loss = v_loss = torch.zeros((), device=dev)
    else:
        if amp_dtype is not None:
            dev = next(network.parameters()).device
            with torch.autocast(dev.type, dtype=amp_dtype):
                p_loss, v_loss = muzero_loss(network, batch)
            p_loss, v_loss = p_loss.float(), v_loss.float()
        else:
            if batched is None:   # the GPU organisation where it pays: on a GPU, in the reference's fp32 arithmetic
                batched = next(network.parameters()).is_cuda and os.environ.get("MZ_TRAIN_PLAIN") != "1"
            if hip_convs:
                if not next(network.parameters()).is_cuda:   # no CPU stand-in for the HIP kernels
                    raise RuntimeError("update_weights(hip_convs=%r) needs the network on a GPU: the hand-written convolution kernels have no CPU fallback" % (hip_convs,))
                batched = True
            dw = DeferredWeightGrads(hip=hip_convs if hip_convs else False) if batched else None
            p_loss, v_loss = muzero_loss_batched(network, batch, dw) if batched else muzero_loss(network, batch)
        total = p_loss + v_loss
        (total if loss_scale == 1.0 else total * float(loss_scale)).backward()
        if amp_dtype is None and dw is not None:
            dw.flush()
    if bucket is not None:
        bucket.all_reduce_mean(group)
    elif _dist_on(group):
        raise RuntimeError("data-parallel update_weights needs a FlatGradBucket")
    optimizer.step()
    if LOG is not None and network.steps % 10 == 0:   # main.py:268-269; the float() is a device sync
        LOG('step {}: p_loss {:.4f} v_loss {:.4f}'.format(network.steps, float(p_loss.detach()), float(v_loss.detach())))
    network.steps += 1
    return p_loss.detach(), v_loss.detach()


class SharedStorage:
    """network.py:169-178 without Ray: holds the latest published weights in-process"""

    def __init__(self, network_id=None):
        self.network_id = network_id

    def get_network(self):
        return self.network_id

    def save_network(self, network_id):
        self.network_id = network_id


def broadcast_weights(network, src=0, group=None):
    """one flat broadcast of all parameters and buffers from `src` (joining ranks, resume)"""
    if not _dist_on(group):
        return
    tensors = [t.data for t in list(network.parameters()) + list(network.buffers())]
    flat = torch.cat([t.reshape(-1).float() for t in tensors])
    dist.broadcast(flat, src=src, group=group)
    o = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[o:o + n].view_as(t))
        o += n


# ---- checkpoints: main.py:223-224 format, plus resume ----------------------------------------------
def checkpoint_path(step, directory="./models"):
    return os.path.join(directory, "{}model.pth".format(step))  # main.py:224


def save_checkpoint(network, step, directory="./models", optimizer=None):
    """the reference's file: torch.save(network.state_dict()) under ./models/{step}model.pth, loadable
    by its test.py:247-250.  With `optimizer`, a sidecar {step}train.pth carries what a resume needs."""
    os.makedirs(directory, exist_ok=True)
    path = checkpoint_path(step, directory)
    torch.save({k: v.detach().cpu() for k, v in network.state_dict().items()}, path)
    if optimizer is not None:
        torch.save({"optimizer": optimizer.state_dict(), "steps": int(network.steps)}, os.path.join(directory, "{}train.pth".format(step)))
    return path


def load_checkpoint(network, path, optimizer=None):
    """load a reference-format checkpoint (weights_only: nothing in the file is executed); resume the
    optimiser/step counter from the sidecar when it exists"""
    sd = torch.load(path, map_location="cpu", weights_only=True)
    network.load_state_dict(sd)
    side = path.replace("model.pth", "train.pth")
    if os.path.exists(side):
        st = torch.load(side, map_location="cpu", weights_only=True)
        network.steps = int(st["steps"])
        if optimizer is not None:
            optimizer.load_state_dict(st["optimizer"])
    return network


def train_network(network, optimizer, storage, replay_buffer, training_steps=None, device=None, group=None,
                  checkpoint_dir="./models", sleep=None, on_publish=None):
    """main.py:201-227: sample a batch, update, publish the weights every config.checkpoint_interval
    steps, save every 1000 steps.  `optimizer` is replaced exactly as the reference does (main.py:206).
    `sleep` defaults to 0 (the reference sleeps config.training_interval = 0.5 s per step to let its CPU
    actors keep up; pass it to reproduce that pacing).  Returns the list of (p_loss, v_loss)."""
    network.train()
    if device is not None:
        network.to(device)
    optimizer = make_optimizer(network)
    bucket = FlatGradBucket(network.parameters())
    print('start training')
    losses = []
    steps = int(config.training_steps if training_steps is None else training_steps)
    world = dist.get_world_size(group) if _dist_on(group) else 1
    rank = dist.get_rank(group) if world > 1 else 0
    if world > 1 and replay_buffer.batch_size % world:
        raise ValueError("global batch %d is not divisible by %d ranks" % (replay_buffer.batch_size, world))
    for i in range(1, steps + 1):
        if world > 1:
            # data parallel: every rank makes the SAME draws (`random` seeded identically on all ranks, main.py:19) and
            # trains on its slice of them, so N ranks see exactly the global batch one process would (loop.generation)
            idx = random.sample(range(replay_buffer.buffer_size), replay_buffer.batch_size)   # utilities.py:76
            pos = [replay_buffer.sample_position(g) for g in idx]                            # utilities.py:78
            sl = slice(rank * len(idx) // world, (rank + 1) * len(idx) // world)
            batch = replay_buffer.sample_batch(config.num_unroll_steps, idx[sl], pos[sl])
        else:
            batch = replay_buffer.sample_batch(config.num_unroll_steps)
        losses.append(update_weights(optimizer, network, batch, bucket, group))
        if i % config.checkpoint_interval == 0:
            state_dict = network.state_dict()
            if storage is not None:
                storage.save_network(state_dict)
            if on_publish is not None:
                on_publish(state_dict)
        if i % 1000 == 0 and (not _dist_on(group) or dist.get_rank(group) == 0):
            save_checkpoint(network, i, checkpoint_dir, optimizer)
        if sleep:
            time.sleep(sleep)
    return losses
