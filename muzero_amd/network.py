"""MuZero networks for 8x8 reversi -- PyTorch-ROCm (the only dense contraction of the engine).

Architecture and ``state_dict`` key names follow reference network.py:35-167 so checkpoints move
between the two code bases unchanged (``representation.conv1.0.weight`` ...
``dynamics.conv.0.weight`` ... ``prediction.value_head.3.bias``); 1,836,526 parameters.
What is added for the batched engine:

  * ``recurrent_inference_packed(x10)`` takes the (B,10,8,8) tensor the select kernel writes
    (hidden planes ++ action planes) so no ``torch.cat`` copy is needed per simulation;
  * ``InferenceEngine`` wraps a network for serving the search: dtype (fp32 for parity runs, bf16
    for throughput), channels_last memory format, fp32 value/policy heads' outputs, and a
    hipGraph (torch.cuda.CUDAGraph) capture of the whole per-simulation step.

Return contract (network.py:146-163): ``(value (B,1), policy (B,65) softmaxed, hidden (B,8,8,8))``.
"""
import copy

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import config

BOARD_CELLS = 64
HIDDEN_PLANES = 8


def _conv3x3(cin, cout, bias=True):
    return nn.Conv2d(cin, cout, kernel_size=3, stride=1, padding=1, bias=bias)


class ResidualBlock(nn.Module):
    """relu(x + conv(relu(conv(x))))  -- network.py:35-47 (no batch norm in the reference)"""

    def __init__(self, filters):
        super().__init__()
        self.block = nn.Sequential(_conv3x3(filters, filters), nn.ReLU(), _conv3x3(filters, filters))

    def forward(self, x):
        return F.relu(x + self.block(x))


def _tower(filters):
    return [ResidualBlock(filters) for _ in range(3)]


class Representation(nn.Module):
    """observation (9,8,8) -> hidden (8,8,8)  -- network.py:49-77"""

    def __init__(self, num_channels=None):
        super().__init__()
        c = num_channels or config.num_channels
        self.conv1 = nn.Sequential(_conv3x3(config.state_shape[0], c), nn.ReLU())
        self.conv2 = nn.Sequential(*_tower(c))
        self.conv3 = nn.Sequential(_conv3x3(c, HIDDEN_PLANES), nn.ReLU())

    def forward(self, x):
        return self.conv3(self.conv2(self.conv1(x)))


class Dynamics(nn.Module):
    """(hidden ++ action planes) (10,8,8) -> next hidden (8,8,8); first/last conv without bias
    -- network.py:113-135"""

    def __init__(self, num_channels=None):
        super().__init__()
        c = num_channels or config.num_channels
        self.conv = nn.Sequential(_conv3x3(HIDDEN_PLANES + 2, c, bias=False), nn.ReLU(), *_tower(c),
                                  _conv3x3(c, HIDDEN_PLANES, bias=False), nn.ReLU())

    def forward(self, rp, a):
        return self.conv(torch.cat((rp, a), dim=1))


class Prediction(nn.Module):
    """hidden -> (policy softmax over 65 actions, value tanh)  -- network.py:79-111"""

    def __init__(self, num_channels=None):
        super().__init__()
        self.board_size = BOARD_CELLS
        self.flatten = nn.Flatten()
        self.policy_head = nn.Sequential(nn.Conv2d(HIDDEN_PLANES, 2, 1, 1), nn.ReLU(), nn.Flatten(),
                                         nn.Linear(2 * BOARD_CELLS, 2 * BOARD_CELLS), nn.ReLU(),
                                         nn.Linear(2 * BOARD_CELLS, config.action_space_size), nn.Softmax(dim=1))
        self.value_head = nn.Sequential(nn.Conv2d(HIDDEN_PLANES, 2, 1, 1), nn.ReLU(), nn.Flatten(),
                                        nn.Linear(2 * BOARD_CELLS, 1), nn.Tanh())

    def forward(self, x):
        return self.policy_head(x), self.value_head(x)


class Network(nn.Module):
    """network.py:137-167"""

    def __init__(self):
        super().__init__()
        self.steps = 0
        self.representation = Representation()
        self.prediction = Prediction()
        self.dynamics = Dynamics()

    def initial_inference(self, image):
        if image.ndim == 3:
            image = image.unsqueeze(0)
        hidden = self.representation(image)
        policy, value = self.prediction(hidden)
        return value, policy, hidden

    def recurrent_inference(self, hidden_state, action):
        if hidden_state.ndim == 3:
            hidden_state = hidden_state.unsqueeze(0)
        if action.ndim == 3:
            action = action.unsqueeze(0)
        hidden = self.dynamics(hidden_state, action)
        policy, value = self.prediction(hidden)
        return value, policy, hidden

    def recurrent_inference_packed(self, x10):
        """same as recurrent_inference(x10[:, :8], x10[:, 8:]) without the concatenation copy"""
        hidden = self.dynamics.conv(x10)
        policy, value = self.prediction(hidden)
        return value, policy, hidden

    def training_steps(self):
        return self.steps


_PERM = (0, 2, 3, 1)   # weight chunk swizzle key of tower_kernels.hip: slot = chunk ^ _PERM[(co >> 2) & 3]


def _pack_stages(w, rows):
    """conv weight (co, 128, 3, 3) -> (36 stages, rows, 4 slots, 8) on w's device: stage = (ky*3+kx)*4 + ci/32, co
    zero-padded to `rows`; the 16-byte chunk c of a row's 32 input channels sits in slot c ^ _PERM[(co >> 2) & 3] -- the
    LDS image tower_kernels.hip reads conflict-free (XOR is an involution, so slot s holds chunk s ^ key).  A handful of
    device ops, no host round trip: publishing new weights costs well under a millisecond."""
    co, dev = w.shape[0], w.device
    full = torch.zeros((rows, 128, 3, 3), dtype=w.dtype, device=dev)
    full[:co] = w
    st = full.permute(2, 3, 1, 0).reshape(9, 4, 32, rows).permute(0, 1, 3, 2).reshape(36, rows, 4, 8)   # (stage, co, chunk, 8)
    key = torch.tensor(_PERM, device=dev)[(torch.arange(rows, device=dev) >> 2) & 3]
    idx = torch.arange(4, device=dev).unsqueeze(0) ^ key.unsqueeze(1)                                    # (rows, 4): chunk held by each slot
    return torch.gather(st, 2, idx.view(1, rows, 4, 1).expand(36, rows, 4, 8))


def _pack_last(w):
    """last conv weight (8, 128, 3, 3) -> (4 quarters, 80 rows, 4 slots, 8): the nine taps sit in the GEMM's M dimension, row =
    tap * 8 + plane (72 real rows, padded to five 16-row MFMA tiles); 64-byte rows with the chunk swizzle of _pack_stages."""
    dev = w.device
    rows = torch.zeros((80, 128), dtype=w.dtype, device=dev)
    rows[:72] = w.permute(2, 3, 0, 1).reshape(72, 128)                       # (ky, kx, plane, ci) -> row = (ky * 3 + kx) * 8 + plane
    st = rows.reshape(80, 4, 4, 8).permute(1, 0, 2, 3)                       # (quarter, row, chunk, 8)
    key = torch.tensor(_PERM, device=dev)[(torch.arange(80, device=dev) >> 2) & 3]
    idx = torch.arange(4, device=dev).unsqueeze(0) ^ key.unsqueeze(1)        # (80, 4): chunk held by each slot
    return torch.gather(st, 2, idx.view(1, 80, 4, 1).expand(4, 80, 4, 8)).contiguous()


def pack_tower(first, blocks, last, device, cin):
    """Prepack one tower (first conv, three ResidualBlocks, last conv -- network.py:54-68 / :118-128) into the
    bf16 LDS images mz_tower_forward streams (include/muzero_amd.h mz_tower_weights).  Returns a dict of device
    tensors (kept alive by the caller) holding the layout documented there."""
    bf = torch.bfloat16
    dev = torch.device(device)
    f = lambda t: t.detach().to(dev).float()  # noqa: E731
    wf = torch.zeros((9, 128, 16), dtype=torch.float32, device=dev)
    wf[:, :, :cin] = f(first.weight).permute(2, 3, 0, 1).reshape(9, 128, cin)
    convs = [c for blk in blocks for c in (blk.block[0], blk.block[2])]
    w_mid = torch.cat([torch.stack([_pack_stages(f(c.weight), 128) for c in convs]).reshape(-1),
                       torch.zeros(4 * 128 * 32, device=dev)])   # + 4 stages of padding: the stream prefetches past the end
    return dict(w_first=wf.to(bf).contiguous(), b_first=None if first.bias is None else f(first.bias).contiguous(),
                w_mid=w_mid.to(bf).contiguous(), b_mid=torch.stack([f(c.bias) for c in convs]).contiguous(),
                w_last=_pack_last(f(last.weight)).to(bf).contiguous(),
                b_last=None if last.bias is None else f(last.bias).contiguous(), cin=int(cin))


class InferenceEngine:
    """Serves a Network to the batched search: fixed batch size, fixed dtype / memory format.

    ``initial(image)`` and ``recurrent(x10)`` return (value f32 (B,), policy f32 (B,65), hidden
    (B,8,8,8) of ``dtype``, contiguous NCHW) -- exactly what the tree kernels consume.

    ``fused=True`` (default, needs channels_last) keeps MIOpen for the convolutions but replaces the
    bias add / residual add / ReLU passes PyTorch runs after each of them by ONE in-place HIP pass
    (``mz_bias_act_nhwc``): same arithmetic in fp32, one rounding instead of up to three.
    """

    def __init__(self, network, batch, dtype=torch.float32, channels_last=True, device="cuda", fused=True, tower="auto"):
        self.device = torch.device(device)
        self.batch = int(batch)
        self.dtype = dtype
        self.net = copy.deepcopy(network).to(self.device).eval()   # the caller's module (fp32 master weights) is left untouched
        if dtype != torch.float32:
            self.net = self.net.to(dtype)
        self.mf = torch.channels_last if channels_last else torch.contiguous_format
        if channels_last:
            self.net = self.net.to(memory_format=torch.channels_last)
        for p in self.net.parameters():
            p.requires_grad_(False)
        # tower = "hip": both convolution towers run as the LDS-resident MFMA kernel (mz_tower_forward, bf16 only);
        # "miopen": PyTorch/MIOpen convolutions (+ the fused epilogue when fused=True).  "auto" = hip for bf16 on a GPU.
        if tower == "auto":
            tower = "hip" if (dtype == torch.bfloat16 and self.device.type == "cuda") else "miopen"
        # "hip_x3": fp32 serving with the towers' 128 -> 128 convolutions on the hand-written kernels in split precision (three
        # bf16 MFMA products per fp32 product, muzero_amd/convgrad.py): the reference's arithmetic to ~1e-5, faster than MIOpen's fp32
        if tower not in ("hip", "miopen", "hip_x3"):
            raise ValueError("tower must be 'auto', 'hip', 'miopen' or 'hip_x3'")
        if tower == "hip" and dtype != torch.bfloat16:
            raise ValueError("the HIP tower serves bf16 only")
        if tower == "hip_x3" and (dtype != torch.float32 or channels_last):
            raise ValueError("tower='hip_x3' serves fp32 in NCHW: pass dtype=torch.float32, channels_last=False")
        self.tower = tower
        self.fused = bool(fused and channels_last and self.device.type == "cuda" and tower != "hip_x3")
        if self.fused or tower == "hip":
            from . import _lib
            self._L = _lib
            self._dt = {torch.float32: _lib.MZ_F32, torch.bfloat16: _lib.MZ_BF16, torch.float16: _lib.MZ_F16}[dtype]
        if tower == "hip":
            self._packs = {}
            self._repack()
        if tower == "hip_x3":
            self._packs3 = {}
            self._repack_x3()

    # ---- split-precision tower (fp32 serving) ------------------------------------------------------
    def _repack_x3(self):
        """the towers in split precision: the residual blocks' weight stream (hi, lo per convolution), the first / last convolutions'
        fragments and the biases, (re)built IN PLACE (a captured hipGraph holds their addresses)"""
        from .convgrad import restower_x3_pack, tower_x3_pack_ends
        for name, (first, blocks, last, _cin) in self._tower_parts().items():
            convs = [blk.block[ci] for blk in blocks for ci in (0, 2)]
            if name not in self._packs3:
                f32 = lambda k: torch.empty(k, dtype=torch.float32, device=self.device)  # noqa: E731
                self._packs3[name] = (restower_x3_pack([c.weight for c in convs]), tower_x3_pack_ends(first.weight, last.weight), f32((6, 128)),
                                      None if first.bias is None else f32(128), None if last.bias is None else f32(8))
            packed, ends, bias, b_first, b_last = self._packs3[name]
            restower_x3_pack([c.weight for c in convs], out=packed)
            tower_x3_pack_ends(first.weight, last.weight, out=ends)
            for i, c in enumerate(convs):
                bias[i].copy_(c.bias.detach())
            if b_first is not None:
                b_first.copy_(first.bias.detach())
            if b_last is not None:
                b_last.copy_(last.bias.detach())
        # the prediction heads in fp32 (the Linear matrices as nn.Linear holds them), copies refilled in place
        from . import _lib as L
        ph, vh = self.net.prediction.policy_head, self.net.prediction.value_head
        f = lambda t: t.detach().float().contiguous()  # noqa: E731
        heads = dict(pc_w=f(ph[0].weight.reshape(2, 8)), pc_b=f(ph[0].bias), p1_w=f(ph[3].weight), p1_b=f(ph[3].bias),
                     p2_w=f(ph[5].weight), p2_b=f(ph[5].bias), vc_w=f(vh[0].weight.reshape(2, 8)), vc_b=f(vh[0].bias),
                     v1_w=f(vh[3].weight.reshape(128)), v1_b=f(vh[3].bias))
        if "heads" not in self._packs3:
            heads = {k: t.clone() for k, t in heads.items()}
            heads["struct"] = L.HeadsWeights(*[heads[k].data_ptr() for k, _ in L.HeadsWeights._fields_])
            self._packs3["heads"] = heads
        else:
            for k, t in heads.items():
                self._packs3["heads"][k].copy_(t)

    def _tower_x3(self, name, x):
        """the whole tower -- first convolution, three residual blocks, last convolution -- as ONE LDS-resident launch in split
        precision (mz_tower_x3_forward): planes in, hidden state out"""
        from .convgrad import tower_x3
        packed, ends, bias, b_first, b_last = self._packs3[name]
        return tower_x3(x.contiguous(), packed, ends, bias, b_first, b_last)

    def _infer_x3(self, name, x):
        """network.initial_inference / recurrent_inference (network.py:146-163) at fp32 grade in ONE launch: the tower in split
        precision and the heads in plain fp32 behind it (mz_inference_x3_forward)"""
        from . import _lib as L
        packed, ends, bias, b_first, b_last = self._packs3[name]
        x = x.contiguous()
        cin = self._tower_parts()[name][3]
        if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4 or tuple(x.shape[1:]) != (cin, 8, 8):
            raise L.MzError("hip_x3 input must be a cuda float32 (n,%d,8,8) tensor, got %s %s" % (cin, tuple(x.shape), x.dtype))
        n = x.shape[0]
        hidden = torch.empty((n, HIDDEN_PLANES, 8, 8), dtype=torch.float32, device=x.device)
        value = torch.empty((n,), dtype=torch.float32, device=x.device)
        policy = torch.empty((n, config.action_space_size), dtype=torch.float32, device=x.device)
        L.call("mz_inference_x3_forward", L.dptr(packed), L.dptr(ends), L.dptr(bias), L.dptr(b_first), L.dptr(b_last), cin,
               L.C.byref(self._packs3["heads"]["struct"]), L.dptr(x), L.dptr(hidden), L.dptr(value), L.dptr(policy), n, L.current_stream())
        return value, policy, hidden

    # ---- HIP tower ---------------------------------------------------------------------------------
    def _tower_parts(self):
        r, d = self.net.representation, self.net.dynamics.conv
        return {"repr": (r.conv1[0], list(r.conv2), r.conv3[0], config.state_shape[0]),
                "dyn": (d[0], [d[2], d[3], d[4]], d[5], HIDDEN_PLANES + 2)}

    def _pack_heads(self):
        """prediction heads (network.py:79-111) as the arrays of mz_heads_weights: f32, except the two Linear matrices,
        which go as bf16 [out][in] (the serving network's weights ARE bf16 values: lossless)"""
        ph, vh = self.net.prediction.policy_head, self.net.prediction.value_head
        f = lambda t: t.detach().float().contiguous()  # noqa: E731
        lin = lambda w: w.detach().to(torch.bfloat16).contiguous()  # noqa: E731
        return dict(pc_w=f(ph[0].weight.reshape(2, 8)), pc_b=f(ph[0].bias), p1_w=lin(ph[3].weight), p1_b=f(ph[3].bias),
                    p2_w=lin(ph[5].weight), p2_b=f(ph[5].bias), vc_w=f(vh[0].weight.reshape(2, 8)), vc_b=f(vh[0].bias),
                    v1_w=f(vh[3].weight.reshape(128)), v1_b=f(vh[3].bias))

    def _repack(self):
        """(re)build the prepacked weight images IN PLACE (a captured hipGraph holds their addresses)"""
        L = self._L
        heads = self._pack_heads()
        if "heads" not in self._packs:
            heads["struct"] = L.HeadsWeights(*[heads[k].data_ptr() for k, _ in L.HeadsWeights._fields_])
            self._packs["heads"] = heads
        else:
            for k, t in heads.items():
                self._packs["heads"][k].copy_(t)
        for name, (first, blocks, last, cin) in self._tower_parts().items():
            new = pack_tower(first, blocks, last, self.device, cin)
            if name not in self._packs:  # noqa: SIM102
                ptr = lambda t: None if t is None else t.data_ptr()  # noqa: E731
                new["struct"] = L.TowerWeights(ptr(new["w_first"]), ptr(new["b_first"]), ptr(new["w_mid"]), ptr(new["b_mid"]),
                                               ptr(new["w_last"]), ptr(new["b_last"]), new["cin"], None)
                self._packs[name] = new
            else:
                for k, t in new.items():
                    if torch.is_tensor(t):
                        self._packs[name][k].copy_(t)

    def _tower_hip(self, name, x):
        L = self._L
        pk = self._packs[name]
        if x.dtype != torch.bfloat16 or not x.is_contiguous() or tuple(x.shape[1:]) != (pk["cin"], 8, 8):
            raise L.MzError("HIP tower input must be a contiguous bf16 (n,%d,8,8) tensor" % pk["cin"])
        hidden = torch.empty((x.shape[0], HIDDEN_PLANES, 8, 8), dtype=torch.bfloat16, device=x.device)
        L.call("mz_tower_forward", L.C.byref(pk["struct"]), L.dptr(x), L.dptr(hidden), L.MZ_BF16, x.shape[0], L.current_stream())
        return hidden

    def output_buffers(self, n=None):
        """(value, policy, hidden) tensors a caller may pass as `out=` to initial() / recurrent() of the HIP path: a search that
        consumes each result before asking for the next (selfplay) reuses ONE set instead of taking three fresh tensors per
        call -- no allocator work in the eager paths, and under graph capture every simulation step writes the same 5 MB
        instead of S different ones"""
        n = self.batch if n is None else n
        return (torch.empty(n, dtype=torch.float32, device=self.device),
                torch.empty((n, config.action_space_size), dtype=torch.float32, device=self.device),
                torch.empty((n, HIDDEN_PLANES, 8, 8), dtype=torch.bfloat16, device=self.device))

    def _infer_hip(self, name, x, out=None):
        """tower + prediction heads in one launch (mz_inference_forward) -> (value, policy, hidden)"""
        L = self._L
        pk = self._packs[name]
        x = x.contiguous()
        if x.dtype != torch.bfloat16 or tuple(x.shape[1:]) != (pk["cin"], 8, 8):
            raise L.MzError("HIP tower input must be a bf16 (n,%d,8,8) tensor" % pk["cin"])
        n = x.shape[0]
        if out is None:
            out = self.output_buffers(n)
        value, policy, hidden = out
        if (value.shape != (n,) or policy.shape != (n, config.action_space_size) or hidden.shape != (n, HIDDEN_PLANES, 8, 8) or
                value.dtype != torch.float32 or policy.dtype != torch.float32 or hidden.dtype != torch.bfloat16 or
                not (value.is_contiguous() and policy.is_contiguous() and hidden.is_contiguous()) or value.device != x.device):
            raise L.MzError("out= must be contiguous (value f32 (n,), policy f32 (n,65), hidden bf16 (n,8,8,8)) on the input's device")
        L.call("mz_inference_forward", L.C.byref(pk["struct"]), L.C.byref(self._packs["heads"]["struct"]), L.dptr(x), L.dptr(hidden),
               L.MZ_BF16, L.dptr(value), L.dptr(policy), n, L.current_stream())
        return value, policy, hidden

    def infer_routed(self, name, x, dst_pool, value, policy, src_pool=None, src_slot=None, dst_slot=None, dst_stride=1):
        """mz_inference_forward_routed: the same launch with its hidden states read from / written into rows of a pool (the
        search pool's hidden region) -- x holds only the planes behind the 8 hidden planes when src_pool is given"""
        L = self._L
        pk = self._packs[name]
        n = value.shape[0]
        planes = pk["cin"] - (8 if src_pool is not None else 0)
        if x.dtype != torch.bfloat16 or tuple(x.shape) != (n, planes, 8, 8) or not x.is_contiguous():
            raise L.MzError("routed HIP tower input must be a contiguous bf16 (n,%d,8,8) tensor" % planes)
        L.call("mz_inference_forward_routed", L.C.byref(pk["struct"]), L.C.byref(self._packs["heads"]["struct"]), L.dptr(x), L.dptr(src_pool),
               L.dptr(src_slot), L.dptr(dst_pool), L.dptr(dst_slot), int(dst_stride), L.MZ_BF16, L.dptr(value), L.dptr(policy), n,
               L.current_stream())
        return value, policy

    @torch.no_grad()
    def load_weights(self, state_dict):
        """publish new weights (SharedStorage.save_network -> run_selfplay's load_state_dict,
        main.py:60-61,216-221) IN PLACE: dtype/memory-format casts happen in copy_, and the parameter
        storage a captured hipGraph points at stays where it is."""
        own = self.net.state_dict()
        missing = set(own) ^ set(state_dict)
        if missing:
            raise KeyError("state_dict mismatch: %s" % sorted(missing))
        for k, t in own.items():
            t.copy_(state_dict[k])
        if self.tower == "hip":
            self._repack()
        if self.tower == "hip_x3":
            self._repack_x3()

    # ---- fused tower: conv (MIOpen, no bias) + one epilogue pass ---------------------------------
    def _epi(self, y, bias, residual=None):
        L = self._L
        if not y.is_contiguous(memory_format=torch.channels_last) or (residual is not None and not residual.is_contiguous(memory_format=torch.channels_last)):
            raise L.MzError("fused inference needs channels_last activations")
        import ctypes as C
        ptr = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
        L.call("mz_bias_act_nhwc", ptr(y), ptr(bias), ptr(residual), y.numel(), y.shape[1], self._dt, 1, L.current_stream())
        return y

    def _conv(self, x, conv, residual=None):
        return self._epi(F.conv2d(x, conv.weight, None, 1, 1), conv.bias, residual)

    def _tower(self, x, first, blocks, last):
        y = self._conv(x, first)
        for blk in blocks:
            t = self._conv(y, blk.block[0])
            y = self._conv(t, blk.block[2], residual=y)   # relu(x + block(x)), network.py:46-47
        return self._conv(y, last)

    def _hidden_initial(self, image):
        r = self.net.representation
        if self.tower == "hip":
            return self._tower_hip("repr", image.contiguous())
        if self.tower == "hip_x3":
            return self._tower_x3("repr", image)
        image = image.contiguous(memory_format=self.mf)
        if not self.fused:
            return r(image)
        return self._tower(image, r.conv1[0], list(r.conv2), r.conv3[0])

    def _hidden_recurrent(self, x10):
        d = self.net.dynamics.conv
        if self.tower == "hip":
            return self._tower_hip("dyn", x10.contiguous())
        if self.tower == "hip_x3":
            return self._tower_x3("dyn", x10)
        x10 = x10.contiguous(memory_format=self.mf)
        if not self.fused:
            return d(x10)
        return self._tower(x10, d[0], [d[2], d[3], d[4]], d[5])

    def _finish(self, hidden):
        policy, value = self.net.prediction(hidden)
        return (value.float().reshape(-1).contiguous(), policy.float().contiguous(),
                hidden.contiguous(memory_format=torch.contiguous_format))

    @torch.no_grad()
    def initial(self, image, out=None):
        """out: see output_buffers() (HIP path only; the MIOpen path returns what PyTorch allocates)"""
        if self.tower == "hip":
            return self._infer_hip("repr", image.to(self.dtype), out)
        if self.tower == "hip_x3":
            return self._infer_x3("repr", image.to(self.dtype))
        return self._finish(self._hidden_initial(image.to(self.dtype)))

    @torch.no_grad()
    def recurrent(self, x10, out=None):
        if self.tower == "hip":
            return self._infer_hip("dyn", x10, out)
        if self.tower == "hip_x3":
            return self._infer_x3("dyn", x10)
        return self._finish(self._hidden_recurrent(x10))
