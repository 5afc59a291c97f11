"""ReplayBuffer -- reference utilities.ReplayBuffer (utilities.py:46-109) with the window resident in
HBM and the training batch built by HIP kernels (mz_replay_save / mz_replay_sample_batch).

Same methods and meaning as the reference: ``save_game``, ``ready``, ``sample_batch``,
``sample_position``, ``count``; knobs ``config.window_size``, ``config.batch_size``,
``config.num_unroll_steps`` read where the reference reads them.  The random draws are the reference's
own (``random.sample`` over the window, then ``random.randint`` per picked game, utilities.py:76,104),
made on the host from ``random`` so a seeded run picks the same (game, position) pairs; everything
that touches game data runs on the device.  ``save_games`` is the batched entry the self-play engine
uses (n lock-step games per call, FIFO order = game order).

``sample_batch`` returns device fp32 tensors shaped as the reference's numpy arrays:
image (B,9,8,8), actions (K,B,2,8,8), target_values (K+1,B), target_policies (K+1,B,65).
"""
import random

import numpy as np
import torch

from . import _lib as L
from . import config


class ReplayBuffer:
    def __init__(self, device="cuda", window_size=None, batch_size=None, max_moves=None):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.MzError("ReplayBuffer needs a GPU device (no CPU fallback)")
        self.window_size = int(window_size if window_size is not None else config.window_size)  # utilities.py:50
        self.batch_size = int(batch_size if batch_size is not None else config.batch_size)      # utilities.py:51
        self.max_moves = int(max_moves if max_moves is not None else config.max_moves)
        W, T, d = self.window_size, self.max_moves, self.device
        self.white = torch.zeros((W, T + 1), dtype=torch.int64, device=d)
        self.black = torch.zeros((W, T + 1), dtype=torch.int64, device=d)
        self.actions = torch.zeros((W, T), dtype=torch.uint8, device=d)
        self.rewards = torch.zeros((W, T), dtype=torch.int8, device=d)
        self.policy = torch.zeros((W, T, 65), dtype=torch.float32, device=d)
        self.root_values = torch.zeros((W, T), dtype=torch.float32, device=d)
        self.length = torch.zeros(W, dtype=torch.int32, device=d)
        self.refused = torch.zeros(1, dtype=torch.int32, device=d)   # samples mz_replay_sample_batch refused (device counter)
        self._rb = L.Replay(self.white.data_ptr(), self.black.data_ptr(), self.actions.data_ptr(), self.rewards.data_ptr(),
                            self.policy.data_ptr(), self.root_values.data_ptr(), self.length.data_ptr(), W, T, self.refused.data_ptr())
        self.buffer_size = 0   # utilities.py:52
        self.counter = 0       # utilities.py:54
        self._head = 0         # slot of buffer[0] (the oldest game)
        self._lengths = [0] * W  # host mirror of `length` by slot: sample_position needs len(action_history)

    # ---- FIFO bookkeeping (utilities.py:56-63) ------------------------------------------------------
    def _claim(self, n):
        """slots for n new games appended in order; evicts the oldest when full (del buffer[0])"""
        if n > self.window_size:
            raise L.MzError("cannot save %d games into a window of %d" % (n, self.window_size))
        slot0 = (self._head + self.buffer_size) % self.window_size
        over = self.buffer_size + n - self.window_size
        if over > 0:
            self._head = (self._head + over) % self.window_size
            self.buffer_size -= over
        self.buffer_size += n
        return slot0

    def save_games(self, rec, lengths=None):
        """n games at once from step-major trajectory tensors (selfplay.GameRecords or the dict
        shard.gather_game_records returns).  lengths: per-game int32 tensor / list, default rec.length."""
        get = (lambda k: rec[k]) if isinstance(rec, dict) else (lambda k: getattr(rec, k))
        hw, hb, acts, rews, visits, rv = (get(k) for k in ("hist_white", "hist_black", "actions", "rewards", "visits", "root_values"))
        n = acts.shape[1]
        # step-major (T, n, ...) batches; a trimmed view of a wider gather buffer is fine as long as every field shares
        # the row stride (mz_replay_save takes it): only a view that is not row-contiguous gets copied
        stride = hw.stride(0) if hw.dim() == 2 and hw.stride(1) == 1 else None
        ok = stride is not None and all(t.stride(0) == stride * (65 if t.dim() == 3 else 1) and t.stride(1) == (65 if t.dim() == 3 else 1)
                                        for t in (hb, acts, rews, visits, rv)) and (visits.dim() == 3 and visits.stride(2) == 1)
        if not ok:
            hw, hb, acts, rews, visits, rv = (t.contiguous() for t in (hw, hb, acts, rews, visits, rv))
            stride = hw.shape[1]
        if hw.shape[0] != self.max_moves + 1 or acts.shape[0] != self.max_moves:
            raise L.MzError("records are for max_moves=%d, buffer for %d" % (acts.shape[0], self.max_moves))
        if lengths is None:
            lengths = rec["length"] if isinstance(rec, dict) and "length" in rec else getattr(rec, "length", self.max_moves)
        per_game = None
        if torch.is_tensor(lengths) or isinstance(lengths, (list, tuple, np.ndarray)):
            host = [int(x) for x in (lengths.cpu().tolist() if torch.is_tensor(lengths) else lengths)]
            per_game = torch.tensor(host, dtype=torch.int32, device=self.device)
        else:
            host = [int(lengths)] * n
        if len(host) != n or any(ln < 0 or ln > self.max_moves for ln in host):
            raise L.MzError("game lengths must be %d values in [0, %d]" % (n, self.max_moves))
        slot0 = self._claim(n)
        vis_dt = {torch.int32: L.VIS_I32, torch.int16: L.VIS_U16, torch.uint16: L.VIS_U16, torch.uint8: L.VIS_U8}.get(visits.dtype)
        if vis_dt is None:
            raise L.MzError("visits must be int32 (-1 = not a root child), int16/uint16 or uint8 counts, got %s" % visits.dtype)
        _ptr = lambda t: L.C.c_void_p(t.data_ptr())  # noqa: E731  (strided views: checked above; L.dptr insists on contiguous tensors)
        if not all(t.is_cuda for t in (hw, hb, acts, rews, visits, rv)):
            raise L.MzError("muzero_amd kernels need device tensors (no CPU fallback)")
        L.call("mz_replay_save", L.C.byref(self._rb), slot0, _ptr(hw), _ptr(hb), _ptr(acts), _ptr(rews), _ptr(visits), vis_dt,
               _ptr(rv), L.dptr(per_game), 0 if per_game is not None else host[0], stride, n, L.current_stream())
        for i, ln in enumerate(host):
            self._lengths[(slot0 + i) % self.window_size] = ln
        self.counter += sum(host)

    def save_game(self, game):
        """utilities.py:56-63 for one reference-shaped Game (compat.Game or the reference's own):
        state_history of (2,8,8) planes, action_history, rewards, child_visits (fractions), root_values."""
        T, n = self.max_moves, len(game.action_history)
        if n > T:
            raise L.MzError("game of %d moves exceeds max_moves=%d" % (n, T))
        bits = np.zeros((2, T + 1), np.uint64)
        wts = np.uint64(1) << np.arange(64, dtype=np.uint64)
        for t, obs in enumerate(game.state_history):
            o = np.asarray(obs).reshape(2, 64).astype(bool)
            bits[0, t], bits[1, t] = wts[o[0]].sum(dtype=np.uint64), wts[o[1]].sum(dtype=np.uint64)
        acts = np.zeros(T, np.uint8)
        acts[:n] = [a.index for a in game.action_history]
        rews = np.zeros(T, np.int8)
        rews[:n] = game.rewards
        pol = np.zeros((T, 65), np.float32)
        pol[:n] = np.asarray(game.child_visits, np.float64)  # the fp32 rounding torch.FloatTensor applies (main.py:236)
        rv = np.zeros(T, np.float32)
        rv[:n] = [float(v) for v in game.root_values]
        slot = self._claim(1)
        d = self.device
        self.white[slot].copy_(torch.from_numpy(bits[0].view(np.int64)).to(d))
        self.black[slot].copy_(torch.from_numpy(bits[1].view(np.int64)).to(d))
        self.actions[slot].copy_(torch.from_numpy(acts).to(d))
        self.rewards[slot].copy_(torch.from_numpy(rews).to(d))
        self.policy[slot].copy_(torch.from_numpy(pol).to(d))
        self.root_values[slot].copy_(torch.from_numpy(rv).to(d))
        self.length[slot] = n
        self._lengths[slot] = n
        self.counter += n

    def ready(self):  # utilities.py:65-69
        return self.buffer_size >= self.batch_size

    def game_length(self, idx):
        """len(buffer[idx].action_history), idx = position in the FIFO (0 = oldest)"""
        return self._lengths[(self._head + idx) % self.window_size]

    def sample_position(self, idx):
        """utilities.py:101-104 (reads config.num_unroll_steps, like the reference, not the argument)"""
        return random.randint(0, self.game_length(idx) - config.num_unroll_steps)

    def sample_batch(self, num_unroll_steps=None, game_idxes=None, positions=None):
        """utilities.py:72-99.  game_idxes / positions override the draws (tests replay recorded ones)."""
        K = int(config.num_unroll_steps if num_unroll_steps is None else num_unroll_steps)
        if game_idxes is None:
            game_idxes = random.sample(range(self.buffer_size), self.batch_size)
        if positions is None:
            positions = [self.sample_position(i) for i in game_idxes]
        B = len(game_idxes)
        for i, p in zip(game_idxes, positions):
            if not (0 <= i < self.buffer_size) or p < 0 or p + K > self.game_length(i):
                raise L.MzError("sample (game %d, position %d) with K=%d outside a game of %d moves" % (i, p, K, self.game_length(i)))
        d = self.device
        sp = torch.tensor([[(self._head + i) % self.window_size for i in game_idxes], list(positions)], dtype=torch.int32).to(d)
        image = torch.empty((B, 9, 8, 8), dtype=torch.float32, device=d)
        actions = torch.empty((K, B, 2, 8, 8), dtype=torch.float32, device=d)
        tv = torch.empty((K + 1, B), dtype=torch.float32, device=d)
        tp = torch.empty((K + 1, B, 65), dtype=torch.float32, device=d)
        L.call("mz_replay_sample_batch", L.C.byref(self._rb), L.dptr(sp[0]), L.dptr(sp[1]), B, K, L.dptr(image), L.dptr(actions),
               L.dptr(tv), L.dptr(tp), L.current_stream())
        return image, actions, tv, tp

    def refused_samples(self):
        """samples the device refused since construction (a host/device length mismatch; synchronises)"""
        return int(self.refused.item())

    def count(self):  # utilities.py:106-109
        temp = self.counter
        self.counter = 0
        return temp
