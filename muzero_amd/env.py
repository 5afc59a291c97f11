"""BatchedEnv -- n reversi games as struct-of-arrays device tensors, stepped by HIP kernels.

Device-side counterpart of reference ``reversi.Environment`` (reversi.py:39-205): the fields are the
reference's attributes (board as two bitboards, player, steps, done, legal_actions as a mask).
"""
import torch

from . import _lib as L


def _u64(t):
    return t.view(torch.int64)  # torch has no uint64 arithmetic; bit patterns are what matters


class BatchedEnv:
    def __init__(self, n, device="cuda"):
        self.n = int(n)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.MzError("BatchedEnv needs a GPU device (no CPU fallback)")
        z = lambda dt: torch.zeros(self.n, dtype=dt, device=self.device)  # noqa: E731
        self.black, self.white, self.legal = z(torch.int64), z(torch.int64), z(torch.int64)
        self.to_play, self.steps, self.done = z(torch.int8), z(torch.uint8), z(torch.uint8)
        self.reward, self.status = z(torch.int8), z(torch.uint8)
        self.reset()

    def _state(self):
        return [L.dptr(t) for t in (self.black, self.white, self.to_play, self.steps, self.done, self.legal)]

    def reset(self):
        """Environment.__init__ (reversi.py:41-58)"""
        L.call("mz_env_reset", *self._state(), self.n, L.current_stream())

    def step(self, action):
        """Environment.step (reversi.py:130-180) for all games. action: uint8 (n,) device tensor.
        Returns (reward int8 (n,), status uint8 (n,)); done/legal/to_play are updated in place."""
        if action.dtype != torch.uint8 or action.numel() != self.n:
            raise L.MzError("action must be a uint8 tensor of %d elements" % self.n)
        L.call("mz_env_step", *self._state(), L.dptr(action), L.dptr(self.reward), L.dptr(self.status), self.n,
               L.current_stream())
        return self.reward, self.status

    # ---- the packed form: to_play / steps / done / action / reward / status in one int32 word per game (mz_env_step_packed) ----
    def pack_meta(self, action):
        """meta word of every game for playing `action` (uint8 (n,)): bits 0-6 steps | 7 WHITE to move | 8 done | 9-15 action"""
        return (self.steps.to(torch.int32) | ((self.to_play == 1).to(torch.int32) << 7) | (self.done.to(torch.int32) << 8) |
                (action.to(torch.int32) << 9)).contiguous()

    def step_packed(self, meta):
        """Environment.step on the packed record, in place (boards, legal, meta).  Returns (reward int8, status uint8) unpacked
        from the word and refreshes to_play / steps / done from it."""
        if meta.dtype != torch.int32 or meta.numel() != self.n:
            raise L.MzError("meta must be an int32 tensor of %d elements" % self.n)
        L.call("mz_env_step_packed", L.dptr(self.black), L.dptr(self.white), L.dptr(self.legal), L.dptr(meta), self.n, L.current_stream())
        self.steps.copy_((meta & 127).to(torch.uint8))
        self.to_play.copy_(torch.where((meta >> 7) & 1 == 1, 1, -1).to(torch.int8))
        self.done.copy_(((meta >> 8) & 1).to(torch.uint8))
        self.reward.copy_((((meta >> 16) & 3) - 1).to(torch.int8))
        self.status.copy_(((meta >> 18) & 3).to(torch.uint8))
        return self.reward, self.status

    def legal_mask(self):
        """recompute legal placements for the side to move (update_legal_actions, reversi.py:119-126)"""
        out = torch.empty_like(self.legal)
        L.call("mz_env_legal_mask", L.dptr(self.black), L.dptr(self.white), L.dptr(self.to_play), L.dptr(out), self.n,
               L.current_stream())
        return out

    def random_action(self, seed, game0, step, out=None):
        out = torch.empty(self.n, dtype=torch.uint8, device=self.device) if out is None else out
        L.call("mz_env_random_action", seed, game0, step, L.dptr(self.legal), L.dptr(out), self.n, L.current_stream())
        return out

    def random_playout(self, seed, game0=0, n_steps=60, voluntary=0, want_digest=False, want_actions=False):
        """config 2: whole random games in registers; overwrites the state with the final one."""
        digest = torch.empty(self.n, dtype=torch.int64, device=self.device) if want_digest else None
        actions = torch.empty((self.n, n_steps), dtype=torch.uint8, device=self.device) if want_actions else None
        L.call("mz_env_random_playout", seed, game0, self.n, n_steps, voluntary, *self._state(), L.dptr(digest),
               L.dptr(actions), L.dptr(self.reward), L.current_stream())
        return digest, actions, self.reward


DT = {torch.float32: L.MZ_F32, torch.bfloat16: L.MZ_BF16, torch.float16: L.MZ_F16}


def action_encode(action, player, dtype=torch.float32):
    """Action.encode (reversi.py:32-36) for n actions -> (n,2,8,8)"""
    n = action.numel()
    out = torch.empty((n, 2, 8, 8), dtype=dtype, device=action.device)
    L.call("mz_action_encode", L.dptr(action), L.dptr(player), L.dptr(out), DT[dtype], n, L.current_stream())
    return out


def make_image(hist_white, hist_black, state_index, n, dtype=torch.float32, out=None):
    """Game.make_image (reversi.py:301-312). hist_*: (61, stride) int64 step-major observation history;
    state_index: int (all games) or int32 (n,) tensor."""
    stride = hist_white.shape[1]
    out = torch.empty((n, 9, 8, 8), dtype=dtype, device=hist_white.device) if out is None else out
    per_game = state_index if torch.is_tensor(state_index) else None
    L.call("mz_make_image", L.dptr(hist_white), L.dptr(hist_black), stride, L.dptr(per_game),
           0 if per_game is not None else int(state_index), L.dptr(out), DT[dtype], n, L.current_stream())
    return out
