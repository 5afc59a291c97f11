"""BatchedMCTS -- n independent MuZero searches, one wavefront per tree, node pool in HBM.

Device-side counterpart of reference main.py:99-196 (run_mcts and helpers), reversi.py:234-251
(Node) and utilities.py:114-129 (MinMaxStats).  The network stays PyTorch: the search hands over
``net_in`` (n,10,8,8) and takes back (value, policy, hidden) once per simulation.
"""
import ctypes as C

import torch

from . import _lib as L
from . import config


# int32 column of each mz_tree_meta field in BatchedMCTS.meta() (include/muzero_amd.h)
META_N_NODES, META_ROOT_N, META_ROOT_W, META_ROOT_PLAYER = 0, 1, 2, 3
META_MM_FLAGS, META_PATH_LEN, META_LEAF_PLAYER, META_ERROR, META_LEAF_PARENT, META_LEAF_ACTION = 8, 9, 10, 11, 12, 13


def search_cfg(cfg=None):
    """mz_search_cfg from the config module (read at call time, like the reference)."""
    c = cfg or config
    kb = c.known_bounds
    return L.SearchCfg(pb_c_base=float(c.pb_c_base), pb_c_init=float(c.pb_c_init), discount=float(c.discount),
                       noise_frac=float(c.root_exploration_fraction), bound_lo=float(kb[0]) if kb else 0.0,
                       bound_hi=float(kb[1]) if kb else 0.0, bounds_none=int(kb is None),
                       tie_lowest=int(bool(getattr(c, "tie_break_lowest", False))),
                       negamax=int(bool(getattr(c, "negamax_value", False))), device_noise=0,
                       noise_alpha=float(getattr(c, "root_dirichlet_alpha", 0.03)), noise_seed=1261)


class BatchedMCTS:
    def __init__(self, n_trees, max_sims, hidden_dtype=torch.float32, device="cuda", cfg=None):
        self.n = int(n_trees)
        self.max_sims = int(max_sims)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise L.MzError("BatchedMCTS needs a GPU device (no CPU fallback)")
        self.hidden_dtype = hidden_dtype
        self.layout = L.PoolLayout()
        L.call("mz_pool_layout_compute", self.n, self.max_sims + 1, {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[hidden_dtype],
               C.byref(self.layout))
        self.pool = torch.empty(self.layout.total_bytes, dtype=torch.uint8, device=self.device)
        assert self.pool.data_ptr() % 256 == 0
        self.cfg = search_cfg(cfg)
        L.call("mz_pool_init", L.dptr(self.pool), C.byref(self.layout), C.byref(self.cfg), L.current_stream())
        self.net_in = torch.zeros((self.n, 10, 8, 8), dtype=hidden_dtype, device=self.device)
        self.sims_done = 0

    # views for the network call (network.py:155-163 takes hidden and action separately)
    @property
    def hidden_in(self):
        return self.net_in[:, :8]

    @property
    def action_in(self):
        return self.net_in[:, 8:]

    def _args(self):
        return L.dptr(self.pool), C.byref(self.layout), C.byref(self.cfg)

    def root_init(self, to_play, legal, policy, hidden, noise=None, pass_ok=None, device_noise_seed=None):
        """play_game main.py:75-83: root expansion over legal actions + exploration noise.
        to_play int8 (n,), legal int64 (n,), policy f32 (n,65), hidden (n,8,8,8), noise f64 (n,65) by action.
        device_noise_seed (with noise=None): draw the Dirichlet noise on the device (mz_search_cfg.device_noise)."""
        if hidden is not None:   # (None: the initial inference wrote the roots' hidden states into their pool slots)
            self._check(policy, hidden)
        cfg = self.cfg
        if device_noise_seed is not None and noise is None:
            cfg = L.SearchCfg.from_buffer_copy(self.cfg)
            cfg.device_noise, cfg.noise_seed = 1, int(device_noise_seed) & (2 ** 64 - 1)
        L.call("mz_mcts_root_init", L.dptr(self.pool), C.byref(self.layout), C.byref(cfg), L.dptr(to_play), L.dptr(legal), L.dptr(pass_ok),
               L.dptr(policy), L.dptr(noise), L.dptr(hidden), self.n, L.current_stream())
        self.sims_done = 0

    def select(self):
        """descent of one simulation for every tree; fills self.net_in"""
        L.call("mz_mcts_select", *self._args(), L.dptr(self.net_in), self.n, L.current_stream())

    def expand_backup(self, value, policy, hidden):
        """hidden None: the new nodes' hidden states are in their pool slots already (routed inference)"""
        if hidden is not None:
            self._check(policy, hidden, value)
        self._count()
        L.call("mz_mcts_expand_backup", *self._args(), L.dptr(value), L.dptr(policy), L.dptr(hidden), self.n,
               L.current_stream())

    # ---- routed form: the inference launch reads / writes hidden states in the pool itself (mz_leaf_route) ----
    def hidden_pool(self):
        """the pool's hidden region as a (n_trees * max_nodes, 512) tensor of the hidden dtype (a view)"""
        o, m = self.layout.off_hidden, self.layout.max_nodes
        eb = self.layout.hidden_elem_bytes
        return self.pool[o:o + self.n * m * 512 * eb].view(self.hidden_dtype).view(self.n * m, 512)

    def route(self):
        if getattr(self, "_route", None) is None:
            self.src_slot = torch.zeros(self.n, dtype=torch.int32, device=self.device)
            self.dst_slot = torch.zeros(self.n, dtype=torch.int32, device=self.device)
            self.act_planes = torch.zeros((self.n, 2, 8, 8), dtype=self.hidden_dtype, device=self.device)
            self._route = L.LeafRoute(self.src_slot.data_ptr(), self.dst_slot.data_ptr(), self.act_planes.data_ptr())
        return self._route

    def select_routed(self):
        L.call("mz_mcts_select_routed", *self._args(), C.byref(self.route()), self.n, L.current_stream())

    def backup_select_routed(self, value, policy):
        self._count()
        L.call("mz_mcts_backup_select_routed", *self._args(), L.dptr(value), L.dptr(policy), C.byref(self.route()), self.n, L.current_stream())

    def backup_select(self, value, policy, hidden):
        """expand_backup of this simulation fused with select of the next one"""
        self._check(policy, hidden, value)
        self._count()
        L.call("mz_mcts_backup_select", *self._args(), L.dptr(value), L.dptr(policy), L.dptr(hidden),
               L.dptr(self.net_in), self.n, L.current_stream())

    def root_stats(self, with_debug=False):
        visits = torch.empty((self.n, 65), dtype=torch.int32, device=self.device)
        root_value = torch.empty(self.n, dtype=torch.float32, device=self.device)
        wsum = torch.empty((self.n, 65), dtype=torch.float32, device=self.device) if with_debug else None
        prior = torch.empty((self.n, 65), dtype=torch.float64, device=self.device) if with_debug else None
        L.call("mz_mcts_root_stats", L.dptr(self.pool), C.byref(self.layout), L.dptr(visits), L.dptr(root_value), L.dptr(wsum), L.dptr(prior),
               self.n, L.current_stream())
        return (visits, root_value, wsum, prior) if with_debug else (visits, root_value)

    def meta(self):
        """(n,16) int32 view of the mz_tree_meta records (tests / debugging)"""
        o = self.layout.off_meta
        return self.pool[o:o + self.n * 64].view(torch.int32).view(self.n, 16)

    def path(self):
        """(n, max_nodes, 4) uint8 view of the path records {u16 node, u8 action, pad}"""
        o = self.layout.off_path
        m = self.layout.max_nodes
        return self.pool[o:o + self.n * m * 4].view(self.n, m, 4)

    def _count(self):
        if self.sims_done >= self.max_sims:
            raise L.MzError("pool sized for %d simulations per tree is full" % self.max_sims)
        self.sims_done += 1

    def _check(self, policy, hidden, value=None):
        if policy.dtype != torch.float32 or tuple(policy.shape) != (self.n, 65):
            raise L.MzError("policy must be float32 (n,65)")
        if hidden.dtype != self.hidden_dtype or hidden.numel() != self.n * 512:
            raise L.MzError("hidden must be %s with n*512 elements" % self.hidden_dtype)
        if value is not None and (value.dtype != torch.float32 or value.numel() != self.n):
            raise L.MzError("value must be float32 with n elements")


class FusedSearch:
    """mz_search -- the whole per-move search of play_game (main.py:75-87) for n trees in ONE C call: initial inference,
    root expansion + noise, S x (recurrent inference + fused tree kernel), root statistics.  Needs an InferenceEngine
    serving through the HIP tower (bf16); the node pool and staging rows live in `workspace` (device scratch)."""

    def __init__(self, engine, n_trees, n_sims, device="cuda", cfg=None, noalloc=False):
        """noalloc=True: no workspace tensor; calls go to mz_search_noalloc, whose scratch is a static device array inside
        the library (limits in include/muzero_amd.h; calls that overlap on one GPU race)"""
        if getattr(engine, "tower", None) != "hip":
            raise L.MzError("FusedSearch needs an InferenceEngine with tower='hip' (bf16)")
        self.engine, self.n, self.S = engine, int(n_trees), int(n_sims)
        self.device = torch.device(device)
        self.cfg = search_cfg(cfg)
        nbytes = L.lib.mz_search_workspace_bytes(self.n, self.S)
        if nbytes < 0:
            raise L.MzError("mz_search_workspace_bytes(%d, %d) rejected its arguments" % (self.n, self.S))
        self.workspace = None if noalloc else torch.empty(nbytes, dtype=torch.uint8, device=self.device)

    def _weights(self):
        pk = self.engine._packs
        return C.byref(pk["repr"]["struct"]), C.byref(pk["dyn"]["struct"]), C.byref(pk["heads"]["struct"])

    def _io(self, image, to_play, legal, n):
        if image.dtype != torch.bfloat16 or tuple(image.shape) != (n, 9, 8, 8):
            raise L.MzError("image must be a bf16 (n,9,8,8) tensor")
        if to_play.dtype != torch.int8 or legal.dtype != torch.int64 or to_play.numel() != n or legal.numel() != n:
            raise L.MzError("to_play must be int8 (n,), legal int64 (n,)")
        visits = torch.empty((n, 65), dtype=torch.int32, device=self.device)
        root_value = torch.empty(n, dtype=torch.float32, device=self.device)
        return visits, root_value

    def __call__(self, image, to_play, legal, noise=None, n_trees=None, n_sims=None):
        n = self.n if n_trees is None else int(n_trees)
        S = self.S if n_sims is None else int(n_sims)
        visits, root_value = self._io(image, to_play, legal, n)
        if self.workspace is None:
            L.call("mz_search_noalloc", *self._weights(), C.byref(self.cfg), L.dptr(image), L.dptr(to_play), L.dptr(legal), L.dptr(noise),
                   S, n, L.dptr(visits), L.dptr(root_value), L.current_stream())
            return visits, root_value
        L.call("mz_search", *self._weights(), C.byref(self.cfg), L.dptr(image), L.dptr(to_play), L.dptr(legal), L.dptr(noise), S, n,
               L.dptr(self.workspace), self.workspace.numel(), L.dptr(visits), L.dptr(root_value), L.current_stream())
        return visits, root_value


def select_action(visits, num_moves, u):
    """select_action + softmax_sample (main.py:128-140, 275-285) for n searches.
    visits int32 (n,65) (-1 = not a child); num_moves int or int32 (n,); u float64 (n,) uniforms."""
    n = visits.shape[0]
    out = torch.empty(n, dtype=torch.uint8, device=visits.device)
    per = num_moves if torch.is_tensor(num_moves) else None
    L.call("mz_select_action", L.dptr(visits), L.dptr(per), 0 if per is not None else int(num_moves), L.dptr(u),
           L.dptr(out), n, L.current_stream())
    return out
