"""ctypes loader for the CPU oracle (oracle/mz_oracle.c).  TEST INFRASTRUCTURE ONLY.

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg -- never from the
product package (muzero_amd/).  See the header of mz_oracle.c for what is restated and how it is
pinned to the reference.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "libmz_oracle.so")


def build(force=False):
    src = os.path.join(HERE, "mz_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", HERE] + (["-B"] if force else []))
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        u64, i64, p = C.c_uint64, C.c_int64, C.c_void_p
        L.mzo_playout_rand.restype = u64
        L.mzo_playout_rand.argtypes = [u64, u64, u64]
        L.mzo_playout_action.restype = C.c_int
        L.mzo_playout_action.argtypes = [u64, u64, u64, u64]
        L.mzo_digest_step.restype = u64
        L.mzo_digest_step.argtypes = [u64, u64, u64, C.c_int, C.c_int, C.c_int, C.c_int, u64, C.c_int]
        L.mzo_env_reset.argtypes = [p] * 6 + [C.c_long]
        L.mzo_env_legal_mask.argtypes = [p] * 4 + [C.c_long]
        L.mzo_env_step.argtypes = [p] * 9 + [C.c_long]
        L.mzo_env_random_playout.argtypes = [u64, u64, C.c_long, C.c_int, C.c_int] + [p] * 9
        L.mzo_action_encode.argtypes = [C.c_int, C.c_int, p]
        L.mzo_make_image.argtypes = [p, p, C.c_int, p]
        L.mzo_tree_new.restype = p
        L.mzo_tree_new.argtypes = [C.c_int]
        L.mzo_tree_free.argtypes = [p]
        L.mzo_tree_config.argtypes = [p, C.c_double, C.c_double, C.c_double, C.c_int, C.c_double, C.c_double,
                                      C.c_double, C.c_int, C.c_int]
        L.mzo_ucb_score_probe.restype = C.c_double
        L.mzo_ucb_score_probe.argtypes = [p, C.c_int, C.c_int, p]
        L.mzo_root_init.argtypes = [p, C.c_int, u64, C.c_int, p, p]
        L.mzo_select.restype = C.c_int
        L.mzo_select.argtypes = [p, p, p, p, p]
        L.mzo_expand_backup.argtypes = [p, C.c_float, p]
        L.mzo_root_stats.argtypes = [p] * 6
        L.mzo_child_visits.argtypes = [p, p]
        L.mzo_select_action.restype = C.c_int
        L.mzo_select_action.argtypes = [p, C.c_int, C.c_double]
        L.mzo_hashnet_image_hash.restype = i64
        L.mzo_hashnet_image_hash.argtypes = [p]
        L.mzo_hashnet_action_code.restype = i64
        L.mzo_hashnet_action_code.argtypes = [C.c_int, C.c_int]
        L.mzo_hashnet_step.restype = i64
        L.mzo_hashnet_step.argtypes = [i64, i64]
        L.mzo_hashnet_outputs.argtypes = [i64, C.c_int, p, p]
        L.mzo_search_hashnet.argtypes = [p, C.c_int, u64, C.c_int, p, i64, C.c_int, C.c_int, p, p, p]
        L.mzo_play_game_hashnet.restype = C.c_int
        L.mzo_play_game_hashnet.argtypes = [C.c_int, C.c_int] + [p] * 8
        L.mzo_controller_image.argtypes = [p, p, C.c_int, C.c_int, p]
        L.mzo_softmax_sample.restype = C.c_int
        L.mzo_softmax_sample.argtypes = [p, C.c_double, C.c_double]
        L.mzo_arena_game_hashnet.restype = C.c_int
        L.mzo_arena_game_hashnet.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, p, C.c_int, p, p, p]
        L.mzo_replay_sample.argtypes = [p] * 5 + [C.c_int] * 3 + [p] * 4
        L.mzo_root_init_batch.argtypes = [p, C.c_int, p, p, p, p]
        L.mzo_select_batch.argtypes = [p, C.c_int, p, p, p]
        L.mzo_expand_backup_batch.argtypes = [p, C.c_int, p, p]
        L.mzo_root_visits_batch.argtypes = [p, C.c_int, p]
        _lib = L
    return _lib


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


# ---------------------------------------------------------------------------------------------
# env
# ---------------------------------------------------------------------------------------------
class EnvBatch:
    """SoA batch of reversi envs on the host (same field layout as muzero_amd's device batch)."""

    def __init__(self, n):
        self.n = n
        self.black = np.zeros(n, np.uint64)
        self.white = np.zeros(n, np.uint64)
        self.to_play = np.zeros(n, np.int8)
        self.steps = np.zeros(n, np.uint8)
        self.done = np.zeros(n, np.uint8)
        self.legal = np.zeros(n, np.uint64)
        self.reset()

    def reset(self):
        lib().mzo_env_reset(ptr(self.black), ptr(self.white), ptr(self.to_play), ptr(self.steps), ptr(self.done),
                            ptr(self.legal), self.n)

    def step(self, action):
        action = np.ascontiguousarray(action, np.uint8)
        reward = np.zeros(self.n, np.int8)
        status = np.zeros(self.n, np.uint8)
        lib().mzo_env_step(ptr(self.black), ptr(self.white), ptr(self.to_play), ptr(self.steps), ptr(self.done),
                           ptr(self.legal), ptr(action), ptr(reward), ptr(status), self.n)
        return reward, status


def legal_mask(black, white, to_play):
    black = np.ascontiguousarray(black, np.uint64)
    white = np.ascontiguousarray(white, np.uint64)
    to_play = np.ascontiguousarray(to_play, np.int8)
    out = np.zeros(black.shape[0], np.uint64)
    lib().mzo_env_legal_mask(ptr(black), ptr(white), ptr(to_play), ptr(out), black.shape[0])
    return out


def random_playout(seed, game0, n, n_steps=60, voluntary=0, want_actions=False):
    e = EnvBatch(n)
    digest = np.zeros(n, np.uint64)
    actions = np.zeros((n, n_steps), np.uint8) if want_actions else None
    last_reward = np.zeros(n, np.int8)
    lib().mzo_env_random_playout(seed, game0, n, n_steps, voluntary, ptr(e.black), ptr(e.white), ptr(e.to_play),
                                 ptr(e.steps), ptr(e.done), ptr(e.legal), ptr(digest), ptr(actions), ptr(last_reward))
    return e, digest, actions, last_reward


def action_encode(index, player):
    out = np.zeros(128, np.uint8)
    lib().mzo_action_encode(index, player, ptr(out))
    return out.reshape(2, 8, 8)


def make_image(hist_white, hist_black, state_index):
    hw = np.ascontiguousarray(hist_white, np.uint64)
    hb = np.ascontiguousarray(hist_black, np.uint64)
    out = np.zeros((9, 8, 8), np.float32)
    lib().mzo_make_image(ptr(hw), ptr(hb), state_index, ptr(out))
    return out


# ---------------------------------------------------------------------------------------------
# search
# ---------------------------------------------------------------------------------------------
class Tree:
    def __init__(self, max_sims, pb_c_base=19652, pb_c_init=1.25, discount=1.0, known_bounds=(-1, 1),
                 noise_frac=0.25, tie_lowest=False, negamax=False):
        self.h = lib().mzo_tree_new(max_sims)
        lo, hi = known_bounds if known_bounds is not None else (0.0, 0.0)
        lib().mzo_tree_config(self.h, pb_c_base, pb_c_init, discount, int(known_bounds is None), lo, hi, noise_frac,
                              int(tie_lowest), int(negamax))

    def __del__(self):
        if getattr(self, "h", None):
            lib().mzo_tree_free(self.h)
            self.h = None

    def root_init(self, to_play, legal, pass_flag, policy, noise=None):
        policy = np.ascontiguousarray(policy, np.float32)
        noise = None if noise is None else np.ascontiguousarray(noise, np.float64)
        lib().mzo_root_init(self.h, int(to_play), int(legal), int(pass_flag), ptr(policy), ptr(noise))

    def select(self):
        """-> (depth, parent_expanded_id, action, plane, path_actions)"""
        pid, a, pl = C.c_int(), C.c_int(), C.c_int()
        path = np.zeros(512, np.int32)
        d = lib().mzo_select(self.h, C.byref(pid), C.byref(a), C.byref(pl), ptr(path))
        return d, pid.value, a.value, pl.value, path[:d].copy()

    def expand_backup(self, value, policy):
        policy = np.ascontiguousarray(policy, np.float32)
        lib().mzo_expand_backup(self.h, float(np.float32(value)), ptr(policy))

    def root_stats(self):
        visits = np.zeros(65, np.int32)
        wsum = np.zeros(65, np.float32)
        prior = np.zeros(65, np.float64)
        rn = C.c_int()
        rv = C.c_float()
        lib().mzo_root_stats(self.h, ptr(visits), ptr(wsum), ptr(prior), C.byref(rn), C.byref(rv))
        return visits, wsum, prior, rn.value, np.float32(rv.value)

    def ucb_score(self, node, action):
        is_t = C.c_int()
        s = lib().mzo_ucb_score_probe(self.h, node, action, C.byref(is_t))
        return s, bool(is_t.value)

    def search_hashnet(self, to_play, legal, pass_flag, noise, h0, mode, S):
        noise = None if noise is None else np.ascontiguousarray(noise, np.float64)
        node_h = np.zeros(S + 1, np.int64)
        depth = np.zeros(S, np.int32)
        leaf_a = np.zeros(S, np.uint8)
        lib().mzo_search_hashnet(self.h, int(to_play), int(legal), int(pass_flag), ptr(noise), int(h0), int(mode), S,
                                 ptr(node_h), ptr(depth), ptr(leaf_a))
        return node_h, depth, leaf_a


def child_visits(visits):
    visits = np.ascontiguousarray(visits, np.int32)
    out = np.zeros(65, np.float64)
    lib().mzo_child_visits(ptr(visits), ptr(out))
    return out


def select_action(visits, num_moves, u):
    visits = np.ascontiguousarray(visits, np.int32)
    return lib().mzo_select_action(ptr(visits), int(num_moves), float(u))


def hashnet_outputs(h, mode):
    v = C.c_float()
    pol = np.zeros(65, np.float32)
    lib().mzo_hashnet_outputs(int(h), int(mode), C.byref(v), ptr(pol))
    return np.float32(v.value), pol


def play_game_hashnet(S, mode, noise, u):
    noise = np.ascontiguousarray(noise, np.float64)
    u = np.ascontiguousarray(u, np.float64)
    actions = np.zeros(60, np.uint8)
    rewards = np.zeros(60, np.int8)
    visits = np.zeros((60, 65), np.int32)
    root_values = np.zeros(60, np.float32)
    hw = np.zeros(61, np.uint64)
    hb = np.zeros(61, np.uint64)
    n = lib().mzo_play_game_hashnet(S, mode, ptr(noise), ptr(u), ptr(actions), ptr(rewards), ptr(visits),
                                    ptr(root_values), ptr(hw), ptr(hb))
    return n, actions, rewards, visits, root_values, hw, hb


# ---------------------------------------------------------------------------------------------
# replay buffer batch (utilities.py:72-99) / make_target (reversi.py:314-341)
# ---------------------------------------------------------------------------------------------
def replay_sample(hist_white, hist_black, actions, rewards, visits, length, pos, K):
    """one (game, position) -> (image (9,8,8), action planes (K,2,8,8), target values (K+1,),
    target policies (K+1,65)), all float32"""
    hw = np.ascontiguousarray(hist_white, np.uint64)
    hb = np.ascontiguousarray(hist_black, np.uint64)
    actions = np.ascontiguousarray(actions, np.uint8)
    rewards = np.ascontiguousarray(rewards, np.int8)
    visits = np.ascontiguousarray(visits, np.int32)
    image = np.zeros((9, 8, 8), np.float32)
    planes = np.zeros((K, 2, 8, 8), np.float32)
    tv = np.zeros(K + 1, np.float32)
    tp = np.zeros((K + 1, 65), np.float32)
    lib().mzo_replay_sample(ptr(hw), ptr(hb), ptr(actions), ptr(rewards), ptr(visits), int(length), int(pos), int(K),
                            ptr(image), ptr(planes), ptr(tv), ptr(tp))
    return image, planes, tv, tp


def replay_batch(games, picks, K):
    """utilities.py:83-99: games = list of dicts (hist_white, hist_black, actions, rewards, visits, length); picks = [(game
    index, position)] -> (image (B,9,8,8), actions (K,B,2,8,8), target_values (K+1,B), target_policies (K+1,B,65))"""
    outs = [replay_sample(g["hist_white"], g["hist_black"], g["actions"], g["rewards"], g["visits"], g["length"], pos, K)
            for g, pos in ((games[i], pos) for i, pos in picks)]
    image = np.stack([o[0] for o in outs], 0)
    acts = np.stack([o[1] for o in outs], 0).swapaxes(0, 1)
    tv = np.stack([o[2] for o in outs], 0).swapaxes(0, 1)
    tp = np.stack([o[3] for o in outs], 0).swapaxes(0, 1)
    return image, np.ascontiguousarray(acts), np.ascontiguousarray(tv), np.ascontiguousarray(tp)


# ---------------------------------------------------------------------------------------------
# arena (test.py)
# ---------------------------------------------------------------------------------------------
def controller_image(hist_white, hist_black, state_index, black_to_move):
    hw = np.ascontiguousarray(hist_white, np.uint64)
    hb = np.ascontiguousarray(hist_black, np.uint64)
    out = np.zeros((9, 8, 8), np.float32)
    lib().mzo_controller_image(ptr(hw), ptr(hb), int(state_index), int(black_to_move), ptr(out))
    return out


def softmax_sample(visits, temperature, u):
    visits = np.ascontiguousarray(visits, np.int32)
    return lib().mzo_softmax_sample(ptr(visits), float(temperature), float(u))


def arena_game_hashnet(mode_black, mode_white, S, temperature, u, max_steps=60):
    """-> (steps, actions, white_score, black_score)"""
    u = np.ascontiguousarray(u, np.float64)
    actions = np.zeros(max_steps, np.uint8)
    ws, bs = C.c_int(), C.c_int()
    n = lib().mzo_arena_game_hashnet(int(mode_black), int(mode_white), int(S), float(temperature), ptr(u), int(max_steps), ptr(actions),
                                     C.byref(ws), C.byref(bs))
    return n, actions[:n], ws.value, bs.value


# ---------------------------------------------------------------------------------------------
# batch forms (bench.py cpu_baseline): one C call per simulation over all trees, OpenMP inside
# ---------------------------------------------------------------------------------------------
class TreeBatch:
    def __init__(self, n, max_sims, **cfg):
        self.trees = [Tree(max_sims, **cfg) for _ in range(n)]
        self.n = n
        self.handles = (C.c_void_p * n)(*[t.h for t in self.trees])

    def root_init(self, to_play, legal, policy, noise=None):
        to_play = np.ascontiguousarray(to_play, np.int8)
        legal = np.ascontiguousarray(legal, np.uint64)
        policy = np.ascontiguousarray(policy, np.float32)
        noise = None if noise is None else np.ascontiguousarray(noise, np.float64)
        lib().mzo_root_init_batch(self.handles, self.n, ptr(to_play), ptr(legal), ptr(policy), ptr(noise))

    def select(self, pool, x, depth=None):
        """pool: float32 [nodes][n][512] hidden states by expansion serial; x: float32 [n][10][8][8] written in place"""
        assert pool.dtype == np.float32 and pool.flags.c_contiguous and x.dtype == np.float32 and x.flags.c_contiguous
        lib().mzo_select_batch(self.handles, self.n, ptr(pool), ptr(x), ptr(depth))

    def expand_backup(self, value, policy):
        value = np.ascontiguousarray(value, np.float32)
        policy = np.ascontiguousarray(policy, np.float32)
        lib().mzo_expand_backup_batch(self.handles, self.n, ptr(value), ptr(policy))

    def root_visits(self):
        v = np.zeros((self.n, 65), np.int32)
        lib().mzo_root_visits_batch(self.handles, self.n, ptr(v))
        return v
