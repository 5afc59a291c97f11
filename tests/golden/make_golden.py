#!/usr/bin/env python3
"""Generate the committed golden fixtures by IMPORTING the reference (this container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [env|mcts|features|game|replay|arena|all]
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py bulk [n_games]    (verdict only, ~25 min on 8 cores)
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py bulk_mcts [n] | bulk_replay [n_windows] | bulk_arena [n_games] | bulk_train [n_runs] | bulk_games [n] | checkpoint_xload    (verdicts only)

The reference lives read-only at /root/reference and never travels to the GPU box; what travels are
the small ``.npz`` files this script writes next to itself (inputs + expected outputs only).
``ray`` is not installed here, so an in-process stub module with an identity ``remote`` decorator
is registered before ``utilities``/``main`` are imported -- nothing else of the
reference is altered.  Harness-side conventions: the reference is handed a legal
list SORTED by action index wherever list order matters (root expansion, noise, select_action),
and every random number it would draw from ``np.random`` is recorded and stored by action index.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

ray = types.ModuleType("ray")


def _remote(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    return lambda f: f


ray.remote = _remote
ray.init = lambda *a, **k: None
ray.put = lambda x: x
ray.get = lambda x: x
sys.modules["ray"] = ray
sys.path.insert(0, "/root/reference")

import main as M  # noqa: E402  (the reference; sets OMP_NUM_THREADS=1 before torch)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import refspec as R  # noqa: E402

SEED = 1261


# ------------------------------------------------------------------------------------------------
# env traces
# ------------------------------------------------------------------------------------------------
def env_state(env):
    cells = env.board.reshape(-1)
    white = int(np.packbits(cells == 1, bitorder="little").view(np.uint64)[0])    # bit i <=> cell (i // 8, i % 8)
    black = int(np.packbits(cells == -1, bitorder="little").view(np.uint64)[0])
    legal = 0
    pas = 0
    for a in env.legal_actions:
        if a.index == 64:
            pas = 1
        else:
            legal |= 1 << int(a.index)
    return white, black, env.player.value, env.steps, int(env.done), legal, pas


def play_random_game(args):
    game_id, seed, voluntary_pass, n_steps = args
    env = M.Environment()
    rec = [env_state(env)]
    actions, rewards = [], []
    h = R.digest_step(0, *rec[0][:5], 0, *rec[0][5:])
    for step in range(n_steps):
        w, b, tp, st, dn, legal, pas = rec[-1]
        a = R.playout_action(seed, game_id, step, legal)
        if voluntary_pass and (R.playout_rand(seed ^ 0x5555, game_id, step) & 7) == 0:
            a = 64  # reversi.py:135-139 accepts a pass without any legality check
        _, reward, done = env.step(M.Action(a, env.player))
        s = env_state(env)
        assert s[4] == int(done)
        rec.append(s)
        actions.append(a)
        rewards.append(int(reward))
        h = R.digest_step(h, s[0], s[1], s[2], s[3], s[4], int(reward), s[5], s[6])
    return game_id, rec, actions, rewards, h


def gen_env():
    import multiprocessing as mp
    n_full, n_dig, n_vol = 128, 4096, 512
    with mp.Pool(8) as pool:
        dig = pool.map(play_random_game, [(g, SEED, 0, 60) for g in range(n_dig)], chunksize=16)
        vol = pool.map(play_random_game, [(g, SEED + 1, 1, 60) for g in range(n_vol)], chunksize=16)
    out = {"seed": np.uint64(SEED)}

    def pack_full(games, prefix):
        n = len(games)
        W = np.zeros((n, 61), np.uint64); B = np.zeros((n, 61), np.uint64)
        L = np.zeros((n, 61), np.uint64)
        TP = np.zeros((n, 61), np.int8); ST = np.zeros((n, 61), np.uint8)
        DN = np.zeros((n, 61), np.uint8); PS = np.zeros((n, 61), np.uint8)
        A = np.zeros((n, 60), np.uint8); RW = np.zeros((n, 60), np.int8)
        for i, (_, rec, actions, rewards, _) in enumerate(games):
            for t, s in enumerate(rec):
                W[i, t], B[i, t], TP[i, t], ST[i, t], DN[i, t], L[i, t], PS[i, t] = s
            A[i] = actions
            RW[i] = rewards
        for k, v in dict(white=W, black=B, legal=L, to_play=TP, steps=ST, done=DN, pass_flag=PS,
                         actions=A, rewards=RW).items():
            out[prefix + k] = v

    pack_full(dig[:n_full], "full_")
    pack_full(vol[:64], "volfull_")
    for prefix, games in (("dig_", dig), ("vol_", vol)):
        out[prefix + "actions"] = np.array([g[2] for g in games], np.uint8)
        out[prefix + "rewards_last"] = np.array([g[3][-1] for g in games], np.int8)
        out[prefix + "digest"] = np.array([g[4] for g in games], np.uint64)
        fin = [g[1][-1] for g in games]
        out[prefix + "final_white"] = np.array([f[0] for f in fin], np.uint64)
        out[prefix + "final_black"] = np.array([f[1] for f in fin], np.uint64)
        out[prefix + "final_to_play"] = np.array([f[2] for f in fin], np.int8)
        out[prefix + "final_done"] = np.array([f[4] for f in fin], np.uint8)
    # coverage the survey asks for: report it so a reader can see the cases exist
    acts = out["dig_actions"]
    n_pass = (acts == 64).sum(1)
    print("env: games with >=1 pass:", int((n_pass > 0).sum()), " >=2 passes:", int((n_pass > 1).sum()),
          " pass as 60th step (done stays 0):", int((acts[:, 59] == 64).sum()),
          " not done at end:", int((out["dig_final_done"] == 0).sum()))
    assert (acts[:, 59] == 64).sum() > 0 and (n_pass > 1).sum() > 0
    np.savez_compressed(os.path.join(HERE, "env_traces.npz"), **out)


# ------------------------------------------------------------------------------------------------
# MCTS traces
# ------------------------------------------------------------------------------------------------
from hashnet_ref import HashNetwork  # noqa: E402  (shared with the compat tests)


class RecordingNetwork:
    """wraps the reference's real Network and records every (value, policy) it returns."""

    def __init__(self, net):
        self.net = net
        self.values, self.policies = [], []

    def initial_inference(self, image):
        with torch.no_grad():
            return self.net.initial_inference(image)

    def recurrent_inference(self, hidden, action):
        with torch.no_grad():
            out = self.net.recurrent_inference(hidden, action)
        self.values.append(out[0].numpy().reshape(()).copy())
        self.policies.append(out[1].numpy().reshape(65).copy())
        return out


def reach_position(game_id, n_moves, seed=SEED):
    """a reference Game advanced n_moves by the build RNG (same stream as the env fixtures)."""
    game = M.Game()
    for step in range(n_moves):
        s = env_state(game.environment)
        a = R.playout_action(seed, game_id, step, s[5])
        game.apply(M.Action(a, game.to_play()))
    return game


def run_reference_search(game, net, S, noise_seed, use_noise=True, known_bounds=(-1, 1),
                         discount=1.0):
    """one reference search (main.py:75-88 composition) with instrumentation; returns a dict."""
    M.config.num_simulations = S
    M.config.known_bounds = known_bounds
    M.config.discount = discount
    paths, cur = [], []
    orig_select = M.select_child

    def rec_select(node, mm):
        a, child = orig_select(node, mm)
        cur.append(a)
        return a, child

    orig_backprop = M.backpropagate

    def rec_backprop(search_path, value, player, discount, mm):
        paths.append(list(cur))
        cur.clear()
        return orig_backprop(search_path, value, player, discount, mm)

    M.select_child, M.backpropagate = rec_select, rec_backprop
    try:
        root = M.Node(0, game.to_play())
        image = torch.from_numpy(game.make_image(game.environment.steps))
        out0 = net.initial_inference(image)
        legal = sorted(game.legal_actions(), key=lambda a: a.index)  # H3: sorted legal list
        M.expand_node(root, legal, out0)
        noise = np.zeros(65, np.float64)
        if use_noise:
            np.random.seed(noise_seed)
            st = np.random.get_state()
            M.add_exploration_noise(root)
            rs = np.random.RandomState()
            rs.set_state(st)
            drawn = rs.dirichlet([M.config.root_dirichlet_alpha] * len(legal))
            for a, n in zip(legal, drawn):
                noise[a.index] = n
        M.run_mcts(root, game.action_history, net)
    finally:
        M.select_child, M.backpropagate = orig_select, orig_backprop
    visits = np.full(65, -1, np.int32)
    wsum = np.zeros(65, np.float32)
    prior = np.zeros(65, np.float64)
    for a, c in root.children.items():
        visits[a] = c.visit_count
        wsum[a] = float(c.value_sum) if c.visit_count else 0.0
        prior[a] = c.prior
    s = env_state(game.environment)
    # select_action (main.py:128-140): greedy = first max in children (= sorted legal) order;
    # sampled = np.random.choice with the uniform u replayed from the same generator state.
    greedy = M.softmax_sample([(c.visit_count, a) for a, c in root.children.items()], 0)
    np.random.seed(noise_seed + 77)
    st = np.random.get_state()
    sampled = M.softmax_sample([(c.visit_count, a) for a, c in root.children.items()], 1)
    rs = np.random.RandomState()
    rs.set_state(st)
    u = rs.random_sample()
    game2 = M.Game()
    game2.store_search_statistics(root)
    return dict(white=s[0], black=s[1], to_play=s[2], steps=s[3], legal=s[5], pass_flag=s[6],
                root_value0=np.float32(out0[0].numpy().reshape(())), root_policy=out0[1].numpy().reshape(65),
                noise=noise, use_noise=int(use_noise), S=S, paths=paths, visits=visits, wsum=wsum,
                prior=prior, root_n=root.visit_count,
                root_value=np.float32(root.value().numpy().reshape(())) if root.visit_count else np.float32(0),
                greedy=int(greedy), sampled=int(sampled), u=u,
                child_visits=np.array(game2.child_visits[0], np.float64),
                image=image.numpy().copy())


def gen_mcts():
    cases = []
    meta = []
    # (mode, S, game_id, n_moves, noise, bounds, discount)
    plan = []
    gid = 0
    for S in (50, 60, 100):
        for n_moves in (0, 1, 7, 20, 33, 45, 58):
            plan.append(("hash", S, gid, n_moves, True, (-1, 1), 1.0)); gid += 1
    for mode in ("uniform", "sat", "const"):
        for n_moves in (0, 12, 40):
            plan.append((mode, 60, gid, n_moves, mode != "uniform", (-1, 1), 1.0)); gid += 1
    plan.append(("uniform", 200, gid, 0, False, (-1, 1), 1.0)); gid += 1   # deep pure tie-break tree
    plan.append(("hash", 60, gid, 5, False, (-1, 1), 1.0)); gid += 1       # no noise
    plan.append(("hash", 60, gid, 9, True, None, 1.0)); gid += 1           # known_bounds = None
    plan.append(("sat", 60, gid, 9, True, None, 1.0)); gid += 1
    plan.append(("hash", 60, gid, 11, True, (-1, 1), 0.5)); gid += 1       # discount knob
    plan.append(("hash", 60, gid, 11, True, None, 0.997)); gid += 1
    plan.append(("hash", 60, gid, 11, True, (-0.5, 0.5), 1.0)); gid += 1   # bounds that DO get replaced
    # a pass-only root: scan the env fixture stream for a game whose position has no legal move
    for g in range(4096):
        game = M.Game()
        found = None
        for step in range(59):
            s = env_state(game.environment)
            if s[6]:
                found = step
                break
            game.apply(M.Action(R.playout_action(SEED, g, step, s[5]), game.to_play()))
        if found is not None and found < 57:
            plan.append(("hash", 60, g, found, True, (-1, 1), 1.0))
            break
    for (mode, S, g, n_moves, noise, bounds, disc) in plan:
        game = reach_position(g, n_moves)
        net = HashNetwork(mode)
        res = run_reference_search(game, net, S, 1000 + g, noise, bounds, disc)
        res.update(mode=R.MODES[mode], kind=0, bounds_none=int(bounds is None),
                   bound_lo=(bounds[0] if bounds else 0.0), bound_hi=(bounds[1] if bounds else 0.0),
                   discount=disc, net_values=np.zeros(0, np.float32), net_policies=np.zeros((0, 65), np.float32))
        cases.append(res)
        print("mcts hash case", mode, S, g, n_moves, "maxdepth", max(len(p) for p in res["paths"]),
              "visits", res["visits"][res["visits"] >= 0].tolist())
    # real-network searches: outputs recorded, replayed in the tests (H6)
    torch.manual_seed(1261)
    real = M.Network()
    real.eval()
    for i, (S, n_moves) in enumerate([(50, 0), (60, 3), (60, 18), (100, 30), (60, 44), (60, 57)]):
        game = reach_position(2000 + i, n_moves)
        net = RecordingNetwork(real)
        res = run_reference_search(game, net, S, 5000 + i, True)
        res.update(mode=-1, kind=1, bounds_none=0, bound_lo=-1.0, bound_hi=1.0, discount=1.0,
                   net_values=np.array(net.values, np.float32), net_policies=np.array(net.policies, np.float32))
        cases.append(res)
        print("mcts real case", S, n_moves, "maxdepth", max(len(p) for p in res["paths"]),
              "visits", res["visits"][res["visits"] >= 0].tolist())
    M.config.num_simulations, M.config.known_bounds, M.config.discount = 60, (-1, 1), 1.0
    out = {}
    scal = ["white", "black", "to_play", "steps", "legal", "pass_flag", "root_value0", "use_noise", "S",
            "root_n", "root_value", "greedy", "sampled", "u", "mode", "kind", "bounds_none", "bound_lo",
            "bound_hi", "discount"]
    dt = dict(white=np.uint64, black=np.uint64, legal=np.uint64, to_play=np.int8, root_value0=np.float32,
              root_value=np.float32, u=np.float64, bound_lo=np.float64, bound_hi=np.float64,
              discount=np.float64)
    for k in scal:
        out[k] = np.array([c[k] for c in cases], dt.get(k, np.int32))
    for k in ("root_policy", "noise", "visits", "wsum", "prior", "child_visits", "image"):
        out[k] = np.stack([c[k] for c in cases])
    # ragged: per-sim paths; per-case net outputs
    path_flat, path_off, sim_off = [], [0], [0]
    for c in cases:
        for p in c["paths"]:
            path_flat.extend(p)
            path_off.append(len(path_flat))
        sim_off.append(len(path_off) - 1)
    out["path_actions"] = np.array(path_flat, np.uint8)
    out["path_off"] = np.array(path_off, np.int64)
    out["sim_off"] = np.array(sim_off, np.int64)
    out["net_values"] = np.concatenate([c["net_values"] for c in cases])
    out["net_policies"] = np.concatenate([c["net_policies"] for c in cases])
    out["net_off"] = np.cumsum([0] + [len(c["net_values"]) for c in cases]).astype(np.int64)
    np.savez_compressed(os.path.join(HERE, "mcts_traces.npz"), **out)


# ------------------------------------------------------------------------------------------------
# feature traces: make_image / Action.encode / make_target inputs
# ------------------------------------------------------------------------------------------------
def gen_features():
    out = {}
    enc = []
    for idx in (0, 7, 19, 63, 64):
        for pl in (M.Player.WHITE, M.Player.BLACK):
            enc.append((idx, pl.value, M.Action(idx, pl).encode().astype(np.uint8)))
    out["enc_index"] = np.array([e[0] for e in enc], np.int32)
    out["enc_player"] = np.array([e[1] for e in enc], np.int8)
    out["enc_planes"] = np.stack([e[2] for e in enc])
    # make_image along whole games (incl. ones with passes): store states + images at many indices
    gids = [0, 1, 2, 3]
    acts = np.load(os.path.join(HERE, "env_traces.npz"))["dig_actions"]
    with_pass = [int(g) for g in np.nonzero((acts[:, :50] == 64).any(1))[0][:4]]
    gids += with_pass
    imgs, meta, hw, hb = [], [], [], []
    for g in gids:
        game = reach_position(g, 60)
        W = np.zeros(61, np.uint64); B = np.zeros(61, np.uint64)
        for t, obs in enumerate(game.state_history):
            w = b = 0
            for i in range(64):
                if obs[0].reshape(-1)[i]:
                    w |= 1 << i
                if obs[1].reshape(-1)[i]:
                    b |= 1 << i
            W[t], B[t] = w, b
        hw.append(W); hb.append(B)
        for idx in list(range(0, 8)) + [20, 35, 59, 60]:
            imgs.append(game.make_image(idx).astype(np.uint8))
            meta.append((len(hw) - 1, idx))
    out["img_game"] = np.array([m[0] for m in meta], np.int32)
    out["img_index"] = np.array([m[1] for m in meta], np.int32)
    out["img"] = np.stack(imgs)
    out["hist_white"] = np.stack(hw)
    out["hist_black"] = np.stack(hb)
    out["game_ids"] = np.array(gids, np.int32)
    np.savez_compressed(os.path.join(HERE, "features.npz"), **out)


# ------------------------------------------------------------------------------------------------
# whole self-play games (main.play_game) with HashNet: the end-to-end composition
# ------------------------------------------------------------------------------------------------
def gen_game():
    """reference main.play_game (main.py:69-92) on HashNet, with np.random draws recorded.

    The only harness intervention: Game.legal_actions returns the env's list sorted by index (H3).
    """
    orig_legal = M.Game.legal_actions
    M.Game.legal_actions = lambda self: sorted(self.environment.legal_actions, key=lambda a: a.index)
    orig_dir, orig_choice = np.random.dirichlet, np.random.choice
    out = {}
    try:
        recs = []
        for gi, (S, seed) in enumerate([(16, 11), (16, 12), (30, 13)]):
            M.config.num_simulations = S
            np.random.seed(seed)
            noises, us = [], []

            def rec_dirichlet(alpha, *a, **k):
                d = orig_dir(alpha, *a, **k)
                noises.append(np.array(d))
                return d

            def rec_choice(actions, p=None):
                st = np.random.get_state()
                r = orig_choice(actions, p=p)
                rs = np.random.RandomState()
                rs.set_state(st)
                us.append(rs.random_sample())
                return r

            np.random.dirichlet, np.random.choice = rec_dirichlet, rec_choice
            game = M.play_game(HashNetwork("hash"))
            np.random.dirichlet, np.random.choice = orig_dir, orig_choice
            n = len(game.action_history)
            # noise by action index: replay legal sets from the stored states
            g2 = M.Game()
            noise = np.zeros((n, 65), np.float64)
            for t in range(n):
                legal = sorted(a.index for a in g2.environment.legal_actions)
                noise[t, legal] = noises[t]
                g2.apply(M.Action(game.action_history[t].index, g2.to_play()))
            u = np.zeros(n, np.float64)
            u[:len(us)] = us  # moves >= 20 are greedy and draw nothing
            tgt_idx = [0, 1, 30, 56, 57, 58, 59]
            tv, tp = [], []
            for K in (3, 5):
                for si in tgt_idx:
                    v_, p_ = game.make_target(si, K)
                    tv.append(np.array(v_ + [0] * (6 - len(v_)), np.float64))
                    tp.append(np.array([list(x) for x in p_] + [[0] * 65] * (6 - len(p_)), np.float64))
            recs.append(dict(S=S, target_index=np.array(tgt_idx, np.int32), target_values=np.stack(tv), target_policies=np.stack(tp),
                             actions=np.array([a.index for a in game.action_history], np.uint8),
                             rewards=np.array(game.rewards, np.int8), noise=noise, u=u,
                             child_visits=np.array(game.child_visits, np.float64),
                             root_values=np.array([float(v) for v in game.root_values], np.float32),
                             done=int(game.terminal())))
            print("game", gi, "S", S, "moves", n, "done", game.terminal(), "last reward", game.rewards[-1])
        for k in recs[0]:
            if k in ("S", "done"):
                out[k] = np.array([r[k] for r in recs], np.int32)
            else:
                for i, r in enumerate(recs):
                    out["%s_%d" % (k, i)] = r[k]
    finally:
        M.Game.legal_actions = orig_legal
        np.random.dirichlet, np.random.choice = orig_dir, orig_choice
        M.config.num_simulations = 60
    np.savez_compressed(os.path.join(HERE, "selfplay_games.npz"), **out)



# ------------------------------------------------------------------------------------------------
# replay buffer (utilities.py:46-109) + update_weights (main.py:230-272)
# ------------------------------------------------------------------------------------------------
def _obs_bits(obs):
    w = b = 0
    fw, fb = obs[0].reshape(-1), obs[1].reshape(-1)
    for i in range(64):
        if fw[i]:
            w |= 1 << i
        if fb[i]:
            b |= 1 << i
    return w, b


def gen_replay():
    """reference ReplayBuffer filled with main.play_game games (HashNet), FIFO eviction included, then
    sample_batch under a recorded `random` stream; then reference update_weights on those batches."""
    import random
    import utilities as U
    orig_legal = M.Game.legal_actions
    M.Game.legal_actions = lambda self: sorted(self.environment.legal_actions, key=lambda a: a.index)
    orig_store = M.Game.store_search_statistics
    raw = {}

    def rec_store(self, root):
        v = np.full(65, -1, np.int32)
        for a, c in root.children.items():
            v[a] = c.visit_count
        raw.setdefault(id(self), []).append(v)
        return orig_store(self, root)

    M.Game.store_search_statistics = rec_store
    out = {}
    try:
        U.config.window_size, U.config.batch_size = 6, 4
        rb = U.ReplayBuffer()
        games = []
        plan = [(6, 60), (6, 60), (6, 9), (8, 60), (6, 60), (6, 14), (8, 60), (6, 60)]  # (S, max_moves): 8 games into a window of 6
        for gi, (S, mm) in enumerate(plan):
            M.config.num_simulations, M.config.max_moves = S, mm
            np.random.seed(300 + gi)
            g = M.play_game(HashNetwork("hash"))
            games.append(g)
            rb.save_game(g)
            assert rb.ready() == (gi + 1 >= 4)
        M.config.num_simulations, M.config.max_moves = 60, 60
        assert rb.buffer_size == 6 and rb.buffer[0] is games[2]
        out["count"] = np.int64(rb.count())
        assert rb.count() == 0
        n = len(games)
        L = np.array([len(g.action_history) for g in games], np.int32)
        HW = np.zeros((n, 61), np.uint64); HB = np.zeros((n, 61), np.uint64)
        A = np.zeros((n, 60), np.uint8); RW = np.zeros((n, 60), np.int8)
        V = np.full((n, 60, 65), -1, np.int32); RV = np.zeros((n, 60), np.float32)
        for i, g in enumerate(games):
            for t, obs in enumerate(g.state_history):
                HW[i, t], HB[i, t] = _obs_bits(obs)
            A[i, :L[i]] = [a.index for a in g.action_history]
            # the mover of step t is BLACK for even t (the build derives it from parity)
            assert all((a.player is M.Player.BLACK) == (t % 2 == 0) for t, a in enumerate(g.action_history))
            RW[i, :L[i]] = g.rewards
            V[i, :L[i]] = np.stack(raw[id(g)])
            RV[i, :L[i]] = [float(v) for v in g.root_values]
        out.update(length=L, hist_white=HW, hist_black=HB, actions=A, rewards=RW, visits=V, root_values=RV,
                   window_size=np.int32(6), batch_size=np.int32(4), evicted=np.int32(2))
        print("replay: lengths", L.tolist(), "last rewards", [int(g.rewards[-1]) for g in games])
        # sample_batch with the `random` draws recorded: game_idxes (list positions, 0 = oldest) and positions
        orig_sample, orig_randint = random.sample, random.randint
        batches = []
        for bi, (K, seed) in enumerate([(3, 1261), (3, 7), (5, 8), (5, 9), (1, 10)]):
            U.config.num_unroll_steps = K  # sample_position reads the config, not the argument (utilities.py:104)
            random.seed(seed)
            idxes, poss = [], []

            def rec_sample(pop, k):
                r = orig_sample(pop, k)
                idxes.extend(r)
                return r

            def rec_randint(a, b):
                r = orig_randint(a, b)
                poss.append(r)
                return r

            random.sample, random.randint = rec_sample, rec_randint
            try:
                img, act, tv, tp = rb.sample_batch(K)
            finally:
                random.sample, random.randint = orig_sample, orig_randint
            pre = "b%d_" % bi
            out[pre + "K"] = np.int32(K); out[pre + "seed"] = np.int32(seed)
            out[pre + "game_idx"] = np.array(idxes, np.int32); out[pre + "pos"] = np.array(poss, np.int32)
            # what update_weights feeds the network: torch.FloatTensor(...) of each (main.py:236)
            out[pre + "image"] = torch.FloatTensor(img).numpy().astype(np.uint8)
            out[pre + "actions"] = torch.FloatTensor(act).numpy().astype(np.uint8)
            out[pre + "target_values"] = torch.FloatTensor(tv).numpy()
            out[pre + "target_policies"] = torch.FloatTensor(tp).numpy()
            batches.append((img, act, tv, tp))
            print("replay batch", bi, "K", K, "idx", idxes, "pos", poss, "tv", tv.T.tolist())
        out["n_batches"] = np.int32(len(batches))
        U.config.num_unroll_steps = 3
        U.config.window_size, U.config.batch_size = 512, 512
        # ---- update_weights (main.py:230-272) with train_network's optimizer (main.py:206) on batches 0, 1 (K=3), 2 (K=5)
        torch.manual_seed(1261)
        net = M.Network()
        net.train()
        opt = M.optim.SGD(net.parameters(), lr=0.001, weight_decay=M.config.lr_decay_rate, momentum=M.config.momentum)
        names = [k for k, _ in net.named_parameters()]
        out["param_names"] = np.array(names)
        out["param_shapes"] = np.array([str(tuple(p.shape)) for _, p in net.named_parameters()])
        gs, ws = [], []
        orig_step = opt.step

        def rec_step(*a, **k):
            gs.append(np.array([[float(p.grad.double().sum()), float(p.grad.double().abs().sum())] for p in net.parameters()]))
            r = orig_step(*a, **k)
            ws.append(np.array([[float(p.detach().double().sum()), float(p.detach().double().abs().sum())] for p in net.parameters()]))
            return r

        opt.step = rec_step
        import io, contextlib
        logs = []
        for b in (0, 1, 2, 0):
            net.steps = 0  # force the loss print (main.py:270-271)
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                M.update_weights(opt, net, batches[b])
            logs.append(buf.getvalue().strip())
        print("train logs:", logs)
        out["train_batches"] = np.array([0, 1, 2, 0], np.int32)
        out["train_grad_sums"] = np.stack(gs); out["train_weight_sums"] = np.stack(ws)
        out["train_logs"] = np.array(logs)
        sd = net.state_dict()
        for k in ("prediction.value_head.3.weight", "prediction.value_head.3.bias", "prediction.policy_head.5.bias",
                  "dynamics.conv.5.weight"):
            out["final_" + k] = sd[k].numpy().copy()
        out["state_dict_keys"] = np.array(list(sd.keys()))
        out["state_dict_shapes"] = np.array([str(tuple(v.shape)) for v in sd.values()])
    finally:
        M.Game.legal_actions, M.Game.store_search_statistics = orig_legal, orig_store
        M.config.num_simulations, M.config.max_moves = 60, 60
    np.savez_compressed(os.path.join(HERE, "replay.npz"), **out)


# ------------------------------------------------------------------------------------------------
# arena (test.py): Controller observation, 100-simulation search without noise, t = 0.25 sampling
# ------------------------------------------------------------------------------------------------
def gen_arena():
    import test as RT  # /root/reference/test.py (first on sys.path)
    assert RT.__file__.startswith("/root/reference")
    orig_choice = np.random.choice
    out = {}
    games = []
    try:
        for gi, (mb, mw, seed) in enumerate([("hash", "sat", 21), ("sat", "hash", 22), ("hash", "hash", 23)]):
            np.random.seed(seed)
            us = []

            def rec_choice(actions, p=None):
                st = np.random.get_state()
                r = orig_choice(actions, p=p)
                rs = np.random.RandomState()
                rs.set_state(st)
                us.append(rs.random_sample())
                return r

            np.random.choice = rec_choice
            env = RT.Environment()
            black_c = RT.Controller(RT.Player.BLACK, HashNetwork(mb))
            white_c = RT.Controller(RT.Player.WHITE, HashNetwork(mw))
            acts, imgs = [], []
            # test.py:51-63 with two harness interventions: legal list sorted by index (H3), and the loop
            # bounded at 60 steps (the reference's `while not env.done` never ends when step 60 is a pass)
            while not env.done and env.steps < 60:
                obs = env.observe()
                legal = sorted(env.legal_actions, key=lambda a: a.index)
                mover = env.player
                a_b = black_c.step(obs, env.player, legal)
                a_w = white_c.step(obs, env.player, legal)
                img = np.zeros((9, 8, 8), np.float32)
                img[:-1] = np.concatenate(black_c.state_history)
                if mover is RT.Player.BLACK:
                    img[-1] = 1
                imgs.append(img.astype(np.uint8))
                a = a_b if a_b else a_w
                assert (a_b is None) != (a_w is None)
                env.step(a)
                acts.append(a.index)
            np.random.choice = orig_choice
            ws, bs = int(np.sum(env.board == 1)), int(np.sum(env.board == -1))
            games.append(dict(actions=np.array(acts, np.uint8), u=np.array(us, np.float64), images=np.stack(imgs),
                              white_score=ws, black_score=bs, done=int(env.done)))
            print("arena game", gi, mb, "vs", mw, "steps", len(acts), "done", env.done, "white", ws, "black", bs)
            out["mode_black_%d" % gi] = np.int32(R.MODES[mb]); out["mode_white_%d" % gi] = np.int32(R.MODES[mw])
            for k, v in games[-1].items():
                out["%s_%d" % (k, gi)] = np.asarray(v)
        out["n_games"] = np.int32(len(games))
        # softmax_sample (test.py:228-243) on its own: temperatures 0 / 0.25 / 0.5 / 1 over random visit vectors
        rng = np.random.RandomState(5)
        V, T, U, A = [], [], [], []
        for i in range(200):
            k = rng.randint(1, 12)
            acts = sorted(rng.choice(65, k, replace=False).tolist())
            vis = rng.randint(0, 101, k)
            vis[rng.randint(k)] += 1
            t = [0, 0.25, 0.5, 1][i % 4]
            np.random.seed(1000 + i)
            st = np.random.get_state()
            a = RT.softmax_sample(list(zip(vis.tolist(), acts)), t)
            rs = np.random.RandomState(); rs.set_state(st)
            v = np.full(65, -1, np.int32); v[acts] = vis
            V.append(v); T.append(t); U.append(rs.random_sample()); A.append(int(a))
        out.update(ss_visits=np.stack(V), ss_t=np.array(T), ss_u=np.array(U), ss_action=np.array(A, np.int32))
    finally:
        np.random.choice = orig_choice
    np.savez_compressed(os.path.join(HERE, "arena.npz"), **out)


# ------------------------------------------------------------------------------------------------
# bulk verdict: >= 100k reference games against the oracle's digests, verdict only
# ------------------------------------------------------------------------------------------------
def _bulk_chunk(args):
    """one chunk of reference games -> (digests, action log, final done flags), nothing else kept"""
    g0, n, seed, vol = args
    dig = np.zeros(n, np.uint64)
    acts = np.zeros((n, 60), np.uint8)
    done = np.zeros(n, np.uint8)
    last = np.zeros(n, np.int8)
    for i in range(n):
        _, rec, actions, rewards, h = play_random_game((g0 + i, seed, vol, 60))
        dig[i], acts[i], done[i], last[i] = h, actions, rec[-1][4], rewards[-1]
    return g0, seed, vol, dig, acts, done, last


def gen_bulk(n_games=131072):
    """Every game is played by the imported reference (reversi.py Environment.step) under the build's counter RNG and
    folded into the 64-bit digest of all 61 states (every field: boards, to_play, steps, done, reward, legal set, pass
    flag); the oracle (oracle/mz_oracle.c) plays the same game ids and the digests, action logs, done flags and last
    rewards are compared in-process.  Only the verdict is written: tests/golden/bulk_env_verdict.json."""
    import json
    import multiprocessing as mp
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))   # repo root: the oracle package
    from oracle import mz_oracle as O
    t0 = time.time()
    game0 = 1 << 20                      # ids disjoint from the committed fixtures (games 0..4095)
    n_vol = n_games // 8                 # an eighth of the games also pass voluntarily (reversi.py:135-139)
    n_reg = n_games - n_vol
    chunk = 256
    jobs = [(game0 + g, min(chunk, n_reg - g), SEED, 0) for g in range(0, n_reg, chunk)]
    jobs += [(game0 + g, min(chunk, n_vol - g), SEED + 1, 1) for g in range(0, n_vol, chunk)]
    stats = dict(games=0, steps=0, mismatched_digests=0, mismatched_actions=0, mismatched_done=0, mismatched_last_reward=0,
                 games_with_pass=0, games_with_2plus_passes=0, pass_as_60th_step_done_stays_0=0, not_done_at_end=0,
                 voluntary_pass_games=0)
    with mp.Pool(8) as pool:
        for k, (g0, seed, v, dig, acts, done, last) in enumerate(pool.imap_unordered(_bulk_chunk, jobs)):
            e, odig, oacts, olast = O.random_playout(seed, g0, len(dig), 60, v, want_actions=True)
            stats["games"] += len(dig)
            stats["steps"] += 60 * len(dig)
            stats["mismatched_digests"] += int((odig != dig).sum())
            stats["mismatched_actions"] += int((oacts != acts).any(1).sum())
            stats["mismatched_done"] += int((e.done != done).sum())
            stats["mismatched_last_reward"] += int((olast != last).sum())
            npass = (acts == 64).sum(1)
            stats["games_with_pass"] += int((npass > 0).sum())
            stats["games_with_2plus_passes"] += int((npass > 1).sum())
            stats["pass_as_60th_step_done_stays_0"] += int(((acts[:, 59] == 64) & (done == 0)).sum())
            stats["not_done_at_end"] += int((done == 0).sum())
            stats["voluntary_pass_games"] += len(dig) if v else 0
            if k % 32 == 0:
                print("bulk: %d games, %.0f s, mismatches so far %d" % (stats["games"], time.time() - t0,
                      stats["mismatched_digests"] + stats["mismatched_actions"]), flush=True)
    stats.update(first_game_id=game0, seed=SEED, seconds=round(time.time() - t0, 1), processes=8,
                 reference="/root/reference/reversi.py Environment.step imported in the build container",
                 compared="64-bit digest of all 61 states per game + action log + final done + last reward vs oracle/mz_oracle.c")
    with open(os.path.join(HERE, "bulk_env_verdict.json"), "w") as f:
        json.dump(stats, f, indent=1)
    print(json.dumps(stats))
    assert stats["mismatched_digests"] == stats["mismatched_actions"] == stats["mismatched_done"] == 0


# ------------------------------------------------------------------------------------------------
# bulk verdict for the SEARCH (main.py:99-196): >= 10k reference searches against the oracle, verdict only
# ------------------------------------------------------------------------------------------------
BULK_S = (50, 60, 50, 100, 60, 50, 60, 200, 50, 60, 100, 60)           # S by job % 12 (mean 75)
BULK_MODES = ("hash", "sat", "const", "uniform", "hash", "hash")        # by (job // 12) % 6
BULK_BOUNDS = ((-1, 1), None, (-0.25, 0.3), (-1, 1), (-0.1, 0.7), (-1, 1), None)   # by (job // 5) % 7; ints as config.py has them
BULK_DISCOUNT = (1.0, 0.9, 1.0, 0.997, 1.0)                             # by (job // 3) % 5


def _bulk_mcts_plan(job):
    S = BULK_S[job % 12]
    mode = BULK_MODES[(job // 12) % 6]
    bounds = BULK_BOUNDS[(job // 5) % 7]
    disc = BULK_DISCOUNT[(job // 3) % 5]
    noise = (job % 4) != 3
    n_moves = (job * 7) % 60                 # 7 is coprime to 60: every depth 0..59 equally often
    want_pass_root = (job % 16) == 9         # a sixteenth of the jobs look for a pass-only root in their game
    return S, mode, bounds, disc, noise, n_moves, want_pass_root


def _bulk_mcts_chunk(args):
    """a chunk of jobs: reference search (instrumented main.run_mcts) and oracle search on the same root, compared here"""
    job0, n = args
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import mz_oracle as O
    st = dict(searches=0, simulations=0, mism_visits=0, mism_w_bits=0, mism_prior_bits=0, mism_root_value_bits=0, mism_root_n=0,
              mism_depth=0, mism_leaf_action=0, mism_paths=0, paths_checked_searches=0, max_depth=0, pass_only_roots=0,
              by_S={}, by_mode={}, by_bounds={}, by_discount={}, by_noise={}, by_moves_decile={})
    game0 = 1 << 21
    for job in range(job0, job0 + n):
        S, mode, bounds, disc, use_noise, n_moves, want_pass = _bulk_mcts_plan(job)
        g = game0 + job
        if want_pass:   # first position of this game where the mover has no move (reversi.py:124-125), if any before step 57
            game = M.Game()
            for step in range(57):
                s = env_state(game.environment)
                if s[6]:
                    break
                game.apply(M.Action(R.playout_action(SEED, g, step, s[5]), game.to_play()))
        else:
            game = reach_position(g, n_moves)
        net = HashNetwork(mode)
        res = run_reference_search(game, net, S, 70000 + job, use_noise, bounds, disc)
        t = O.Tree(S, discount=disc, known_bounds=bounds)
        noise = res["noise"] if use_noise else None
        h0 = R.hashnet_image_hash(res["image"])
        _, depth, leaf_a = t.search_hashnet(res["to_play"], res["legal"], res["pass_flag"], noise, h0, R.MODES[mode], S)
        visits, wsum, prior, rn, rv = t.root_stats()
        ref_depth = np.array([len(p) for p in res["paths"]], np.int32)
        ref_leaf = np.array([p[-1] for p in res["paths"]], np.uint8)
        st["searches"] += 1
        st["simulations"] += S
        st["mism_visits"] += int((visits != res["visits"]).any())
        st["mism_w_bits"] += int((wsum.view(np.uint32) != res["wsum"].view(np.uint32)).any())
        st["mism_prior_bits"] += int((prior.view(np.uint64) != res["prior"].view(np.uint64)).any())
        st["mism_root_value_bits"] += int(np.float32(rv).view(np.uint32) != res["root_value"].view(np.uint32))
        st["mism_root_n"] += int(rn != res["root_n"])
        st["mism_depth"] += int((depth != ref_depth).any())
        st["mism_leaf_action"] += int((leaf_a != ref_leaf).any())
        if job % 8 == 0:   # every eighth search also replays the oracle step by step and compares WHOLE paths per simulation
            t2 = O.Tree(S, discount=disc, known_bounds=bounds)
            t2.root_init(res["to_play"], res["legal"], res["pass_flag"], res["root_policy"], noise)
            node_h = {0: h0}
            bad = 0
            for s_ in range(S):
                d, pid, a, plane, path = t2.select()
                want = res["paths"][s_]
                bad |= int(len(path) != len(want) or (path != np.array(want)).any())
                hh = R.hashnet_step(node_h[pid], R.action_code(a, plane))
                node_h[s_ + 1] = hh
                v, p = R.hashnet_outputs(hh, mode)
                t2.expand_backup(v, p)
            st["mism_paths"] += bad
            st["paths_checked_searches"] += 1
        st["max_depth"] = max(st["max_depth"], int(ref_depth.max()))
        st["pass_only_roots"] += int(res["pass_flag"])
        for key, val in (("by_S", S), ("by_mode", mode), ("by_bounds", str(bounds)), ("by_discount", disc),
                         ("by_noise", int(use_noise)), ("by_moves_decile", int(res["steps"]) // 10)):
            st[key][str(val)] = st[key].get(str(val), 0) + 1
    M.config.num_simulations, M.config.known_bounds, M.config.discount = 60, (-1, 1), 1.0
    return st


def gen_bulk_mcts(n_searches=10240):
    """Every search is run by the imported reference (main.expand_node / add_exploration_noise / run_mcts on HashNet, the
    harness of run_reference_search) and by the oracle (mzo_search_hashnet) on the same root, noise and knobs; compared
    in-process: visit counts, child value_sum bits, prior bits (float64), root value bits, root visit count, and per
    simulation the path depth and the leaf action (whole paths on every eighth search).  Only the verdict is written:
    tests/golden/bulk_mcts_verdict.json."""
    import json
    import multiprocessing as mp
    import time
    t0 = time.time()
    chunk = 16
    jobs = [(j, min(chunk, n_searches - j)) for j in range(0, n_searches, chunk)]
    tot = None
    with mp.Pool(8) as pool:
        for k, st in enumerate(pool.imap_unordered(_bulk_mcts_chunk, jobs)):
            if tot is None:
                tot = st
            else:
                for key, v in st.items():
                    if isinstance(v, dict):
                        for kk, vv in v.items():
                            tot[key][kk] = tot[key].get(kk, 0) + vv
                    elif key == "max_depth":
                        tot[key] = max(tot[key], v)
                    else:
                        tot[key] += v
            if k % 16 == 0:
                print("bulk_mcts: %d searches, %.0f s, mismatching searches so far %d" % (
                    tot["searches"], time.time() - t0, sum(v for kk, v in tot.items() if kk.startswith("mism_"))), flush=True)
    for key in list(tot):
        if isinstance(tot[key], dict):
            tot[key] = dict(sorted(tot[key].items()))
    tot["mismatches"] = sum(v for kk, v in tot.items() if kk.startswith("mism_"))
    tot.update(first_game_id=1 << 21, seed=SEED, seconds=round(time.time() - t0, 1), processes=8,
               reference="/root/reference/main.py expand_node + add_exploration_noise + run_mcts (select_child, ucb_score, "
                         "backpropagate, MinMaxStats) imported in the build container, HashNet as the network",
               compared="per search: visits[65], value_sum bits (f32), prior bits (f64), root value bits, root N, per-simulation depth "
                        "and leaf action (whole paths on every 8th search) vs oracle/mz_oracle.c mzo_search_hashnet")
    with open(os.path.join(HERE, "bulk_mcts_verdict.json"), "w") as f:
        json.dump(tot, f, indent=1)
    print(json.dumps(tot))
    assert tot["mismatches"] == 0


# ------------------------------------------------------------------------------------------------
# bulk verdict for the REPLAY BUFFER (utilities.py:46-109, reversi.py:301-341): reference sample_batch vs the oracle, verdict only
# ------------------------------------------------------------------------------------------------
class _FakeChild:
    def __init__(self, n):
        self.visit_count = n


class _FakeRoot:
    """what Game.store_search_statistics reads of a searched root (reversi.py:292-299): children's visit counts and value()"""

    def __init__(self, children, value):
        self.children, self._v = children, value

    def value(self):
        return self._v


def _bulk_replay_chunk(args):
    """one window: `n_games` reference Games of random legal moves (voluntary passes included) with random visit counts and
    root values, pushed through a reference ReplayBuffer smaller than n_games (FIFO eviction), then reference sample_batch
    calls for K = 1, 3, 5 under recorded `random` draws, each compared bit for bit with the oracle's replay_batch"""
    import random
    import utilities as U
    from oracle import mz_oracle as O
    job, n_games, window, batch, n_batches = args
    rng = np.random.default_rng(977 + job)
    U.config.window_size, U.config.batch_size = window, batch
    rb = U.ReplayBuffer()
    recs = []
    for gi in range(n_games):
        max_moves = int(rng.choice([60, 60, 60, 9, 14, 23, 37, 59]))
        S = int(rng.choice([1, 6, 50, 100, 255, 800]))
        g = M.Game()
        visits = np.full((60, 65), -1, np.int32)
        while not g.terminal() and len(g.action_history) < max_moves:
            legal = sorted(g.legal_actions(), key=lambda a: a.index)
            counts = rng.multinomial(S, rng.dirichlet(np.ones(len(legal)) * 0.5))      # sums to S; zeros allowed (an unvisited child)
            t = len(g.action_history)
            for a, c in zip(legal, counts):
                visits[t, a.index] = int(c)
            g.store_search_statistics(_FakeRoot({a.index: _FakeChild(int(c)) for a, c in zip(legal, counts)}, float(rng.uniform(-1, 1))))   # children are keyed by action index (main.py:165-175)
            g.apply(legal[int(rng.integers(len(legal)))])
        L = len(g.action_history)
        hw, hb = np.zeros(61, np.uint64), np.zeros(61, np.uint64)
        for t, obs in enumerate(g.state_history):
            hw[t], hb[t] = _obs_bits(obs)
        acts, rew = np.zeros(60, np.uint8), np.zeros(60, np.int8)
        acts[:L] = [a.index for a in g.action_history]
        rew[:L] = g.rewards
        recs.append(dict(hist_white=hw, hist_black=hb, actions=acts, rewards=rew, visits=visits, length=L))
        rb.save_game(g)
    win = recs[-window:]                                   # the FIFO the reference keeps (utilities.py:56-63), oldest first
    assert rb.buffer_size == len(win) == window
    st = dict(samples=0, batches=0, mismatched_images=0, mismatched_actions=0, mismatched_target_values=0, mismatched_target_policies=0,
              short_games=sum(r["length"] < 60 for r in win), samples_reaching_past_the_end=0, games_with_pass=sum(int((r["actions"][:r["length"]] == 64).any()) for r in win))
    orig_sample, orig_randint = random.sample, random.randint
    for bi in range(n_batches):
        K = (1, 3, 5)[bi % 3]
        U.config.num_unroll_steps = K
        random.seed(1000 * job + bi)
        idxes, poss = [], []

        def rec_sample(pop, k):
            r = orig_sample(pop, k)
            idxes.extend(r)
            return r

        def rec_randint(a, b):
            r = orig_randint(a, b)
            poss.append(r)
            return r

        random.sample, random.randint = rec_sample, rec_randint
        try:
            img, act, tv, tp = rb.sample_batch(K)
        finally:
            random.sample, random.randint = orig_sample, orig_randint
        img, act, tv, tp = (torch.FloatTensor(x).numpy() for x in (img, act, tv, tp))      # what update_weights feeds the network (main.py:236)
        oi, oa, otv, otp = O.replay_batch(win, list(zip(idxes, poss)), K)
        st["batches"] += 1
        st["samples"] += batch
        st["mismatched_images"] += int((oi != img).reshape(batch, -1).any(1).sum())
        st["mismatched_actions"] += int((oa != act).any((0, 2, 3, 4)).sum())
        st["mismatched_target_values"] += int((otv.view(np.uint32) != tv.view(np.uint32)).any(0).sum())
        st["mismatched_target_policies"] += int((otp.view(np.uint32) != tp.view(np.uint32)).any((0, 2)).sum())
        st["samples_reaching_past_the_end"] += sum(int(p + K >= win[i]["length"]) for i, p in zip(idxes, poss))
    U.config.num_unroll_steps = 3
    return st


def gen_bulk_replay(n_windows=64, n_games=40, window=32, batch=16, n_batches=24):
    """SURVEY 8(f) row 1 at bulk: n_windows x n_batches reference sample_batch calls (utilities.py:72-99 -> Game.make_image /
    make_target, reversi.py:301-341) against oracle/mz_oracle.c's replay_batch, bit for bit on all four outputs.  Only the
    verdict is written: tests/golden/bulk_replay_verdict.json."""
    import json
    import multiprocessing as mp
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    t0 = time.time()
    tot = {}
    with mp.Pool(8) as pool:
        for st in pool.imap_unordered(_bulk_replay_chunk, [(j, n_games, window, batch, n_batches) for j in range(n_windows)]):
            for k, v in st.items():
                tot[k] = tot.get(k, 0) + int(v)
    tot.update(windows=n_windows, games_per_window=n_games, window_size=window, batch_size=batch, unroll_steps=[1, 3, 5],
               seconds=round(time.time() - t0, 1), processes=8,
               reference="/root/reference utilities.ReplayBuffer.save_game / sample_batch, reversi.Game.make_image / make_target, imported in the build container",
               compared="image, action planes, target values, target policies of every sample vs oracle/mz_oracle.c (float32 bit patterns)")
    tot["mismatches"] = tot["mismatched_images"] + tot["mismatched_actions"] + tot["mismatched_target_values"] + tot["mismatched_target_policies"]
    with open(os.path.join(HERE, "bulk_replay_verdict.json"), "w") as f:
        json.dump(tot, f, indent=1)
    print(json.dumps(tot))
    assert tot["mismatches"] == 0


# ------------------------------------------------------------------------------------------------
# bulk verdict for the ARENA (test.py): whole reference arena games vs the oracle, verdict only
# ------------------------------------------------------------------------------------------------
def _ref_arena_game(mb, mw, seed):
    """one game of the reference's arena loop (test.py:51-63) between two HashNet Controllers, with the interventions of
    gen_arena (legal list sorted by index, loop bounded at 60 steps); returns (actions, recorded uniforms, white, black)"""
    import test as RT
    orig_choice = np.random.choice
    np.random.seed(seed)
    us = []

    def rec_choice(actions, p=None):
        st = np.random.get_state()
        r = orig_choice(actions, p=p)
        rs = np.random.RandomState()
        rs.set_state(st)
        us.append(rs.random_sample())
        return r

    np.random.choice = rec_choice
    try:
        env = RT.Environment()
        black_c = RT.Controller(RT.Player.BLACK, HashNetwork(mb))
        white_c = RT.Controller(RT.Player.WHITE, HashNetwork(mw))
        acts = []
        while not env.done and env.steps < 60:
            obs = env.observe()
            legal = sorted(env.legal_actions, key=lambda a: a.index)
            a_b = black_c.step(obs, env.player, legal)
            a_w = white_c.step(obs, env.player, legal)
            a = a_b if a_b else a_w
            env.step(a)
            acts.append(a.index)
    finally:
        np.random.choice = orig_choice
    return np.array(acts, np.uint8), np.array(us, np.float64), int(np.sum(env.board == 1)), int(np.sum(env.board == -1))


def _bulk_arena_chunk(job):
    from oracle import mz_oracle as O
    modes = ("hash", "sat", "const", "uniform")
    mb, mw = modes[job % 4], modes[(job // 4) % 4]
    acts, us, ws, bs = _ref_arena_game(mb, mw, 5000 + job)
    n, oacts, ows, obs_ = O.arena_game_hashnet(R.MODES[mb], R.MODES[mw], 100, 0.25, us)
    same = n == len(acts) and bool((oacts == acts).all())
    return dict(games=1, moves=len(acts), mismatched_games=int(not same), mismatched_scores=int((ows, obs_) != (ws, bs)),
                passes=int((acts == 64).sum()), black_wins=int(bs > ws), white_wins=int(ws > bs), draws=int(ws == bs))


def gen_bulk_arena(n_games=128):
    """SURVEY 8(f) row 3 at bulk: whole arena games played by the IMPORTED reference (test.py Controller.step: 100-simulation
    run_mcts without noise, softmax_sample at t = 0.25) between HashNet networks of every mode pair, against
    oracle/mz_oracle.c's arena_game_hashnet fed the same recorded uniforms: action log and final scores.  Verdict only."""
    import json
    import multiprocessing as mp
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    t0 = time.time()
    tot = {}
    with mp.Pool(8) as pool:
        for k, st in enumerate(pool.imap_unordered(_bulk_arena_chunk, range(n_games))):
            for key, v in st.items():
                tot[key] = tot.get(key, 0) + int(v)
            if k % 16 == 0:
                print("bulk_arena: %d games, %.0f s, mismatches %d" % (tot["games"], time.time() - t0, tot["mismatched_games"]), flush=True)
    tot.update(simulations=100, temperature=0.25, mode_pairs=16, seconds=round(time.time() - t0, 1), processes=8,
               reference="/root/reference/test.py Controller / run_mcts / softmax_sample imported in the build container",
               compared="every move of every game + final disc counts vs oracle/mz_oracle.c arena_game_hashnet")
    tot["mismatches"] = tot["mismatched_games"] + tot["mismatched_scores"]
    with open(os.path.join(HERE, "bulk_arena_verdict.json"), "w") as f:
        json.dump(tot, f, indent=1)
    print(json.dumps(tot))
    assert tot["mismatches"] == 0


# ------------------------------------------------------------------------------------------------
# bulk verdict for the TRAINING STEP (main.py:230-272): reference update_weights vs muzero_amd.train on the CPU, verdict only
# ------------------------------------------------------------------------------------------------
def gen_bulk_train(n_runs=24):
    """SURVEY 8(f) row 2 at bulk: n_runs x 2 consecutive reference update_weights steps (random-init reference Network, SGD as
    train_network builds it, random batches of B in {2,3,4,8}, K in {1,3,5}) against muzero_amd.train.update_weights on the same
    weights and batches, in its reference-shaped form (muzero_loss) and its batched form (muzero_loss_batched + deferred weight
    gradients) -- every parameter and gradient after every step.  fp32 on the CPU in both: the plain form must agree to the bit,
    the batched form to fp32 reassociation (a few units in the last place of a weight).  Verdict only: tests/golden/bulk_train_verdict.json."""
    import contextlib
    import io
    import json
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from muzero_amd import config as C, train as T
    from muzero_amd.network import Network
    T.LOG = None
    t0 = time.time()
    tot = dict(runs=0, steps=0, tensors=0, plain_bit_equal_tensors=0, plain_worst_rel=0.0, batched_worst_rel=0.0,
               plain_bit_equal_gradients=0, batched_worst_grad_rel_l2=0.0, batched_worst_grad_rel_max=0.0, loss_worst_abs=0.0)
    rels, steps_off = [], 0
    for j in range(n_runs):
        B, K = (2, 3, 4, 8)[j % 4], (1, 3, 5)[(j // 4) % 3]
        rng = np.random.default_rng(4000 + j)
        torch.manual_seed(7000 + j)
        ref = M.Network()
        ref.train()
        nets = {"ref": ref}
        for name in ("plain", "batched"):
            nets[name] = Network()
            nets[name].load_state_dict(ref.state_dict())
            nets[name].train()
        opts = {k: M.optim.SGD(n.parameters(), lr=0.001, weight_decay=M.config.lr_decay_rate, momentum=M.config.momentum) if k == "ref"
                else T.make_optimizer(n) for k, n in nets.items()}
        C.num_unroll_steps = K
        for step in range(2):
            batch = (rng.integers(0, 2, (B, 9, 8, 8)).astype(np.float32), rng.integers(0, 2, (K, B, 2, 8, 8)).astype(np.float32),
                     rng.uniform(-1, 1, (K + 1, B)).astype(np.float32),
                     torch.softmax(torch.from_numpy(rng.normal(size=(K + 1, B, 65)).astype(np.float32)), -1).numpy())
            w_before = [p.detach().clone() for p in ref.parameters()]
            ref.steps = 0
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                M.update_weights(opts["ref"], ref, batch)
            line = buf.getvalue().split()                       # 'step 0: p_loss 4.1744 v_loss 4.4605'
            for name in ("plain", "batched"):
                pl, vl = T.update_weights(opts[name], nets[name], batch, batched=(name == "batched"))
                tot["loss_worst_abs"] = max(tot["loss_worst_abs"], abs(float(pl) - float(line[3])), abs(float(vl) - float(line[5])))
                for a, b, w0 in zip(nets[name].parameters(), ref.parameters(), w_before):
                    upd = float((b.detach() - w0).abs().max())
                    rel = float((a.detach() - b.detach()).abs().max()) / max(upd, 1e-30)
                    tot[name + "_worst_rel"] = max(tot[name + "_worst_rel"], rel)
                    ga, gb = a.grad.double(), b.grad.double()      # the step's gradients are still in .grad on both sides
                    if name == "plain":
                        tot["tensors"] += 1
                        tot["plain_bit_equal_tensors"] += int(torch.equal(a.detach(), b.detach()))
                        tot["plain_bit_equal_gradients"] += int(torch.equal(a.grad, b.grad))
                    else:
                        rels.append(float((ga - gb).norm() / gb.norm().clamp_min(1e-300)))
                        tot["batched_worst_grad_rel_l2"] = max(tot["batched_worst_grad_rel_l2"], rels[-1])
                        tot["batched_worst_grad_rel_max"] = max(tot["batched_worst_grad_rel_max"], float((ga - gb).abs().max() / gb.abs().max().clamp_min(1e-300)))
            tot["steps"] += 1
            steps_off += int(max(rels[-len(w_before):]) > 1e-5)
        tot["runs"] += 1
        print("bulk_train: run %d (B %d, K %d), %.0f s: plain worst %.3g, batched worst %.3g" %
              (j, B, K, time.time() - t0, tot["plain_worst_rel"], tot["batched_worst_rel"]), flush=True)
    C.num_unroll_steps = 3
    rels.sort()
    # the batched form sums the same terms in another order; a ReLU input within fp32 rounding of zero can then sit on the other
    # side of its gate, which switches that unit's upstream gradient (seen the same way between GPU and CPU, tests/test_gpu_train.py)
    tot.update(batched_median_grad_rel_l2=rels[len(rels) // 2], batched_p99_grad_rel_l2=rels[int(len(rels) * 0.99)],
               batched_steps_with_a_tensor_above_1e5=steps_off)
    tot.update(seconds=round(time.time() - t0, 1), batch_sizes=[2, 3, 4, 8], unroll_steps=[1, 3, 5],
               reference="/root/reference/main.py update_weights + Network imported in the build container (CPU, fp32)",
               compared="every parameter and every gradient after every step (weights: max |mine - reference| relative to the step's largest weight change of that tensor, and in units of the fp32 spacing of the weight; gradients: relative L2 and max per tensor); losses vs the reference's printed line (4 decimals)")
    with open(os.path.join(HERE, "bulk_train_verdict.json"), "w") as f:
        json.dump(tot, f, indent=1)
    print(json.dumps(tot))
    assert tot["plain_bit_equal_tensors"] == tot["plain_bit_equal_gradients"] == tot["tensors"]
    assert tot["batched_median_grad_rel_l2"] <= 1e-6 and tot["batched_worst_grad_rel_l2"] <= 2e-2 and tot["loss_worst_abs"] <= 1e-4


# ------------------------------------------------------------------------------------------------
# checkpoints (main.py:223-224, test.py:247-250): cross-loading with the reference, verdict only
# ------------------------------------------------------------------------------------------------
def gen_checkpoint_xload(n=8):
    """(a) a file written the reference's way (torch.save(network.state_dict()), main.py:224) from the
    IMPORTED reference Network loads through muzero_amd.train.load_checkpoint; (b) a file written by
    muzero_amd.train.save_checkpoint loads the reference's way (Network().load_state_dict(torch.load(...)), test.py:247-250);
    after either, the two networks' initial_inference / recurrent_inference agree BIT FOR BIT on random inputs (CPU, fp32).
    Verdict only: tests/golden/checkpoint_xload_verdict.json."""
    import json
    import tempfile
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from muzero_amd import train as T
    from muzero_amd.network import Network
    tot = dict(cases=0, reference_file_loaded_by_build=0, build_file_loaded_by_reference=0, outputs_compared=0, outputs_not_bit_equal=0)

    def same(a_net, b_net, seed):
        g = torch.Generator().manual_seed(seed)
        img = (torch.rand(3, 9, 8, 8, generator=g) < 0.3).float()
        act = (torch.rand(3, 2, 8, 8, generator=g) < 0.1).float()
        bad = 0
        with torch.no_grad():
            a, b = a_net.initial_inference(img), b_net.initial_inference(img)
            a2, b2 = a_net.recurrent_inference(a[2], act), b_net.recurrent_inference(b[2], act)
        for u, v in zip(tuple(a) + tuple(a2), tuple(b) + tuple(b2)):
            tot["outputs_compared"] += 1
            bad += int(not torch.equal(u, v))
        tot["outputs_not_bit_equal"] += bad
        return bad == 0

    with tempfile.TemporaryDirectory() as d:
        for i in range(n):
            torch.manual_seed(100 + i)
            ref = M.Network()
            path = os.path.join(d, "%dmodel.pth" % (1000 * (i + 1)))
            torch.save(ref.state_dict(), path)                                   # main.py:224
            mine = T.load_checkpoint(Network(), path)
            tot["reference_file_loaded_by_build"] += int(same(ref.eval(), mine.eval(), i))
            torch.manual_seed(200 + i)
            mine2 = Network()
            path2 = T.save_checkpoint(mine2, 1000 * (i + 1), os.path.join(d, "models"))
            ref2 = M.Network()
            ref2.load_state_dict(torch.load(path2, map_location="cpu"))           # test.py:247 (a file this build wrote)
            tot["build_file_loaded_by_reference"] += int(same(ref2.eval(), mine2.eval(), 50 + i))
            tot["cases"] += 1
    tot.update(reference="/root/reference main.Network + torch.save / load_state_dict as main.py:224 and test.py:247-250 use them, imported in the build container",
               compared="value, policy, hidden of initial_inference and recurrent_inference on random inputs, torch.equal")
    with open(os.path.join(HERE, "checkpoint_xload_verdict.json"), "w") as f:
        json.dump(tot, f, indent=1)
    print(json.dumps(tot))
    assert tot["outputs_not_bit_equal"] == 0 and tot["reference_file_loaded_by_build"] == tot["build_file_loaded_by_reference"] == n


# ------------------------------------------------------------------------------------------------
# bulk verdict for WHOLE GAMES (main.play_game, main.py:69-92): reference self-play vs the oracle, verdict only
# ------------------------------------------------------------------------------------------------
def _bulk_game_chunk(job):
    """one game of the IMPORTED reference's play_game on a HashNet (mode by job), np.random draws recorded as in gen_game, then
    the oracle's play_game_hashnet on the recorded noise / uniforms: actions, rewards, visit distributions, root value bits"""
    from oracle import mz_oracle as O
    modes = ("hash", "sat", "const", "uniform")
    mode, S = modes[job % 4], (6, 16, 30, 50)[(job // 4) % 4]
    orig_legal = M.Game.legal_actions
    M.Game.legal_actions = lambda self: sorted(self.environment.legal_actions, key=lambda a: a.index)
    orig_dir, orig_choice = np.random.dirichlet, np.random.choice
    M.config.num_simulations = S
    np.random.seed(9000 + job)
    noises, us = [], []

    def rec_dirichlet(alpha, *a, **k):
        d = orig_dir(alpha, *a, **k)
        noises.append(np.array(d))
        return d

    def rec_choice(actions, p=None):
        st = np.random.get_state()
        r = orig_choice(actions, p=p)
        rs = np.random.RandomState()
        rs.set_state(st)
        us.append(rs.random_sample())
        return r

    np.random.dirichlet, np.random.choice = rec_dirichlet, rec_choice
    try:
        game = M.play_game(HashNetwork(mode))
    finally:
        np.random.dirichlet, np.random.choice = orig_dir, orig_choice
        M.Game.legal_actions = orig_legal
        M.config.num_simulations = 60
    n = len(game.action_history)
    g2 = M.Game()
    noise = np.zeros((n, 65), np.float64)
    for t in range(n):
        legal = sorted(a.index for a in g2.environment.legal_actions)
        noise[t, legal] = noises[t]
        g2.apply(M.Action(game.action_history[t].index, g2.to_play()))
    u = np.zeros(n, np.float64)
    u[:len(us)] = us
    on, oact, orew, ovis, orv, _, _ = O.play_game_hashnet(S, R.MODES[mode], noise, u)
    acts = np.array([a.index for a in game.action_history], np.uint8)
    same_len = on == n
    cv = np.stack([O.child_visits(v) for v in ovis[:n]]) if same_len else None
    return dict(games=1, moves=n,
                mismatched_actions=int(not same_len or not (oact[:n] == acts).all()),
                mismatched_rewards=int(not same_len or not (orew[:n] == np.array(game.rewards, np.int8)).all()),
                mismatched_child_visits=int(not same_len or not (cv == np.array(game.child_visits, np.float64)).all()),
                mismatched_root_value_bits=int(not same_len or orv[:n].view(np.uint32).tolist() !=
                                               np.array([float(v) for v in game.root_values], np.float32).view(np.uint32).tolist()),
                passes=int((acts == 64).sum()), done=int(game.terminal()), sampled_moves=len(us))


def gen_bulk_games(n_games=256):
    """whole reference self-play games (main.play_game: per move expand_node + add_exploration_noise + run_mcts + select_action
    + store_search_statistics, main.py:69-92) on HashNets of 4 modes at 6 / 16 / 30 / 50 simulations, against
    oracle/mz_oracle.c's play_game_hashnet fed the recorded noise and uniforms.  Verdict only."""
    import json
    import multiprocessing as mp
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    t0 = time.time()
    tot = {}
    with mp.Pool(8) as pool:
        for k, st in enumerate(pool.imap_unordered(_bulk_game_chunk, range(n_games))):
            for key, v in st.items():
                tot[key] = tot.get(key, 0) + int(v)
            if k % 16 == 0:
                print("bulk_games: %d games, %.0f s" % (tot["games"], time.time() - t0), flush=True)
    tot["mismatches"] = tot["mismatched_actions"] + tot["mismatched_rewards"] + tot["mismatched_child_visits"] + tot["mismatched_root_value_bits"]
    tot.update(simulations=[6, 16, 30, 50], modes=["hash", "sat", "const", "uniform"], seconds=round(time.time() - t0, 1), processes=8,
               reference="/root/reference/main.py play_game imported in the build container (legal list sorted by index, np.random draws recorded)",
               compared="every move, reward, child_visits row (float64) and root value (float32 bits) of every game vs oracle/mz_oracle.c play_game_hashnet")
    with open(os.path.join(HERE, "bulk_games_verdict.json"), "w") as f:
        json.dump(tot, f, indent=1)
    print(json.dumps(tot))
    assert tot["mismatches"] == 0


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what == "bulk":
        gen_bulk(int(sys.argv[2]) if len(sys.argv) > 2 else 131072)
        sys.exit(0)
    if what == "bulk_games":
        gen_bulk_games(int(sys.argv[2]) if len(sys.argv) > 2 else 256)
        sys.exit(0)
    if what == "checkpoint_xload":
        gen_checkpoint_xload()
        sys.exit(0)
    if what == "bulk_train":
        gen_bulk_train(int(sys.argv[2]) if len(sys.argv) > 2 else 24)
        sys.exit(0)
    if what == "bulk_arena":
        gen_bulk_arena(int(sys.argv[2]) if len(sys.argv) > 2 else 128)
        sys.exit(0)
    if what == "bulk_replay":
        gen_bulk_replay(int(sys.argv[2]) if len(sys.argv) > 2 else 64)
        sys.exit(0)
    if what == "bulk_mcts":
        gen_bulk_mcts(int(sys.argv[2]) if len(sys.argv) > 2 else 10240)
        sys.exit(0)
    if what in ("env", "all"):
        gen_env()
    if what in ("mcts", "all"):
        gen_mcts()
    if what in ("features", "all"):
        gen_features()
    if what in ("game", "all"):
        gen_game()
    if what in ("replay", "all"):
        gen_replay()
    if what in ("arena", "all"):
        gen_arena()
