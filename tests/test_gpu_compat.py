"""The reference-shaped Python API (muzero_amd.compat) on the GPU: tests read like calls into the
reference's reversi.py / main.py, and are checked against fixtures the reference produced."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from hashnet_ref import HashNetwork  # noqa: E402
from oracle import mz_oracle as O  # noqa: E402  (checker only)


@pytest.fixture()
def C():
    import muzero_amd.compat as compat
    saved = {k: getattr(compat.config, k) for k in ("num_simulations", "known_bounds", "discount")}
    yield compat
    for k, v in saved.items():
        setattr(compat.config, k, v)


def test_environment_follows_reference_trace(C, golden_dir):
    fx = np.load(os.path.join(golden_dir, "env_traces.npz"))
    for g in (0, 5, 17):
        env = C.Environment()
        assert env.player is C.Player.BLACK and env.steps == 0 and not env.done
        assert [a.index for a in env.legal_actions] == [19, 26, 37, 44]
        assert env.possible_pos == {(2, 2), (2, 3), (2, 4), (2, 5), (3, 2), (3, 5), (4, 2), (4, 5), (5, 2), (5, 3), (5, 4), (5, 5)}
        for t in range(60):
            a = int(fx["full_actions"][g, t])
            obs, reward, done = env.step(C.Action(a, env.player))
            w = int(fx["full_white"][g, t + 1]); b = int(fx["full_black"][g, t + 1])
            assert obs.dtype == bool and obs.shape == (2, 8, 8)
            assert (np.packbits(obs[0].reshape(-1), bitorder="little").view("<u8")[0] == w)
            assert (np.packbits(obs[1].reshape(-1), bitorder="little").view("<u8")[0] == b)
            assert reward == fx["full_rewards"][g, t] and done == bool(fx["full_done"][g, t + 1])
            assert env.player.value == fx["full_to_play"][g, t + 1] and env.steps == t + 1
            assert (env.board == 1).sum() == bin(w).count("1")


def test_environment_error_behaviour(C):
    env = C.Environment()
    with pytest.raises(AssertionError):            # wrong player (reversi.py:131)
        env.step(C.Action(19, C.Player.WHITE))
    with pytest.raises(AssertionError):            # not a legal action (reversi.py:143-144)
        env.step(C.Action(0, C.Player.BLACK))
    with pytest.raises(AssertionError):            # Action index range (reversi.py:26)
        C.Action(65, C.Player.BLACK)
    assert env.steps == 0                          # nothing happened
    obs, r, done = env.step(C.Action(64, C.Player.BLACK))  # a pass is accepted without check (reversi.py:135-139)
    assert env.player is C.Player.WHITE and env.steps == 1 and r == 0 and not done
    assert C.Action(64, C.Player.WHITE).encode().sum() == 64 and C.Action(10, C.Player.BLACK).encode()[1].sum() == 16


def test_play_game_reproduces_reference_games(C, golden_dir):
    """main.play_game on HashNet with the same np.random seed: the shims draw Dirichlet noise and
    sampling uniforms from np.random exactly as the reference does, so whole games coincide."""
    g = np.load(os.path.join(golden_dir, "selfplay_games.npz"))
    for i, (S, seed) in enumerate([(16, 11), (16, 12)]):
        C.config.num_simulations = S
        np.random.seed(seed)
        game = C.play_game(HashNetwork("hash"))
        assert [a.index for a in game.action_history] == g["actions_%d" % i].tolist()
        assert game.rewards == g["rewards_%d" % i].tolist()
        assert (np.array(game.child_visits) == g["child_visits_%d" % i]).all()
        rv = np.array([float(v) for v in game.root_values], np.float32)
        assert rv.view(np.uint32).tolist() == g["root_values_%d" % i].view(np.uint32).tolist()
        assert game.terminal() == bool(g["done"][i])
        # Game.make_target (reversi.py:314-341)
        k = 0
        for K in (3, 5):
            for si in g["target_index_%d" % i]:
                tv, tp = game.make_target(int(si), K)
                assert tv == g["target_values_%d" % i][k][:K + 1].tolist()
                assert (np.array([list(x) for x in tp]) == g["target_policies_%d" % i][k][:K + 1]).all()
                k += 1
        # Game.make_image at a few indices vs the oracle restatement
        hw = game._hw[:, 0].cpu().numpy().view(np.uint64); hb = game._hb[:, 0].cpu().numpy().view(np.uint64)
        for si in (0, 1, 4, 33, 60):
            assert (game.make_image(si) == O.make_image(hw, hb, si)).all()


def test_search_entry_points_one_by_one(C, golden_dir):
    """run_mcts re-assembled from select_child / expand_node / backpropagate (main.py:103-125),
    with ucb_score's Python-float vs fp32-tensor kinds, against the fixture and the oracle."""
    fx = np.load(os.path.join(golden_dir, "mcts_traces.npz"))
    i = int(np.nonzero((fx["kind"] == 0) & (fx["S"] == 50) & (fx["steps"] == 7))[0][0])
    S = 12
    C.config.num_simulations = S
    net = HashNetwork("hash")
    game = C.Game()
    import refspec as R
    for step in range(7):
        game.apply(C.Action(R.playout_action(1261, int(i), step, sum(1 << a.index for a in game.legal_actions() if a.index < 64)), game.to_play()))
    assert (game.make_image(7) == fx["image"][i]).all()
    root = C.Node(0, game.to_play())
    out0 = net.initial_inference(torch.from_numpy(game.make_image(game.environment.steps)))
    C.expand_node(root, game.legal_actions(), out0)
    noise = fx["noise"][i]
    np.random.seed(0)
    C.add_exploration_noise(root)            # draws its own noise; re-initialise with the fixture's below
    from muzero_amd.compat import _root_init
    _root_init(root._tree, torch.from_numpy(noise[None]).cuda())
    assert [a for a in root.children] == [a.index for a in game.legal_actions()]
    assert np.allclose([c.prior for c in root.children.values()], fx["prior"][i][fx["visits"][i] >= 0], rtol=0, atol=0)
    # oracle twin
    t = O.Tree(S)
    t.root_init(fx["to_play"][i], fx["legal"][i], fx["pass_flag"][i], fx["root_policy"][i], noise)
    mm = C.MinMaxStats(C.config.known_bounds)
    s0, off = fx["sim_off"][i], fx["path_off"]
    for s in range(S):
        want = fx["path_actions"][off[s0 + s]:off[s0 + s + 1]]
        node, search_path, acts = root, [root], []
        while node.expanded():
            if s == 3 and node is root:  # ucb_score kinds at the root after 3 simulations
                for a, ch in root.children.items():
                    sc = C.ucb_score(root, ch, mm)
                    osc, is_t = t.ucb_score(0, a)
                    assert torch.is_tensor(sc) == is_t == (ch.visit_count > 0)
                    assert float(sc) == osc
            a, node = C.select_child(node, mm)
            acts.append(a)
            search_path.append(node)
        assert acts == want.tolist(), (s, acts, want)
        d, pid, oa, plane, _ = t.select()
        parent = search_path[-2]
        last = C.Action(acts[-1], node.player)
        out = net.recurrent_inference(parent.hidden_state.cpu(), torch.from_numpy(last.encode()))
        C.expand_node(node, [C.Action(k, node.player) for k in range(65)], out)
        C.backpropagate(search_path, out[0], last.player, C.config.discount, mm)
        t.expand_backup(float(out[0]), out[1].numpy().reshape(65))
    ov = t.root_stats()[0]
    got = np.full(65, -1, np.int32)
    for a, ch in root.children.items():
        got[a] = ch.visit_count
    assert (got == ov).all() and root.visit_count == S
    assert np.float32(float(root.value())).view(np.uint32) == np.float32(t.root_stats()[4]).view(np.uint32)
    assert mm.minimum == -1 and mm.maximum == 1


def test_run_mcts_and_select_action_vs_fixture(C, golden_dir):
    fx = np.load(os.path.join(golden_dir, "mcts_traces.npz"))
    for i in np.nonzero((fx["kind"] == 0) & (fx["bounds_none"] == 0) & (fx["discount"] == 1.0) & (fx["bound_lo"] == -1) & (fx["mode"] == 0))[0][:6]:
        S = int(fx["S"][i])
        C.config.num_simulations = S
        net = HashNetwork("hash")
        root = C.Node(0, C.Player(int(fx["to_play"][i])))
        legal = [C.Action(a, root.player) for a in range(64) if (int(fx["legal"][i]) >> a) & 1] or [C.Action(64, root.player)]
        C.expand_node(root, legal, net.initial_inference(torch.from_numpy(fx["image"][i])))
        if fx["use_noise"][i]:
            from muzero_amd.compat import _root_init
            _root_init(root._tree, torch.from_numpy(fx["noise"][i][None]).cuda())
        C.run_mcts(root, [], net)
        got = np.full(65, -1, np.int32)
        for a, ch in root.children.items():
            got[a] = ch.visit_count
        assert (got == fx["visits"][i]).all(), i
        assert C.select_action(30, root) == fx["greedy"][i]
        np.random.seed(1000 + int(fx["white"][i] % 1000))
        st = np.random.get_state()
        a = C.select_action(0, root)
        rs = np.random.RandomState(); rs.set_state(st)
        assert a == O.select_action(got, 0, rs.random_sample())
        game = C.Game()
        game.store_search_statistics(root)
        assert (np.array(game.child_visits[0]) == fx["child_visits"][i]).all()


def test_noise_follows_the_callers_action_order(C):
    """add_exploration_noise (main.py:191-196) hands the i-th Dirichlet draw to the i-th action of the list expand_node
    received -- the reference's root.children order, which is Python-set order there (reversi.py:121-123)"""
    net = HashNetwork("hash")
    game = C.Game()
    out0 = net.initial_inference(torch.from_numpy(game.make_image(0)))
    order = [44, 19, 37, 26]                       # an unsorted legal list of the opening position
    priors = {}
    for acts in (order, sorted(order)):
        root = C.Node(0, game.to_play())
        C.expand_node(root, [C.Action(a, game.to_play()) for a in acts], out0)
        clean = {a: ch.prior for a, ch in root.children.items()}
        np.random.seed(5)
        C.add_exploration_noise(root)
        np.random.seed(5)
        drawn = np.random.dirichlet([C.config.root_dirichlet_alpha] * 4)
        frac = C.config.root_exploration_fraction
        for k, a in enumerate(acts):
            assert root.children[a].prior == clean[a] * (1 - frac) + drawn[k] * frac, (acts, a)
        priors[tuple(acts)] = [root.children[a].prior for a in sorted(order)]
    assert priors[tuple(order)] != priors[tuple(sorted(order))]
