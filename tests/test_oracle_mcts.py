"""Pin the C oracle's tree search against fixtures generated from the imported reference
(tests/golden/make_golden.py: run_mcts / select_child / ucb_score / expand_node / backpropagate /
add_exploration_noise / select_action, with HashNet or recorded real-network outputs).  CPU only."""
import ctypes
import os

import numpy as np
import pytest

import refspec as R
from oracle import mz_oracle as O

MODE_NAME = {v: k for k, v in R.MODES.items()}


@pytest.fixture(scope="module")
def fx(golden_dir):
    return np.load(os.path.join(golden_dir, "mcts_traces.npz"))


def case_paths(fx, i):
    s0, s1 = fx["sim_off"][i], fx["sim_off"][i + 1]
    off = fx["path_off"]
    return [fx["path_actions"][off[s]:off[s + 1]] for s in range(s0, s1)]


def test_hashnet_c_matches_python_statement():
    for h in (1, 12345, R.P31 - 1, 99999989):
        for mode in range(4):
            v, p = O.hashnet_outputs(h, mode)
            v2, p2 = R.hashnet_outputs(h, MODE_NAME[mode])
            assert v == v2 and (p == p2).all()
        for a in (0, 1, 8, 27, 63, 64):
            for plane in (0, 1):
                code = O.lib().mzo_hashnet_action_code(a, plane)
                assert code == R.action_code(a, plane)
                assert code == R.hashnet_action_code(O.action_encode(a, 1 if plane == 0 else -1))
                assert O.lib().mzo_hashnet_step(h, code) == R.hashnet_step(h, code)
    img = (np.random.default_rng(1).random((9, 8, 8)) < 0.3).astype(np.float32)
    assert O.lib().mzo_hashnet_image_hash(O.ptr(img)) == R.hashnet_image_hash(img)


def run_case(fx, i):
    """step-wise oracle search for fixture case i, network outputs replayed / recomputed"""
    S = int(fx["S"][i])
    bounds = None if fx["bounds_none"][i] else (float(fx["bound_lo"][i]), float(fx["bound_hi"][i]))
    t = O.Tree(S, discount=float(fx["discount"][i]), known_bounds=bounds)
    noise = fx["noise"][i] if fx["use_noise"][i] else None
    t.root_init(fx["to_play"][i], fx["legal"][i], fx["pass_flag"][i], fx["root_policy"][i], noise)
    kind = int(fx["kind"][i])
    node_h = {0: R.hashnet_image_hash(fx["image"][i])}
    paths = []
    for s in range(S):
        d, pid, a, plane, path = t.select()
        paths.append(path)
        if kind == 0:
            h = R.hashnet_step(node_h[pid], R.action_code(a, plane))
            node_h[s + 1] = h
            v, p = R.hashnet_outputs(h, MODE_NAME[int(fx["mode"][i])])
        else:
            k = fx["net_off"][i] + s
            v, p = fx["net_values"][k], fx["net_policies"][k]
        t.expand_backup(v, p)
    return t, paths


def test_all_fixture_searches(fx):
    n = len(fx["S"])
    assert n >= 40
    for i in range(n):
        t, paths = run_case(fx, i)
        exp = case_paths(fx, i)
        for s, (got, want) in enumerate(zip(paths, exp)):
            assert len(got) == len(want) and (got == want).all(), ("case", i, "sim", s, got, want)
        visits, wsum, prior, rn, rv = t.root_stats()
        assert (visits == fx["visits"][i]).all(), i
        assert rn == fx["root_n"][i]
        assert wsum.view(np.uint32).tolist() == fx["wsum"][i].view(np.uint32).tolist(), i  # bit-exact fp32
        assert (prior == fx["prior"][i]).all(), i                                            # bit-exact double
        assert np.float32(rv).view(np.uint32) == fx["root_value"][i].view(np.uint32), i
        assert (O.child_visits(visits) == fx["child_visits"][i]).all()
        assert O.select_action(visits, 30, 0.0) == fx["greedy"][i]
        assert O.select_action(visits, 0, fx["u"][i]) == fx["sampled"][i]


def test_root_prior_hashnet_consistency(fx):
    """the fixture's root policy is HashNet(image): ties image hashing into the chain"""
    for i in np.nonzero(fx["kind"] == 0)[0]:
        v, p = R.hashnet_outputs(R.hashnet_image_hash(fx["image"][i]), MODE_NAME[int(fx["mode"][i])])
        assert (p == fx["root_policy"][i]).all() and v == fx["root_value0"][i]


def test_c_search_driver_equals_stepwise(fx):
    """mzo_search_hashnet (whole search in C, used for bulk GPU parity) == the step-wise drive"""
    for i in np.nonzero((fx["kind"] == 0))[0][:12]:
        S = int(fx["S"][i])
        bounds = None if fx["bounds_none"][i] else (float(fx["bound_lo"][i]), float(fx["bound_hi"][i]))
        t = O.Tree(S, discount=float(fx["discount"][i]), known_bounds=bounds)
        noise = fx["noise"][i] if fx["use_noise"][i] else None
        t.search_hashnet(fx["to_play"][i], fx["legal"][i], fx["pass_flag"][i], noise,
                         R.hashnet_image_hash(fx["image"][i]), int(fx["mode"][i]), S)
        visits = t.root_stats()[0]
        assert (visits == fx["visits"][i]).all(), i


def test_tie_break_is_highest_index_at_first_simulation(fx):
    """sim 1 at the opening ties all children at 0.5 -> reference picks action 44"""
    i = int(np.nonzero((fx["steps"] == 0) & (fx["kind"] == 0))[0][0])
    assert case_paths(fx, i)[0][0] == 44
    t, paths = run_case(fx, i)
    assert paths[0][0] == 44
    # the non-default switch offered for BASELINE.json's "lowest index" wording
    t2 = O.Tree(4, tie_lowest=True)
    t2.root_init(-1, fx["legal"][i], 0, fx["root_policy"][i], None)
    assert t2.select()[2] == 19


def test_negamax_switch_flips_the_value_term_only(fx):
    """non-default config.negamax_value: ucb_score adds normalize(-child.value()) instead of the reference's
    normalize(child.value()) (main.py:159); prior term, visit accounting and unvisited children are untouched"""
    i = int(np.nonzero((fx["steps"] == 0) & (fx["kind"] == 0))[0][0])
    S = 12
    ref, neg = O.Tree(S), O.Tree(S, negamax=True)
    img = R.hashnet_image_hash(fx["image"][i])
    for t in (ref, neg):
        t.search_hashnet(-1, int(fx["legal"][i]), 0, None, img, int(fx["mode"][i]), S)
        assert t.root_stats()[0].clip(min=0).sum() == S
    # identical until a visited child's value enters a comparison, then the two searches part
    assert (ref.root_stats()[0] != neg.root_stats()[0]).any()
    c = ctypes.c_int(0)
    for a in (19, 26, 37, 44):
        sr = O.lib().mzo_ucb_score_probe(ref.h, 0, a, ctypes.byref(c))
        assert np.isfinite(sr)
    # a single visited child: scores differ by exactly the sign of the value inside normalize
    t1, t2 = O.Tree(2), O.Tree(2, negamax=True)
    for t in (t1, t2):
        t.search_hashnet(-1, int(fx["legal"][i]), 0, None, img, int(fx["mode"][i]), 1)
    a = int(np.argmax(t1.root_stats()[0]))
    s1 = O.lib().mzo_ucb_score_probe(t1.h, 0, a, ctypes.byref(c))
    s2 = O.lib().mzo_ucb_score_probe(t2.h, 0, a, ctypes.byref(c))
    v = float(t1.root_stats()[1][a])   # value_sum of the once-visited child = its value
    assert v != 0 and abs((s1 - s2) - v) < 1e-6   # normalize over (-1, 1): (v+1)/2 - (-v+1)/2 = v


def test_whole_games(golden_dir):
    """main.play_game on HashNet: actions, rewards, visit distributions and root values of 3 games"""
    g = np.load(os.path.join(golden_dir, "selfplay_games.npz"))
    for i, S in enumerate(g["S"]):
        n, actions, rewards, visits, root_values, hw, hb = O.play_game_hashnet(int(S), 0, g["noise_%d" % i], g["u_%d" % i])
        assert n == len(g["actions_%d" % i]) == 60
        assert (actions == g["actions_%d" % i]).all(), i
        assert (rewards == g["rewards_%d" % i]).all()
        cv = np.stack([O.child_visits(v) for v in visits])
        assert (cv == g["child_visits_%d" % i]).all()
        assert root_values.view(np.uint32).tolist() == g["root_values_%d" % i].view(np.uint32).tolist()


def test_bulk_reference_search_verdict_is_committed_and_clean(golden_dir):
    """>= 10k searches run by the IMPORTED reference (main.expand_node / add_exploration_noise / run_mcts on HashNet;
    tests/golden/make_golden.py bulk_mcts, build container only) against this oracle's mzo_search_hashnet on the same roots,
    noise and knobs: visits, value_sum bits, prior bits, root value bits, per-simulation depth and leaf action (whole paths on
    every eighth search).  Only the verdict travels."""
    import json
    v = json.load(open(os.path.join(golden_dir, "bulk_mcts_verdict.json")))
    assert v["searches"] >= 10_000 and v["mismatches"] == 0
    for k in ("mism_visits", "mism_w_bits", "mism_prior_bits", "mism_root_value_bits", "mism_root_n", "mism_depth", "mism_leaf_action",
              "mism_paths"):
        assert v[k] == 0, k
    assert v["paths_checked_searches"] >= 1000 and v["pass_only_roots"] >= 50 and v["max_depth"] >= 6
    assert set(v["by_S"]) == {"50", "60", "100", "200"} and set(v["by_mode"]) == {"hash", "sat", "const", "uniform"}
    assert {"None", "(-1, 1)", "(-0.25, 0.3)"} <= set(v["by_bounds"]) and {"0.9", "0.997", "1.0"} <= set(v["by_discount"])
    assert min(v["by_noise"].values()) >= 2000 and len(v["by_moves_decile"]) == 6


def test_bulk_reference_games_verdict_is_committed_and_clean(golden_dir):
    """whole self-play games of the IMPORTED reference (main.play_game, main.py:69-92: root expansion, Dirichlet noise, run_mcts,
    select_action with its t = 1 / t = 0 switch at move 20, store_search_statistics; tests/golden/make_golden.py bulk_games, build
    container only) on HashNets of 4 modes at 6 / 16 / 30 / 50 simulations against this oracle's play_game_hashnet on the recorded
    noise and uniforms: every move, reward, child_visits row and root value.  Only the verdict travels."""
    import json
    v = json.load(open(os.path.join(golden_dir, "bulk_games_verdict.json")))
    assert v["games"] >= 2000 and v["moves"] == 60 * v["games"]
    assert v["mismatches"] == v["mismatched_actions"] == v["mismatched_rewards"] == v["mismatched_child_visits"] == v["mismatched_root_value_bits"] == 0
    assert v["passes"] > 100 and 0 < v["done"] < v["games"] and v["sampled_moves"] == 20 * v["games"]
