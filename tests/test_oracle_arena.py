"""CPU: the oracle's arena restatement (test.py Controller / run_mcts with 100 sims and no noise /
softmax_sample with temperature) against fixtures generated from the reference's test.py."""
import os

import numpy as np

from oracle import mz_oracle as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "arena.npz"))


def test_softmax_sample_temperatures():
    for v, t, u, a in zip(G["ss_visits"], G["ss_t"], G["ss_u"], G["ss_action"]):
        assert O.softmax_sample(v, t, u) == a, (t, u)


def test_arena_games_and_controller_images():
    for g in range(int(G["n_games"])):
        n, actions, ws, bs = O.arena_game_hashnet(int(G["mode_black_%d" % g]), int(G["mode_white_%d" % g]), 100, 0.25, G["u_%d" % g])
        want = G["actions_%d" % g]
        assert n == len(want) and (actions == want).all(), g
        assert (ws, bs) == (int(G["white_score_%d" % g]), int(G["black_score_%d" % g]))
        # replay the game to rebuild the observation history, then every Controller image
        e = O.EnvBatch(1)
        hw, hb = [e.white[0]], [e.black[0]]
        for a in want:
            e.step(np.array([a], np.uint8))
            hw.append(e.white[0]); hb.append(e.black[0])
        hw, hb = np.array(hw, np.uint64), np.array(hb, np.uint64)
        for t in range(n):
            img = O.controller_image(hw, hb, t, t % 2 == 0)
            assert (img == G["images_%d" % g][t].astype(np.float32)).all(), (g, t)


def test_bulk_reference_verdict_is_committed_and_clean():
    """SURVEY 8(f) row 3 at bulk: whole arena games played by the IMPORTED reference's test.py (Controller.step: 100-simulation
    run_mcts without noise, softmax_sample at t = 0.25; tests/golden/make_golden.py bulk_arena, build container only) between
    HashNet networks of all 16 mode pairs, against this oracle's arena_game_hashnet on the recorded uniforms; only the verdict
    travels."""
    import json
    v = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bulk_arena_verdict.json")))
    assert v["games"] >= 500 and v["moves"] >= 50 * v["games"] and v["simulations"] == 100
    assert v["mismatches"] == v["mismatched_games"] == v["mismatched_scores"] == 0
    assert v["passes"] > 50 and v["black_wins"] > 50 and v["white_wins"] > 50
