"""CPU: the oracle's restatement of ReplayBuffer.sample_batch / Game.make_target (utilities.py:72-99,
reversi.py:314-341) against the fixture generated from the reference (tests/golden/replay.npz:
8 games of 60/9/14 moves pushed through a window of 6, five recorded sample_batch calls)."""
import os

import numpy as np

from oracle import mz_oracle as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "replay.npz"))


def fixture_games():
    return [dict(hist_white=G["hist_white"][i], hist_black=G["hist_black"][i], actions=G["actions"][i], rewards=G["rewards"][i],
                 visits=G["visits"][i], length=int(G["length"][i])) for i in range(len(G["length"]))]


def window(games):
    """FIFO of utilities.py:56-63: the last window_size games saved, oldest first"""
    return games[int(G["evicted"]):]


def test_sample_batch_matches_reference():
    games = window(fixture_games())
    assert len(games) == int(G["window_size"])
    for b in range(int(G["n_batches"])):
        pre = "b%d_" % b
        K = int(G[pre + "K"])
        picks = list(zip(G[pre + "game_idx"].tolist(), G[pre + "pos"].tolist()))
        image, acts, tv, tp = O.replay_batch(games, picks, K)
        assert image.shape == (4, 9, 8, 8) and acts.shape == (K, 4, 2, 8, 8) and tv.shape == (K + 1, 4) and tp.shape == (K + 1, 4, 65)
        assert (image == G[pre + "image"].astype(np.float32)).all()
        assert (acts == G[pre + "actions"].astype(np.float32)).all()
        assert (tv.view(np.uint32) == G[pre + "target_values"].view(np.uint32)).all()
        assert (tp.view(np.uint32) == G[pre + "target_policies"].view(np.uint32)).all()


def test_make_target_matches_selfplay_fixture():
    """the same function against Game.make_target outputs recorded on whole play_game games"""
    S = np.load(os.path.join(os.path.dirname(__file__), "golden", "selfplay_games.npz"))
    for g in range(3):
        cv = S["child_visits_%d" % g]
        n = cv.shape[0]
        # integer visit counts that normalise to the recorded child_visits: N = round(cv * S)
        visits = np.rint(cv * int(S["S"][g])).astype(np.int32)
        assert (visits.sum(1) == int(S["S"][g])).all()
        rew = np.zeros(60, np.int8)
        rew[:n] = S["rewards_%d" % g]
        hw = np.zeros(61, np.uint64)
        row = 0
        for K in (3, 5):
            for si in S["target_index_%d" % g].tolist():
                want_v, want_p = S["target_values_%d" % g][row], S["target_policies_%d" % g][row]
                row += 1
                if si + K > n:
                    continue  # outside sample_position's range (utilities.py:104)
                _, _, tv, tp = O.replay_sample(hw, hw, S["actions_%d" % g], rew, np.pad(visits, ((0, 60 - n), (0, 0))), n, si, K)
                assert (tv == want_v[:K + 1].astype(np.float32)).all()
                assert (tp == want_p[:K + 1].astype(np.float32)).all()


def test_bulk_reference_verdict_is_committed_and_clean():
    """SURVEY 8(f) row 1 at bulk: >= 100k samples drawn by the IMPORTED reference's ReplayBuffer.sample_batch
    (tests/golden/make_golden.py bulk_replay, build container only: random games with voluntary passes, 9..60 moves, visit counts
    of 1..800 simulations with unvisited children, FIFO eviction, K = 1 / 3 / 5) against this oracle, float32 bit patterns of all
    four outputs; only the verdict travels."""
    import json
    v = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "bulk_replay_verdict.json")))
    assert v["samples"] >= 100_000 and v["samples"] == v["batches"] * v["batch_size"]
    assert v["mismatches"] == v["mismatched_images"] == v["mismatched_actions"] == v["mismatched_target_values"] == v["mismatched_target_policies"] == 0
    assert v["short_games"] > 1000 and v["samples_reaching_past_the_end"] > 1000 and v["games_with_pass"] > 1000
    assert v["games_per_window"] > v["window_size"]   # eviction happened in every window
