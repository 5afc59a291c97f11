"""Pin the C oracle's env + features against fixtures generated from the imported reference
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

import refspec as R
from oracle import mz_oracle as O


@pytest.fixture(scope="module")
def env_fx(golden_dir):
    return np.load(os.path.join(golden_dir, "env_traces.npz"))


def test_rng_and_digest_match_python_statement():
    for seed, g, s in [(1261, 0, 0), (1261, 12345, 59), (2**63 + 5, 2**40 + 3, 7)]:
        assert O.lib().mzo_playout_rand(seed, g, s) == R.playout_rand(seed, g, s)
    rng = np.random.default_rng(0)
    for _ in range(200):
        legal = int(rng.integers(0, 2**63)) | (int(rng.integers(0, 2)) << 63)
        if rng.integers(0, 4) == 0:
            legal &= int(rng.integers(0, 2**63))
        seed, g, s = int(rng.integers(0, 2**62)), int(rng.integers(0, 2**40)), int(rng.integers(0, 60))
        assert O.lib().mzo_playout_action(seed, g, s, legal) == R.playout_action(seed, g, s, legal)
    assert O.lib().mzo_playout_action(1, 2, 3, 0) == 64
    h = O.lib().mzo_digest_step(5, 2**63 + 1, 77, -1, 60, 1, -1, 2**64 - 1, 1)
    assert h == R.digest_step(5, 2**63 + 1, 77, -1, 60, 1, -1, 2**64 - 1, 1)


def test_reset_is_reference_opening(env_fx):
    e = O.EnvBatch(3)
    assert (e.white == env_fx["full_white"][0, 0]).all() and (e.black == env_fx["full_black"][0, 0]).all()
    assert (e.to_play == -1).all() and (e.steps == 0).all() and (e.done == 0).all()
    assert (e.legal == sum(1 << i for i in (19, 26, 44, 37))).all()


@pytest.mark.parametrize("prefix", ["full_", "volfull_"])
def test_step_by_step_states(env_fx, prefix):
    """every field after every step of fully recorded games (incl. voluntary passes)"""
    A = env_fx[prefix + "actions"]
    n = A.shape[0]
    e = O.EnvBatch(n)
    for t in range(60):
        reward, status = e.step(A[:, t])
        assert (status == 0).all()
        for name, arr in (("white", e.white), ("black", e.black), ("to_play", e.to_play), ("steps", e.steps),
                          ("done", e.done), ("legal", e.legal)):
            assert (arr == env_fx[prefix + name][:, t + 1]).all(), (name, t)
        assert ((e.legal == 0).astype(np.uint8) == env_fx[prefix + "pass_flag"][:, t + 1]).all() or t == 59
        assert (reward == env_fx[prefix + "rewards"][:, t]).all(), t
    # terminal quirk (reversi.py:170-172): a placing 60th step leaves to_play/legal stale, done=1;
    # a passing 60th step leaves done=0
    last_pass = A[:, 59] == 64
    assert (e.done == (~last_pass).astype(np.uint8)).all()
    # stepping a finished game is a flagged no-op in the batch API
    w0 = e.white.copy()
    reward, status = e.step(np.full(n, 64, np.uint8))
    assert (status[e.done == 1] == 2).all() and (e.white[e.done == 1] == w0[e.done == 1]).all()


@pytest.mark.parametrize("prefix,seed,vol,n", [("dig_", 1261, 0, 4096), ("vol_", 1262, 1, 512)])
def test_playout_digests(env_fx, prefix, seed, vol, n):
    """4096 + 512 whole games: action log, digest of all 61 states, final state, last reward"""
    e, digest, actions, last_reward = O.random_playout(seed, 0, n, 60, vol, want_actions=True)
    assert (actions == env_fx[prefix + "actions"]).all()
    assert (digest == env_fx[prefix + "digest"]).all()
    assert (e.white == env_fx[prefix + "final_white"]).all() and (e.black == env_fx[prefix + "final_black"]).all()
    assert (e.to_play == env_fx[prefix + "final_to_play"]).all() and (e.done == env_fx[prefix + "final_done"]).all()
    assert (last_reward == env_fx[prefix + "rewards_last"]).all()


def test_illegal_actions_are_flagged_and_leave_state(env_fx):
    e = O.EnvBatch(4)
    before = (e.white.copy(), e.black.copy(), e.steps.copy())
    reward, status = e.step(np.array([0, 27, 65, 200], np.uint8))  # not capturing / occupied / out of range
    assert (status == 1).all() and (reward == 0).all()
    assert (e.white == before[0]).all() and (e.black == before[1]).all() and (e.steps == before[2]).all()


def test_legal_mask_on_recorded_positions(env_fx):
    W, B, TP, L = (env_fx["full_" + k][:, :60].reshape(-1) for k in ("white", "black", "to_play", "legal"))
    assert (O.legal_mask(B, W, TP) == L).all()


def test_features(golden_dir):
    fx = np.load(os.path.join(golden_dir, "features.npz"))
    for idx, pl, planes in zip(fx["enc_index"], fx["enc_player"], fx["enc_planes"]):
        assert (O.action_encode(int(idx), int(pl)) == planes).all()
    assert O.action_encode(64, 1).sum() == 64  # pass sets the whole plane (reversi.py:32-36)
    for g, idx, img in zip(fx["img_game"], fx["img_index"], fx["img"]):
        got = O.make_image(fx["hist_white"][g], fx["hist_black"][g], int(idx))
        assert (got == img.astype(np.float32)).all(), (g, idx)
    assert O.make_image(fx["hist_white"][0], fx["hist_black"][0], 0).sum() == 64  # only the turn plane


def test_bulk_reference_verdict_is_committed_and_clean(golden_dir):
    """>= 100k games played by the IMPORTED reference (tests/golden/make_golden.py bulk, build container only)
    against this oracle's digests of all 61 states; only the verdict travels."""
    import json
    v = json.load(open(os.path.join(golden_dir, "bulk_env_verdict.json")))
    assert v["games"] >= 100_000 and v["steps"] == 60 * v["games"]
    assert v["mismatched_digests"] == v["mismatched_actions"] == v["mismatched_done"] == v["mismatched_last_reward"] == 0
    assert v["games_with_2plus_passes"] > 1000 and v["pass_as_60th_step_done_stays_0"] > 1000 and v["voluntary_pass_games"] > 1000
