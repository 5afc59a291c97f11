"""GPU parity of the batched arena (test.py) through the C ABI: the reference fixture games move for
move, the temperature sampler on the reference's draws, and a batch of games vs the C oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mz_oracle as O  # noqa: E402  (checker only)
from test_gpu_selfplay import HashServe  # noqa: E402

MODES = {0: "hash", 1: "uniform", 2: "sat", 3: "const"}


def test_softmax_sample_reference_draws(golden_dir):
    from muzero_amd import arena
    G = np.load(os.path.join(golden_dir, "arena.npz"))
    for t in (0.0, 0.25, 0.5, 1.0):
        m = G["ss_t"] == t
        got = arena.softmax_sample(torch.from_numpy(G["ss_visits"][m]).cuda(), t, torch.from_numpy(G["ss_u"][m]).cuda())
        assert (got.cpu().numpy() == G["ss_action"][m]).all(), t
    with pytest.raises(Exception):
        arena.softmax_sample(torch.zeros((1, 65), dtype=torch.int32, device="cuda"), 2.0, torch.zeros(1, dtype=torch.float64, device="cuda"))


def test_reference_fixture_games(golden_dir):
    from muzero_amd import arena
    G = np.load(os.path.join(golden_dir, "arena.npz"))
    for g in range(int(G["n_games"])):
        black, white = HashServe(MODES[int(G["mode_black_%d" % g])]), HashServe(MODES[int(G["mode_white_%d" % g])])
        u = torch.from_numpy(G["u_%d" % g]).cuda().unsqueeze(1).repeat(1, 2)
        r = arena.play_games(black, white, 2, u=u, use_graph=False)
        assert black.ok and white.ok
        for k in range(2):
            assert (r.actions[:, k].cpu().numpy() == G["actions_%d" % g]).all(), g
        assert r.white_score.tolist() == [int(G["white_score_%d" % g])] * 2 and r.black_score.tolist() == [int(G["black_score_%d" % g])] * 2


def test_controller_image_vs_oracle():
    from muzero_amd import arena
    rng = np.random.default_rng(2)
    n = 64
    hw = rng.integers(0, 2 ** 63, (61, n), dtype=np.int64)
    hb = rng.integers(0, 2 ** 63, (61, n), dtype=np.int64)
    idx = rng.integers(0, 61, n).astype(np.int32)
    idx[:6] = [0, 1, 2, 3, 4, 60]
    got = arena.controller_image(torch.from_numpy(hw).cuda(), torch.from_numpy(hb).cuda(), torch.from_numpy(idx).cuda(), n).cpu().numpy()
    for i in range(n):
        want = O.controller_image(hw[:, i].view(np.uint64), hb[:, i].view(np.uint64), int(idx[i]), idx[i] % 2 == 0)
        assert (got[i] == want).all(), i


def test_match_vs_oracle():
    """test(base, target) with 48+48 games of 30 simulations: every game equals the oracle's"""
    from muzero_amd import arena
    n, S = 48, 30
    rng = np.random.default_rng(9)
    u = rng.random((60, n))
    base, target = HashServe("hash"), HashServe("sat")
    wins, r1, r2 = arena.test(base, target, n, num_simulations=S, u=torch.from_numpy(u).cuda(), use_graph=False)
    want_wins = 0
    for r, (mb, mw, tgt_white) in ((r1, (0, 2, True)), (r2, (2, 0, False))):
        acts = r.actions.cpu().numpy()
        for k in range(n):
            m, oa, ws, bs = O.arena_game_hashnet(mb, mw, S, 0.25, np.ascontiguousarray(u[:, k]))
            assert m == 60 and (acts[:, k] == oa).all(), k
            assert int(r.white_score[k]) == ws and int(r.black_score[k]) == bs
            want_wins += int(ws > bs) if tgt_white else int(bs > ws)
    assert wins == want_wins
