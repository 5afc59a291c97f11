"""GPU parity of the device replay window and batch builder (mz_replay_save / mz_replay_sample_batch)
through the C ABI: the reference fixture (tests/golden/replay.npz: FIFO eviction, short games, five
recorded sample_batch calls, also under the reference's own `random` stream), and a large randomised
comparison with the C oracle."""
import os
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mz_oracle as O  # noqa: E402  (checker only)


def step_major(G, ids, dev="cuda"):
    """trajectory tensors as the self-play engine records them: (T, n[, 65])"""
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    return dict(hist_white=t(G["hist_white"][ids].view(np.int64).T), hist_black=t(G["hist_black"][ids].view(np.int64).T),
                actions=t(G["actions"][ids].T), rewards=t(G["rewards"][ids].T), visits=t(G["visits"][ids].transpose(1, 0, 2)),
                root_values=t(G["root_values"][ids].T))


def check_batches(rb, G, replay_random):
    from muzero_amd import config
    for b in range(int(G["n_batches"])):
        pre = "b%d_" % b
        K = int(G[pre + "K"])
        if replay_random:  # the reference's own draws from the same seed
            config.num_unroll_steps = K
            random.seed(int(G[pre + "seed"]))
            out = rb.sample_batch(K)
            config.num_unroll_steps = 3
        else:
            out = rb.sample_batch(K, G[pre + "game_idx"].tolist(), G[pre + "pos"].tolist())
        image, acts, tv, tp = (x.cpu().numpy() for x in out)
        assert (image == G[pre + "image"].astype(np.float32)).all(), b
        assert (acts == G[pre + "actions"].astype(np.float32)).all(), b
        assert (tv.view(np.uint32) == G[pre + "target_values"].view(np.uint32)).all(), b
        assert (tp.view(np.uint32) == G[pre + "target_policies"].view(np.uint32)).all(), b


@pytest.mark.parametrize("chunks", [[1] * 8, [3, 2, 3], [4, 4], [2, 6]])
def test_reference_fixture(golden_dir, chunks):
    """8 games saved in batches of various sizes into a window of 6 (eviction mid-batch included)"""
    from muzero_amd.replay import ReplayBuffer
    G = np.load(os.path.join(golden_dir, "replay.npz"))
    rb = ReplayBuffer(window_size=int(G["window_size"]), batch_size=int(G["batch_size"]))
    g0 = 0
    for c in chunks:
        ids = list(range(g0, g0 + c))
        assert rb.ready() == (g0 >= 4)
        rb.save_games(step_major(G, ids), lengths=G["length"][ids].tolist())
        g0 += c
    assert rb.buffer_size == 6 and rb.ready()
    assert rb.count() == int(G["count"]) and rb.count() == 0
    assert [rb.game_length(i) for i in range(6)] == G["length"][2:].tolist()
    check_batches(rb, G, False)
    check_batches(rb, G, True)
    # stored Game fields
    sl = [(rb._head + i) % 6 for i in range(6)]
    assert (rb.root_values[sl].cpu().numpy() == G["root_values"][2:]).all()
    assert (rb.rewards[sl].cpu().numpy() == G["rewards"][2:]).all()


def test_save_game_objects(golden_dir):
    """ReplayBuffer.save_game with reference-shaped Game objects (fraction child_visits, plane observations)"""
    from muzero_amd.replay import ReplayBuffer
    G = np.load(os.path.join(golden_dir, "replay.npz"))

    class A:
        def __init__(self, i):
            self.index = i

    class Game:
        pass

    rb = ReplayBuffer(window_size=6, batch_size=4)
    for i in range(8):
        n = int(G["length"][i])
        g = Game()
        bits = lambda x: np.unpackbits(np.array([x], "<u8").view(np.uint8), bitorder="little").reshape(8, 8).astype(bool)  # noqa: E731
        g.state_history = [np.stack((bits(G["hist_white"][i, t]), bits(G["hist_black"][i, t]))) for t in range(n + 1)]
        g.action_history = [A(int(a)) for a in G["actions"][i, :n]]
        g.rewards = G["rewards"][i, :n].tolist()
        g.child_visits = [O.child_visits(G["visits"][i, t]).tolist() for t in range(n)]
        g.root_values = [torch.tensor([[v]]) for v in G["root_values"][i, :n]]
        rb.save_game(g)
    check_batches(rb, G, False)


def test_large_random_vs_oracle():
    """4096 synthetic games (random lengths, actions, visit counts, rewards) in a window of 3000 -> 8192 samples, K=5"""
    from muzero_amd.replay import ReplayBuffer
    rng = np.random.default_rng(11)
    n, W, K, B = 4096, 3000, 5, 8192
    G = dict(hist_white=rng.integers(0, 2 ** 63, (n, 61), dtype=np.int64).view(np.uint64),
             hist_black=rng.integers(0, 2 ** 63, (n, 61), dtype=np.int64).view(np.uint64),
             actions=rng.integers(0, 65, (n, 60)).astype(np.uint8), rewards=rng.integers(-1, 2, (n, 60)).astype(np.int8),
             visits=rng.integers(-1, 40, (n, 60, 65)).astype(np.int32), root_values=rng.random((n, 60)).astype(np.float32))
    G["visits"][:, :, 7] = 50  # at least one visited child everywhere
    length = rng.integers(K, 61, n).astype(np.int32)
    rb = ReplayBuffer(window_size=W, batch_size=B)
    for g0 in range(0, n, 1024):
        ids = list(range(g0, g0 + 1024))
        rb.save_games(step_major(G, ids), lengths=length[ids].tolist())
    assert rb.buffer_size == W
    idx = rng.integers(0, W, B)
    pos = np.array([rng.integers(0, length[n - W + i] - K + 1) for i in idx])
    pos[:64] = length[n - W + idx[:64]] - K  # the zero-padded tail step (reversi.py:336-339)
    out = rb.sample_batch(K, idx.tolist(), pos.tolist())
    image, acts, tv, tp = (x.cpu().numpy() for x in out)
    games = [dict(hist_white=G["hist_white"][i], hist_black=G["hist_black"][i], actions=G["actions"][i], rewards=G["rewards"][i],
                  visits=G["visits"][i], length=int(length[i])) for i in range(n - W, n)]
    oi, oa, ov, op = O.replay_batch(games, list(zip(idx.tolist(), pos.tolist())), K)
    assert (image == oi).all() and (acts == oa).all()
    assert (tv.view(np.uint32) == ov.view(np.uint32)).all() and (tp.view(np.uint32) == op.view(np.uint32)).all()
    with pytest.raises(Exception):
        rb.sample_batch(K, [0], [int(length[n - W]) - K + 1])


def test_selfplay_records_feed_the_buffer():
    """engine trajectories -> buffer -> batch: targets equal the oracle's on the engine's own records"""
    from hashnet_torch import HashNetTorch
    from muzero_amd.replay import ReplayBuffer
    from muzero_amd.selfplay import SelfPlayEngine
    from test_gpu_selfplay import HashServe
    eng = SelfPlayEngine(64, HashServe(), num_simulations=8, use_graph=False, seed=3)
    rec = eng.play_games()
    rb = ReplayBuffer(window_size=64, batch_size=32)
    rb.save_games(rec)
    assert rb.ready() and rb.count() == 64 * 60
    random.seed(5)
    idx = random.sample(range(64), 32)
    pos = [random.randint(0, 57) for _ in idx]
    out = rb.sample_batch(3, idx, pos)
    C = lambda t: t.cpu().numpy()  # noqa: E731
    games = [dict(hist_white=C(rec.hist_white[:, i]).view(np.uint64), hist_black=C(rec.hist_black[:, i]).view(np.uint64),
                  actions=C(rec.actions[:, i]), rewards=C(rec.rewards[:, i]), visits=C(rec.visits[:, i]), length=60) for i in range(64)]
    want = O.replay_batch(games, list(zip(idx, pos)), 3)
    for a, b in zip(out, want):
        assert (C(a).view(np.uint32) == b.view(np.uint32)).all()
    assert HashNetTorch is not None


@pytest.mark.parametrize("dt", [torch.uint8, torch.int16])
def test_narrow_visit_counts_and_strided_views(golden_dir, dt):
    """trajectory records with uint8 / int16 visit counts (0 for non-children, as GameRecords keeps them) build the very
    same batches as the int32 (-1) form; and a trimmed VIEW of a wider gather buffer is saved without a re-layout"""
    from muzero_amd.replay import ReplayBuffer
    G = np.load(os.path.join(golden_dir, "replay.npz"))
    rb = ReplayBuffer(window_size=int(G["window_size"]), batch_size=int(G["batch_size"]))
    for ids in ([0, 1, 2], [3, 4, 5, 6, 7]):
        part = step_major(G, ids)
        part["visits"] = part["visits"].clamp_min(0).to(dt)
        if len(ids) == 5:   # as shard.gather_game_records hands a short rank's shard over: a (T, 5) view of a (T, 9) buffer
            wide = {k: torch.zeros((v.shape[0], 9) + tuple(v.shape[2:]), dtype=v.dtype, device="cuda") for k, v in part.items()}
            for k, v in part.items():
                wide[k][:, :5] = v
            part = {k: v[:, :5] for k, v in wide.items()}
            assert not part["visits"].is_contiguous()
        rb.save_games(part, lengths=G["length"][ids].tolist())
    check_batches(rb, G, False)
    with pytest.raises(Exception):
        bad = step_major(G, [0])
        bad["visits"] = bad["visits"].float()
        rb.save_games(bad, lengths=[60])


def test_refused_samples_are_zero_filled_and_counted():
    """a (slot, position) pair outside its game reaching the kernel (a direct C caller, or a host/device length mismatch): the
    sample's rows come back as zeros, never as whatever the output buffers held, and rb->refused counts it"""
    from muzero_amd import _lib as L
    from muzero_amd.replay import ReplayBuffer
    rng = np.random.default_rng(3)
    n, K = 8, 3
    G = dict(hist_white=rng.integers(0, 2 ** 63, (n, 61), dtype=np.int64).view(np.uint64),
             hist_black=rng.integers(0, 2 ** 63, (n, 61), dtype=np.int64).view(np.uint64),
             actions=rng.integers(0, 65, (n, 60)).astype(np.uint8), rewards=rng.integers(-1, 2, (n, 60)).astype(np.int8),
             visits=rng.integers(1, 40, (n, 60, 65)).astype(np.int32), root_values=rng.random((n, 60)).astype(np.float32))
    rb = ReplayBuffer(window_size=n, batch_size=4)
    rb.save_games(step_major(G, list(range(n))), lengths=[20] * n)
    slot = torch.tensor([0, n, -1, 1, 2, 3], dtype=torch.int32, device="cuda")        # 1, 2: slot outside the window
    pos = torch.tensor([5, 0, 0, 18, -1, 17], dtype=torch.int32, device="cuda")       # 3: pos + K > length; 4: pos < 0; 0 and 5 fine
    B = 6
    image = torch.full((B, 9, 8, 8), 7.0, device="cuda")
    acts = torch.full((K, B, 2, 8, 8), 7.0, device="cuda")
    tv = torch.full((K + 1, B), 7.0, device="cuda")
    tp = torch.full((K + 1, B, 65), 7.0, device="cuda")
    L.call("mz_replay_sample_batch", L.C.byref(rb._rb), L.dptr(slot), L.dptr(pos), B, K, L.dptr(image), L.dptr(acts), L.dptr(tv), L.dptr(tp),
           L.current_stream())
    assert rb.refused_samples() == 4
    for b in (1, 2, 3, 4):
        assert float(image[b].abs().sum()) == 0 and float(acts[:, b].abs().sum()) == 0
        assert float(tv[:, b].abs().sum()) == 0 and float(tp[:, b].abs().sum()) == 0
    ref = rb.sample_batch(K, [0, 3], [5, 17])
    for got, want, sel in ((image, ref[0], None), (acts, ref[1], 1), (tv, ref[2], 1), (tp, ref[3], 1)):
        g = got[[0, 5]] if sel is None else got[:, [0, 5]]
        assert torch.equal(g, want)
    assert not bool((image == 7.0).any() | (acts == 7.0).any() | (tv == 7.0).any() | (tp == 7.0).any())
