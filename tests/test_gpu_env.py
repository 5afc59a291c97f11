"""GPU parity of the HIP env kernels (through the C ABI) against the C oracle and the committed
reference fixtures.  Bit-exact: boards, legality, steps, done, rewards."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import mz_oracle as O  # noqa: E402  (checker only)


def dev(a):
    if a.dtype == np.uint64:
        a = a.view(np.int64)
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host_u64(t):
    return t.cpu().numpy().view(np.uint64)


@pytest.fixture(scope="module")
def menv():
    from muzero_amd import env
    return env


@pytest.fixture(scope="module")
def env_fx(golden_dir):
    return np.load(os.path.join(golden_dir, "env_traces.npz"))


def assert_state(e, o):
    assert (host_u64(e.black) == o.black).all() and (host_u64(e.white) == o.white).all()
    assert (e.to_play.cpu().numpy() == o.to_play).all() and (e.steps.cpu().numpy() == o.steps).all()
    assert (e.done.cpu().numpy() == o.done).all() and (host_u64(e.legal) == o.legal).all()


def test_reset_and_empty_batch(menv):
    e = menv.BatchedEnv(5)
    assert_state(e, O.EnvBatch(5))
    z = menv.BatchedEnv(0)  # empty batch: every call is a no-op
    z.step(torch.zeros(0, dtype=torch.uint8, device="cuda"))
    z.random_playout(1)


@pytest.mark.parametrize("prefix", ["full_", "volfull_"])
def test_step_by_step_vs_reference_fixture(menv, env_fx, prefix):
    A = env_fx[prefix + "actions"]
    n = A.shape[0]
    e = menv.BatchedEnv(n)
    for t in range(60):
        reward, status = e.step(dev(A[:, t]))
        assert (status.cpu().numpy() == 0).all()
        assert (host_u64(e.white) == env_fx[prefix + "white"][:, t + 1]).all(), t
        assert (host_u64(e.black) == env_fx[prefix + "black"][:, t + 1]).all(), t
        assert (host_u64(e.legal) == env_fx[prefix + "legal"][:, t + 1]).all(), t
        assert (e.to_play.cpu().numpy() == env_fx[prefix + "to_play"][:, t + 1]).all(), t
        assert (e.steps.cpu().numpy() == env_fx[prefix + "steps"][:, t + 1]).all(), t
        assert (e.done.cpu().numpy() == env_fx[prefix + "done"][:, t + 1]).all(), t
        assert (reward.cpu().numpy() == env_fx[prefix + "rewards"][:, t]).all(), t
    # finished games are flagged no-ops
    w0 = e.white.clone()
    reward, status = e.step(torch.full((n,), 64, dtype=torch.uint8, device="cuda"))
    d = e.done.cpu().numpy() == 1
    assert (status.cpu().numpy()[d] == 2).all() and (e.white == w0).cpu().numpy()[d].all()


@pytest.mark.parametrize("prefix,seed,vol,n", [("dig_", 1261, 0, 4096), ("vol_", 1262, 1, 512)])
def test_fused_playout_vs_reference_fixture(menv, env_fx, prefix, seed, vol, n):
    e = menv.BatchedEnv(n)
    digest, actions, last_reward = e.random_playout(seed, 0, 60, vol, want_digest=True, want_actions=True)
    assert (actions.cpu().numpy() == env_fx[prefix + "actions"]).all()
    assert (host_u64(digest) == env_fx[prefix + "digest"]).all()
    assert (host_u64(e.white) == env_fx[prefix + "final_white"]).all()
    assert (host_u64(e.black) == env_fx[prefix + "final_black"]).all()
    assert (e.to_play.cpu().numpy() == env_fx[prefix + "final_to_play"]).all()
    assert (e.done.cpu().numpy() == env_fx[prefix + "final_done"]).all()
    assert (last_reward.cpu().numpy() == env_fx[prefix + "rewards_last"]).all()


def oracle_playout_threads(seed, g0, n, vol=0, threads=16):
    """the oracle's playout over n games, split over host threads (ctypes releases the GIL)"""
    from concurrent.futures import ThreadPoolExecutor
    chunk = (n + threads - 1) // threads
    parts = [(g, min(chunk, n - g)) for g in range(0, n, chunk)]
    with ThreadPoolExecutor(threads) as ex:
        return list(ex.map(lambda p: (p, O.random_playout(seed, g0 + p[0], p[1], 60, vol)), parts))


@pytest.mark.parametrize("vol", [0, 1])
def test_playout_vs_oracle_1M_games(menv, vol):
    """BASELINE configs[1] at full size against the ORACLE: 1,048,576 fresh games (seed/offset not in any fixture),
    digests of all 61 states of every game, last rewards and every field of the final states"""
    n, seed, g0 = 1 << 20, 987654321 + vol, 5_000_000
    e = menv.BatchedEnv(n)
    digest, _, last_reward = e.random_playout(seed, g0, 60, vol, want_digest=True)
    digest, last_reward = host_u64(digest), last_reward.cpu().numpy()
    st = {k: getattr(e, k).cpu().numpy() for k in ("black", "white", "legal", "to_play", "steps", "done")}
    for (a, m), (o, odig, _, orw) in oracle_playout_threads(seed, g0, n, vol):
        sl = slice(a, a + m)
        assert (digest[sl] == odig).all() and (last_reward[sl] == orw).all()
        for k in ("black", "white", "legal"):
            assert (st[k][sl].view(np.uint64) == getattr(o, k)).all(), k
        for k in ("to_play", "steps", "done"):
            assert (st[k][sl] == getattr(o, k)).all(), k


def test_step_kernel_path_equals_fused_path_1M(menv):
    """config 2 at full size (1,048,576 games): the HBM-resident step kernel driven by
    mz_env_random_action must land on exactly the states of the fused in-register playout;
    plus size-independent invariants of the rules."""
    n, seed = 1 << 20, 1261
    a = menv.BatchedEnv(n)
    a.random_playout(seed, 0, 60, 0)
    b = menv.BatchedEnv(n)
    act = torch.empty(n, dtype=torch.uint8, device="cuda")
    placed = torch.zeros(n, dtype=torch.int32, device="cuda")
    for t in range(60):
        b.random_action(seed, 0, t, out=act)
        _, status = b.step(act)
        placed += (act != 64).int()
    assert int(status.max()) == 0
    for x, y in ((a.black, b.black), (a.white, b.white), (a.legal, b.legal), (a.to_play, b.to_play), (a.steps, b.steps), (a.done, b.done)):
        assert bool((x == y).all())
    # invariants: no overlap; discs = 4 + placements; 60 steps each; done <=> 60th step was a placement
    assert int((a.black & a.white).ne(0).sum()) == 0
    pop = lambda t: torch.from_numpy(np.array([bin(int(v)).count("1") for v in host_u64(t[:4096])]))  # noqa: E731
    assert (pop(a.black) + pop(a.white) == 4 + placed[:4096].cpu()).all()
    assert bool((a.steps == 60).all())
    assert bool((a.done == (act != 64).to(torch.uint8)).all())


def test_legal_and_step_on_arbitrary_positions(menv):
    """random (mostly unreachable) positions: Kogge-Stone fills vs literal ray walks"""
    rng = np.random.default_rng(7)
    n = 200000
    occ = rng.integers(0, 2**63, n, dtype=np.uint64) | (rng.integers(0, 2, n, dtype=np.uint64) << np.uint64(63))
    dens = rng.integers(0, 4, n)
    for k in range(3):  # vary density
        m = rng.integers(0, 2**63, n, dtype=np.uint64) | (rng.integers(0, 2, n, dtype=np.uint64) << np.uint64(63))
        occ = np.where(dens > k, occ & m, occ)
    col = rng.integers(0, 2**63, n, dtype=np.uint64) | (rng.integers(0, 2, n, dtype=np.uint64) << np.uint64(63))
    black, white = occ & col, occ & ~col
    to_play = rng.choice(np.array([-1, 1], np.int8), n)
    e = menv.BatchedEnv(n)
    e.black.copy_(dev(black)); e.white.copy_(dev(white)); e.to_play.copy_(dev(to_play))
    legal = e.legal_mask()
    olegal = O.legal_mask(black, white, to_play)
    assert (host_u64(legal) == olegal).all()
    # one step from those positions with arbitrary actions (legal, illegal, pass, out of range)
    steps = rng.integers(0, 61, n).astype(np.uint8)
    action = rng.integers(0, 67, n).astype(np.uint8)
    o = O.EnvBatch(n)
    o.black[:], o.white[:], o.to_play[:], o.steps[:], o.legal[:] = black, white, to_play, steps, olegal
    e.steps.copy_(dev(steps)); e.legal.copy_(legal)
    reward, status = e.step(dev(action))
    orw, ost = o.step(action)
    assert (status.cpu().numpy() == ost).all()
    assert (reward.cpu().numpy() == orw).all()
    assert_state(e, o)
    assert (ost == 0).sum() > 1000 and (ost == 1).sum() > 1000


def test_features_vs_oracle_and_fixture(menv, golden_dir):
    fx = np.load(os.path.join(golden_dir, "features.npz"))
    # Action.encode incl. the row/column quirk and the whole-plane pass
    acts = np.arange(65, dtype=np.uint8).repeat(2)
    pl = np.tile(np.array([1, -1], np.int8), 65)
    for dt in (torch.float32, torch.bfloat16, torch.float16):
        got = menv.action_encode(dev(acts), dev(pl), dt).float().cpu().numpy()
        for k in range(len(acts)):
            assert (got[k] == O.action_encode(int(acts[k]), int(pl[k]))).all(), (k, dt)
    for idx, p, planes in zip(fx["enc_index"], fx["enc_player"], fx["enc_planes"]):
        got = menv.action_encode(dev(np.array([idx], np.uint8)), dev(np.array([p], np.int8))).cpu().numpy()[0]
        assert (got == planes).all()
    # make_image: step-major history of the 8 fixture games, every index 0..60, per-game and uniform index
    hw, hb = fx["hist_white"], fx["hist_black"]          # (8, 61)
    g = hw.shape[0]
    hwd, hbd = dev(hw.T.copy()), dev(hb.T.copy())        # (61, 8)
    for idx in range(61):
        img = menv.make_image(hwd, hbd, idx, g).cpu().numpy()
        for k in range(g):
            assert (img[k] == O.make_image(hw[k], hb[k], idx)).all(), (k, idx)
    per = np.array([0, 1, 2, 3, 4, 5, 33, 60], np.int32)
    for dt in (torch.float32, torch.bfloat16):
        img = menv.make_image(hwd, hbd, dev(per), g, dt).float().cpu().numpy()
        for k in range(g):
            assert (img[k] == O.make_image(hw[k], hb[k], int(per[k]))).all()
    for gi, idx, want in zip(fx["img_game"], fx["img_index"], fx["img"]):
        img = menv.make_image(hwd, hbd, int(idx), g).cpu().numpy()[gi]
        assert (img == want).all()


def test_packed_step_equals_step_and_oracle(menv):
    """mz_env_step_packed (to_play / steps / done / action / reward / status in one u32 word per game) against mz_env_step on
    1,048,576 random games step by step, and against the oracle on arbitrary positions with arbitrary actions (legal,
    illegal, pass, out of range, finished games)"""
    n, seed = 1 << 20, 99
    a = menv.BatchedEnv(n)
    b = menv.BatchedEnv(n)
    act = torch.empty(n, dtype=torch.uint8, device="cuda")
    for t in range(60):
        a.random_action(seed, 0, t, out=act)
        if t % 7 == 3:
            act[::5] = 64                 # voluntary passes (reversi.py:135-139 accepts them)
        ra, sa = a.step(act)
        rb, sb = b.step_packed(b.pack_meta(act))
        assert torch.equal(ra, rb) and torch.equal(sa, sb), t
    for x, y in ((a.black, b.black), (a.white, b.white), (a.legal, b.legal), (a.to_play, b.to_play), (a.steps, b.steps), (a.done, b.done)):
        assert bool((x == y).all())
    # arbitrary positions / actions vs the oracle (the refused cases leave the state untouched)
    rng = np.random.default_rng(8)
    n = 100000
    occ = rng.integers(0, 2**63, n, dtype=np.uint64) & rng.integers(0, 2**63, n, dtype=np.uint64)
    col = rng.integers(0, 2**63, n, dtype=np.uint64)
    black, white = occ & col, occ & ~col
    to_play = rng.choice(np.array([-1, 1], np.int8), n)
    steps = rng.integers(0, 61, n).astype(np.uint8)
    done = (rng.integers(0, 8, n) == 0).astype(np.uint8)
    action = rng.integers(0, 67, n).astype(np.uint8)
    olegal = O.legal_mask(black, white, to_play)
    o = O.EnvBatch(n)
    o.black[:], o.white[:], o.to_play[:], o.steps[:], o.done[:], o.legal[:] = black, white, to_play, steps, done, olegal
    e = menv.BatchedEnv(n)
    e.black.copy_(dev(black)); e.white.copy_(dev(white)); e.to_play.copy_(dev(to_play)); e.steps.copy_(dev(steps))
    e.done.copy_(dev(done)); e.legal.copy_(dev(olegal.view(np.int64)))
    reward, status = e.step_packed(e.pack_meta(dev(action)))
    orw, ost = o.step(action)
    assert (status.cpu().numpy() == ost).all() and (reward.cpu().numpy() == orw).all()
    assert_state(e, o)
    assert (ost == 0).sum() > 1000 and (ost == 1).sum() > 1000 and (ost == 2).sum() > 1000
