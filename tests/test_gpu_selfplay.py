"""GPU parity of the batched self-play loop (main.play_game composition): whole games on HashNet
must reproduce the reference fixture games and the C oracle's games move for move."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from hashnet_torch import HashNetTorch  # noqa: E402
from oracle import mz_oracle as O  # noqa: E402  (checker only)


class HashServe:
    """HashNetTorch behind the engine's network protocol (initial(image) / recurrent(x10))"""
    dtype = torch.float32

    def __init__(self, mode="hash"):
        self.net = HashNetTorch(mode)
        self.ok = True

    def initial(self, image):
        v, p, h = self.net.initial_inference(image)
        return v.reshape(-1).contiguous(), p.contiguous(), h.contiguous()

    def recurrent(self, x10):
        self.ok = self.ok and self.net.check_hidden(x10[:, :8])
        v, p, h = self.net.recurrent_inference(x10[:, :8], x10[:, 8:])
        return v.reshape(-1).contiguous(), p.contiguous(), h.contiguous()


def test_reference_fixture_games(golden_dir):
    """the 3 games main.play_game produced on HashNet (noise and sampling uniforms replayed)"""
    from muzero_amd.selfplay import SelfPlayEngine
    g = np.load(os.path.join(golden_dir, "selfplay_games.npz"))
    for i, S in enumerate(g["S"]):
        noise = torch.from_numpy(g["noise_%d" % i]).cuda().unsqueeze(1).repeat(1, 2, 1)  # 2 identical games
        u = torch.from_numpy(g["u_%d" % i]).cuda().unsqueeze(1).repeat(1, 2)
        net = HashServe()
        eng = SelfPlayEngine(2, net, num_simulations=int(S), use_graph=False)
        rec = eng.play_games(noise=noise, u=u)
        assert net.ok
        for k in range(2):
            assert (rec.actions[:, k].cpu().numpy() == g["actions_%d" % i]).all(), i
            assert (rec.rewards[:, k].cpu().numpy() == g["rewards_%d" % i]).all()
            assert (rec.child_visits()[:, k].cpu().numpy() == g["child_visits_%d" % i]).all()
            assert rec.root_values[:, k].cpu().numpy().view(np.uint32).tolist() == g["root_values_%d" % i].view(np.uint32).tolist()
        assert (rec.done.cpu().numpy() == g["done"][i]).all()


def test_batch_of_games_vs_oracle():
    """256 different games x 60 moves x 24 sims (supplied noise / uniforms) == C oracle play_game"""
    from muzero_amd.selfplay import SelfPlayEngine
    n, S = 256, 24
    rng = np.random.default_rng(3)
    noise = np.zeros((60, n, 65))
    u = rng.random((60, n))
    # noise must live on the legal set of the position it is used in, which depends on the game so
    # far: draw a full 65-vector of gammas and let both sides normalise over the legal entries
    gam = rng.gamma(0.03, size=(60, n, 65)) + 1e-300
    net = HashServe()
    eng = SelfPlayEngine(n, net, num_simulations=S, use_graph=False)

    class Replay:
        t = 0

    def noise_for(legal):
        bits = (legal.unsqueeze(1) >> torch.arange(64, device="cuda", dtype=torch.int64)) & 1
        mask = torch.cat([bits, (legal == 0).long().unsqueeze(1)], 1).double()
        gz = torch.from_numpy(gam[Replay.t]).cuda() * mask
        Replay.t += 1
        nz = (gz / gz.sum(1, keepdim=True)).contiguous()
        noise[Replay.t - 1] = nz.cpu().numpy()
        return nz

    eng.dirichlet_noise = noise_for
    rec = eng.play_games(u=torch.from_numpy(u).cuda())
    assert net.ok
    acts, rews = rec.actions.cpu().numpy(), rec.rewards.cpu().numpy()
    vis, rv = rec.visits.cpu().numpy(), rec.root_values.cpu().numpy()
    for k in range(n):
        m, oa, orw, ov, orv, hw, hb = O.play_game_hashnet(S, 0, np.ascontiguousarray(noise[:, k]), np.ascontiguousarray(u[:, k]))
        assert m == 60
        assert (acts[:, k] == oa).all(), k
        assert (rews[:, k] == orw).all()
        assert vis.dtype == np.uint8 and (vis[:, k] == np.maximum(ov, 0)).all()   # records keep 0 for non-children
        assert rv[:, k].view(np.uint32).tolist() == orv.view(np.uint32).tolist()
        assert (rec.hist_white[:, k].cpu().numpy().view(np.uint64) == hw).all()


def test_graph_replay_equals_eager():
    """the hipGraph-captured simulation step (network + fused tree kernel) must give exactly the
    eager result.  HashNet (integer ops) is the network here: MIOpen may pick different conv
    algorithms under capture, so a live conv net is not bit-reproducible across the two paths (H6)."""
    from muzero_amd.selfplay import SelfPlayEngine

    class Serve(HashServe):
        def recurrent(self, x10):  # no host-synchronising check inside a capture
            v, p, h = self.net.recurrent_inference(x10[:, :8], x10[:, 8:])
            return v.reshape(-1).contiguous(), p.contiguous(), h.contiguous()

    out = []
    for use_graph in (False, True):
        eng = SelfPlayEngine(512, Serve(), num_simulations=40, use_graph=use_graph, seed=5)
        img = torch.zeros(512, 9, 8, 8, device="cuda")
        img[:, 8] = 1
        noise = eng.dirichlet_noise(eng.env.legal)
        visits, rv = eng.search(img, noise)
        visits2, rv2 = eng.search(img, noise)  # second search re-uses the captured graph
        assert (visits == visits2).all() and (rv == rv2).all()
        out.append((visits.cpu(), rv.cpu(), noise.cpu()))
    assert (out[0][2] == out[1][2]).all()
    assert (out[0][0] == out[1][0]).all() and (out[0][1] == out[1][1]).all()
    assert int(out[0][0].clamp(min=0).sum(1).min()) == 40


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_real_network_search_is_sane(dtype):
    """the real MuZero network through InferenceEngine + hipGraph: every tree completes S
    simulations, values are tanh-bounded, no pool error"""
    from muzero_amd import mcts as MC
    from muzero_amd.network import InferenceEngine, Network
    from muzero_amd.selfplay import SelfPlayEngine
    torch.manual_seed(1261)
    net = InferenceEngine(Network(), 64, dtype)
    eng = SelfPlayEngine(64, net, num_simulations=12, use_graph=True, seed=5)
    img = torch.zeros(64, 9, 8, 8, device="cuda", dtype=dtype)
    img[:, 8] = 1
    visits, rv = eng.search(img)
    assert int(visits.clamp(min=0).sum(1).min()) == 12 and int(visits.clamp(min=0).sum(1).max()) == 12
    assert bool((rv.abs() <= 1).all())
    assert int(eng.mcts.meta()[:, MC.META_ERROR].max()) == 0
    assert int(eng.mcts.meta()[:, MC.META_N_NODES].min()) == 13


def test_fp32_grade_engine_plays_the_fp32_games():
    """whole self-play games (play_game, main.py:69-92) served by the split-precision tower kernel (InferenceEngine(tower=
    "hip_x3"): ~1e-5 of fp32) against the same games served by PyTorch/MIOpen fp32, same seeds: a game follows the fp32 game move
    for move unless a root's visit counts differ somewhere -- measured on MI355X: every one of 128 games identical"""
    from muzero_amd.network import InferenceEngine, Network
    from muzero_amd.selfplay import SelfPlayEngine
    recs = []
    for kw in (dict(fused=False, tower="miopen"), dict(channels_last=False, tower="hip_x3")):
        torch.manual_seed(1261)
        net = InferenceEngine(Network(), 128, torch.float32, **kw)
        eng = SelfPlayEngine(128, net, num_simulations=16, use_graph=False, seed=11)
        rec = eng.play_games()
        recs.append((rec.length, rec.actions[:rec.length].cpu(), rec.visits[:rec.length].cpu(), rec.root_values[:rec.length].cpu()))
    assert recs[0][0] == recs[1][0]
    same_game = (recs[0][1] == recs[1][1]).all(0)
    print("games identical move for move: %d of 128; root visit vectors identical: %.4f" %
          (int(same_game.sum()), float((recs[0][2] == recs[1][2]).all(-1).float().mean())))
    assert int(same_game.sum()) >= 120
    g = same_game.nonzero().flatten()
    assert (recs[0][3][:, g] - recs[1][3][:, g]).abs().max().item() <= 1e-4   # root values of the identical games


def test_device_dirichlet_noise_has_the_reference_law():
    """mz_search_cfg.device_noise: np.random.dirichlet([alpha] * n_children) (main.py:192-193) drawn inside k_root_init.
    With root_exploration_fraction = 1 the root priors ARE the noise: on the legal actions only, summing to 1, with the
    moments and the peakedness of numpy's Dirichlet(0.03); a new search draws anew, the same seed and count reproduce."""
    from muzero_amd import mcts as MC

    class Cfg:
        pass

    from muzero_amd import config
    for k in dir(config):
        if not k.startswith("_"):
            setattr(Cfg, k, getattr(config, k))
    Cfg.root_exploration_fraction = 1.0
    n = 16384
    rng = np.random.default_rng(5)
    e, _, _ = _random_roots(n, rng)
    legal = torch.from_numpy(e.legal.view(np.int64)).cuda()
    to_play = torch.from_numpy(e.to_play).cuda()
    policy = torch.full((n, 65), 1.0 / 65, device="cuda")
    hidden = torch.zeros((n, 8, 8, 8), device="cuda")

    def draw(m, seed):
        m.root_init(to_play, legal, policy, hidden, None, device_noise_seed=seed)
        return m.root_stats(with_debug=True)[3].cpu().numpy()

    m = MC.BatchedMCTS(n, 2, torch.float32, cfg=Cfg)
    a = draw(m, 77)
    b = draw(m, 77)                       # the slot's search count moved on: other draws
    m2 = MC.BatchedMCTS(n, 2, torch.float32, cfg=Cfg)
    a2 = draw(m2, 77)                     # same seed, same count: the same draws
    c = draw(MC.BatchedMCTS(n, 2, torch.float32, cfg=Cfg), 78)
    assert (a == a2).all() and (a != b).any() and (a != c).any()
    bits = ((e.legal[:, None] >> np.arange(64, dtype=np.uint64)) & 1).astype(bool)
    mask = np.concatenate([bits, (e.legal == 0)[:, None]], 1)
    assert (a[~mask] == 0).all() and np.abs(a.sum(1) - 1).max() < 1e-12 and (a >= 0).all()
    # law: compare with numpy's Dirichlet per number of children k (every k with enough roots)
    kk = mask.sum(1)
    checked = 0
    for k in np.unique(kk):
        rows = np.nonzero(kk == k)[0]
        if len(rows) < 600 or k < 2:
            continue
        x = np.concatenate([a[rows][mask[rows]].reshape(len(rows), k), b[rows][mask[rows]].reshape(len(rows), k)])
        ref = np.random.default_rng(1).dirichlet([0.03] * k, 200000)
        a0 = 0.03 * k
        var = 0.03 * (a0 - 0.03) / (a0 * a0 * (a0 + 1))
        se = np.sqrt(var / x.size)
        assert abs(x.mean() - 1.0 / k) < 1e-9                     # rows sum to 1
        assert abs(x[:, 0].mean() - 1.0 / k) < 5 * np.sqrt(var / len(x)), k
        assert abs(x.var() - var) < 0.02, (k, x.var(), var)
        for thr in (0.5, 0.9, 0.999):
            p_got, p_ref = (x.max(1) > thr).mean(), (ref.max(1) > thr).mean()
            assert abs(p_got - p_ref) < 5 * np.sqrt(p_ref * (1 - p_ref) / len(x)) + 0.01, (k, thr, p_got, p_ref)
        checked += 1
        del se
    assert checked >= 3
    one = np.nonzero(kk == 1)[0]          # a single child (a forced move or the pass): the sample is exactly [1]
    if len(one):
        assert (a[one][mask[one]] == 1.0).all()


def _random_roots(n, rng):
    """positions after 0..59 random moves (oracle playouts), as host arrays"""
    from oracle import mz_oracle as O
    steps = rng.integers(0, 60, n)
    out = O.EnvBatch(n)
    for s in np.unique(steps):
        idx = np.nonzero(steps == s)[0]
        e = O.random_playout(4242, int(idx[0]), len(idx), int(s))[0]
        for f in ("black", "white", "to_play", "steps", "done", "legal"):
            getattr(out, f)[idx] = getattr(e, f)
    return out, None, None


def test_whole_search_graph_equals_the_kernel_by_kernel_search():
    """the search as ONE hipGraph (initial inference, root expansion, S x [inference + tree kernel], root statistics) against
    the same launches issued one by one: identical visits and root values with caller-supplied noise and with none; with the
    engine's own noise (drawn inside the root expansion) every tree completes S simulations and consecutive searches differ"""
    from muzero_amd import mcts as MC
    from muzero_amd.env import make_image
    from muzero_amd.network import InferenceEngine, Network
    from muzero_amd.selfplay import SelfPlayEngine
    from bench import make_roots
    torch.manual_seed(1261)
    n, S = 256, 24
    served = InferenceEngine(Network(), n, torch.bfloat16)
    r, hw, hb, idx = make_roots(n, 1261, 0, torch.device("cuda"))
    image = make_image(hw, hb, idx, n, torch.bfloat16)
    eager = SelfPlayEngine(n, served, num_simulations=S, use_graph=False, seed=3)
    graph = SelfPlayEngine(n, served, num_simulations=S, use_graph=True, seed=3)
    eager.env = graph.env = r
    noise = eager.dirichlet_noise(r.legal)
    for nz in (noise, None, noise):
        want = eager.search(image, nz)
        got = graph.search(image, nz)
        assert torch.equal(got[0], want[0]) and torch.equal(got[1].view(torch.int32), want[1].view(torch.int32))
    assert set(graph._search_graphs) == {"given", "none"}
    v1, _ = graph.search(image)            # "auto": device noise
    p1 = graph.mcts.root_stats(with_debug=True)[3].clone()
    v2, _ = graph.search(image)
    p2 = graph.mcts.root_stats(with_debug=True)[3]
    assert int(v1.clamp(min=0).sum(1).min()) == S and int(v2.clamp(min=0).sum(1).max()) == S
    assert bool((p1 != p2).any()) and int(graph.mcts.meta()[:, MC.META_ERROR].max()) == 0
    assert abs(float(p1.sum(1).mean()) - 1.0) < 1e-9
