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
