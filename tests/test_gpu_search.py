"""mz_search (one C call per search) against the call-by-call engine path on the same inputs: same kernels, same
launch order, so visits and root values must be bit-identical; plus the per-simulation semantics the engine path is
itself pinned on (tests/test_gpu_mcts.py, test_gpu_selfplay.py: oracle and reference fixtures)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def roots(n, moves, seed=77):
    """n positions after `moves` random moves, with their (61, n) observation history"""
    from muzero_amd.env import BatchedEnv
    e = BatchedEnv(n)
    hw = torch.zeros((61, n), dtype=torch.int64, device="cuda")
    hb = torch.zeros((61, n), dtype=torch.int64, device="cuda")
    hw[0], hb[0] = e.white, e.black
    act = torch.empty(n, dtype=torch.uint8, device="cuda")
    for t in range(moves):
        e.random_action(seed, 0, t, out=act)
        e.step(act)
        hw[t + 1], hb[t + 1] = e.white, e.black
    return e, hw, hb


@pytest.fixture(scope="module")
def served():
    from muzero_amd.network import InferenceEngine, Network
    torch.manual_seed(1261)
    return InferenceEngine(Network(), 64, torch.bfloat16)


@pytest.mark.parametrize("n,S,moves,use_noise", [(64, 20, 0, True), (37, 50, 23, True), (5, 1, 40, False), (16, 0, 7, True), (64, 9, 58, True)])
def test_mz_search_equals_call_by_call_search(served, n, S, moves, use_noise):
    from muzero_amd.env import make_image
    from muzero_amd.mcts import FusedSearch
    from muzero_amd.selfplay import SelfPlayEngine
    e, hw, hb = roots(n, moves)
    image = make_image(hw, hb, moves, n, torch.bfloat16)
    eng = SelfPlayEngine(n, served, num_simulations=max(S, 1), use_graph=False, seed=5)
    eng.env = e
    noise = eng.dirichlet_noise(e.legal) if use_noise else None
    fs = FusedSearch(served, n, S)
    v2, r2 = fs(image, e.to_play, e.legal, noise)
    if S == 0:   # expand_node + noise only (main.py:75-83): every legal move is a child with no visits, the root has no value
        bits = (e.legal.unsqueeze(1) >> torch.arange(64, device="cuda")) & 1
        child = torch.cat([bits, (e.legal == 0).long().unsqueeze(1)], 1).bool()
        assert torch.equal(v2, torch.where(child, 0, -1).to(torch.int32)) and bool((r2 == 0).all())
        return
    visits, root_value = eng.search(image, noise)
    assert torch.equal(visits, v2)
    assert torch.equal(root_value.view(torch.int32), r2.view(torch.int32))
    assert int(v2.clamp(min=0).sum(1).min()) == S and int(v2.clamp(min=0).sum(1).max()) == S


def test_mz_search_rejects_a_short_workspace(served):
    from muzero_amd import _lib as L
    from muzero_amd.env import make_image
    from muzero_amd.mcts import FusedSearch
    e, hw, hb = roots(8, 0)
    fs = FusedSearch(served, 8, 10)
    assert L.lib.mz_search_workspace_bytes(8, 10) == fs.workspace.numel()
    assert L.lib.mz_search_workspace_bytes(0, 10) == -1
    image = make_image(hw, hb, 0, 8, torch.bfloat16)
    with pytest.raises(L.MzError):
        fs(image, e.to_play, e.legal, None, n_sims=11)      # more simulations than the workspace was sized for


def test_bf16_serving_vs_fp32_inference_at_search_level():
    """What bf16 serving (HIP tower) does to the search's OUTPUT, against the reference's fp32 network arithmetic
    (network.py:146-163; here fp32 PyTorch-ROCm inference with an fp32 hidden pool) on the same roots and noise.
    Not a bit-parity claim: the thresholds are what was measured on MI355X with margin, and they guard
    against a serving path that drifts (wrong rounding point, wrong weight image) rather than rounds."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import make_roots
    from muzero_amd.env import make_image
    from muzero_amd.metrics import visit_agreement
    from muzero_amd.network import InferenceEngine, Network
    from muzero_amd.selfplay import SelfPlayEngine
    n, S = 1024, 50
    r, hw, hb, idx = make_roots(n, 1261, 0, torch.device("cuda"))
    out = {}
    noise = None
    for name, dt, tower in (("bf16", torch.bfloat16, "hip"), ("fp32", torch.float32, "miopen")):
        torch.manual_seed(1261)
        eng = SelfPlayEngine(n, InferenceEngine(Network(), n, dt, tower=tower), num_simulations=S, use_graph=False, seed=3)
        eng.env = r
        if noise is None:
            noise = eng.dirichlet_noise(r.legal)
        visits, rv = eng.search(make_image(hw, hb, idx, n, dt), noise)
        out[name] = (visits.clone(), rv.clone())
    m = visit_agreement(out["fp32"][0], out["bf16"][0])
    dv = (out["fp32"][1] - out["bf16"][1]).abs()
    print("bf16 vs fp32 search:", m, "root value |diff| mean %.3g max %.3g" % (float(dv.mean()), float(dv.max())))
    # measured (1024 roots, 50 simulations): top-1 1.000, identical visit vectors 0.985, total variation 2.9e-4, root value
    # |diff| mean 2.9e-5 / max 6.5e-4
    assert m["top1_agreement"] >= 0.98 and m["identical_visit_vectors"] >= 0.97 and m["mean_total_variation"] <= 5e-3
    assert float(dv.mean()) <= 5e-4
