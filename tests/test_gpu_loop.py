"""GPU: the whole loop (main.py:25-65) in one process -- self-play, replay window, training with the
flat gradient bucket, in-place weight publish under a captured hipGraph."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_two_generations():
    from muzero_amd import config
    from muzero_amd.loop import MuZeroLoop
    old = config.num_unroll_steps
    config.num_unroll_steps = 5   # BASELINE config 5
    try:
        loop = MuZeroLoop(64, num_simulations=6, window_size=128, batch_size=32, infer_dtype=torch.bfloat16)
        w0 = [p.detach().clone() for p in loop.infer.net.parameters()]
        l1 = loop.generation(3)
        assert len(l1) == 3 and all(np.isfinite(float(p)) and np.isfinite(float(v)) for p, v in l1)
        assert loop.replay.buffer_size == 64 and loop.replay.count() == 64 * 60
        # published: the serving copy equals the trained weights (bf16-rounded) and differs from the initial ones
        for a, b, c in zip(loop.infer.net.state_dict().values(), loop.network.state_dict().values(), w0):
            assert torch.equal(a, b.to(a.dtype))
        assert any(not torch.equal(a, c) for a, c in zip(loop.infer.net.parameters(), w0))
        l2 = loop.generation(2)   # replays the captured graph against the updated weights
        assert len(l2) == 2 and loop.replay.buffer_size == 128
        loop.generation(1)
        assert loop.replay.buffer_size == 128   # FIFO window full: oldest 64 evicted
        assert loop.network.steps == 6
    finally:
        config.num_unroll_steps = old


def test_generations_with_hip_training_convolutions():
    """train_convs="hip": the same loop with the towers' convolutions and their gradients on the hand-written kernels; the
    first generation plays the same games (same initial weights) and draws the same batches; its losses follow the fp32 loop's
    to bf16 rounding, step by step"""
    from muzero_amd import config
    from muzero_amd.loop import MuZeroLoop
    old = config.num_unroll_steps
    config.num_unroll_steps = 5
    try:
        out = {}
        for mode in ("fp32", "hip"):
            loop = MuZeroLoop(64, num_simulations=6, window_size=64, batch_size=32, infer_dtype=torch.bfloat16, train_convs=mode)
            out[mode] = [(float(p), float(v)) for p, v in loop.generation(6)]
            assert loop.network.steps == 6
            for a, b in zip(loop.infer.net.state_dict().values(), loop.network.state_dict().values()):
                assert torch.equal(a, b.to(a.dtype))
        a, b = np.array(out["fp32"]), np.array(out["hip"])
        assert np.isfinite(b).all()
        assert np.allclose(a[0], b[0], rtol=2e-3)          # same weights, same batch: forward in bf16 vs fp32
        assert np.allclose(a, b, rtol=5e-2)                 # six SGD steps apart
    finally:
        config.num_unroll_steps = old


def test_generation_at_fp32_grade():
    """infer_tower="hip_x3" + train_convs="hip_x3": self-play served by the split-precision tower kernel (fp32 hidden-state pool)
    and the training step on the split-precision kernels -- the reference's fp32 arithmetic to ~1e-5 at every step; the first
    generation's losses follow the fp32 loop's (bf16 self-play plays other games, so the fp32 MIOpen engine is the partner)"""
    from muzero_amd import config
    from muzero_amd.loop import MuZeroLoop
    old = config.num_unroll_steps
    config.num_unroll_steps = 5
    try:
        out = {}
        for mode, kw in (("miopen", dict(infer_tower="miopen", train_convs="fp32")), ("x3", dict(infer_tower="hip_x3", train_convs="hip_x3"))):
            loop = MuZeroLoop(64, num_simulations=6, window_size=64, batch_size=32, infer_dtype=torch.float32, **kw)
            out[mode] = [(float(p), float(v)) for p, v in loop.generation(4)]
            assert loop.replay.buffer_size == 64 and loop.network.steps == 4
            for a, b in zip(loop.infer.net.state_dict().values(), loop.network.state_dict().values()):
                assert a.dtype == torch.float32 and torch.equal(a, b)
        a, b = np.array(out["miopen"]), np.array(out["x3"])
        assert np.isfinite(b).all()
        assert np.allclose(a, b, rtol=2e-2), (a, b)
    finally:
        config.num_unroll_steps = old


def test_sharded_window_equals_replicated_on_one_rank():
    """replay_mode="sharded" (no trajectory gather, local batches weighted in the all-reduce) makes the same draws and the
    same updates as the replicated window; with one rank the two must agree exactly"""
    from muzero_amd import config
    from muzero_amd.loop import MuZeroLoop
    old = config.num_unroll_steps
    config.num_unroll_steps = 3
    try:
        runs = []
        for mode in ("replicated", "sharded"):
            loop = MuZeroLoop(32, num_simulations=4, window_size=64, batch_size=16, infer_dtype=torch.bfloat16, replay_mode=mode, use_graph=False)
            losses = loop.generation(2) + loop.generation(2) + loop.generation(1)   # the third generation evicts the first
            runs.append(([float(p) for p, _ in losses], [p.detach().clone() for p in loop.network.parameters()]))
        # same games, same draws, same arithmetic; MIOpen's backward may sum in a different order from run to run
        assert len(runs[0][0]) == 5 and np.allclose(runs[0][0], runs[1][0], rtol=1e-4)
        for a, b in zip(runs[0][1], runs[1][1]):
            assert torch.allclose(a, b, rtol=1e-3, atol=1e-6)
    finally:
        config.num_unroll_steps = old


def test_overlapped_generation_equals_the_same_dataflow_in_series():
    """generation(overlap=True) -- training on a second stream while the games are played, as the reference's actors and
    trainer run at the same time (main.py:32-37) -- against the same dataflow executed one after the other
    (order="train_first": train on the window as it stood, play, save, publish).  Generations 1 and 2 play with the initial
    weights in both forms, so their games must be IDENTICAL and the training steps that ran beside generation 2 saw identical
    data; from generation 3 on the games are played with trained weights, and MIOpen's weight-gradient kernels do not sum in
    a fixed order (a last-bit difference can flip a bf16 rounding of a served weight), so only the losses are compared."""
    from muzero_amd import config
    from muzero_amd.loop import MuZeroLoop
    old = config.num_unroll_steps
    config.num_unroll_steps = 3
    try:
        runs = []
        for overlap in (False, True):
            loop = MuZeroLoop(64, num_simulations=6, window_size=128, batch_size=32, infer_dtype=torch.bfloat16)
            kw = dict(overlap=True) if overlap else dict(order="train_first")
            assert loop.generation(4, **kw) == []        # window empty: nothing to train on yet
            l2 = loop.generation(4, **kw)
            torch.cuda.synchronize()
            games12 = loop.replay.actions.clone()
            # published at the boundary, after the last training step: the serving copy is the trained weights, bf16-rounded
            for a, b in zip(loop.infer.net.state_dict().values(), loop.network.state_dict().values()):
                assert torch.equal(a, b.to(a.dtype))
            l3 = loop.generation(7, **kw)
            l4 = loop.generation(3, **kw)
            torch.cuda.synchronize()
            runs.append(dict(l2=[(float(p), float(v)) for p, v in l2], l3=[(float(p), float(v)) for p, v in l3],
                             l4=[(float(p), float(v)) for p, v in l4], games12=games12, steps=loop.network.steps,
                             size=loop.replay.buffer_size))
        a, b = runs
        assert a["steps"] == b["steps"] == 14 and a["size"] == b["size"] == 128
        assert len(a["l2"]) == 4 and len(a["l3"]) == 7 and len(a["l4"]) == 3
        assert torch.equal(a["games12"], b["games12"])
        assert np.allclose(a["l2"], b["l2"], rtol=1e-4) and np.allclose(a["l3"], b["l3"], rtol=1e-3)
        assert np.allclose(a["l4"], b["l4"], rtol=5e-2)
    finally:
        config.num_unroll_steps = old
