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
