"""N>1 path on CPU: world_size-2 gloo processes.  Checks that sharding by game id is invisible in
the results (each rank's block of oracle playouts equals the matching slice of a single run --
the playout RNG is keyed by game id exactly as on the GPUs), that trajectory records gather in
game-id order with uneven shards, and the max-over-ranks timing reduction."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_game_block_partitions():
    from muzero_amd.shard import game_block
    for total, world in ((65536, 8), (4096, 3), (7, 8), (0, 2)):
        blocks = [game_block(total, r, world) for r in range(world)]
        assert blocks[0][0] == 0 and sum(n for _, n in blocks) == total
        for (g0, n), (g1, _) in zip(blocks, blocks[1:]):
            assert g0 + n == g1
        assert max(n for _, n in blocks) - min(n for _, n in blocks) <= 1
    with pytest.raises(ValueError):
        game_block(10, 2, 2)


def _worker(rank, world, port, total, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from muzero_amd.shard import game_block, gather_game_records, max_over_ranks
    from oracle import mz_oracle as O
    g0, n = game_block(total, rank, world)
    e, digest, actions, last_reward = O.random_playout(1261, g0, n, 60, 0, want_actions=True)

    class Rec:
        pass

    rec = Rec()
    rec.hist_white = torch.from_numpy(np.tile(e.white.view(np.int64)[None], (61, 1)))
    rec.hist_black = torch.from_numpy(np.tile(e.black.view(np.int64)[None], (61, 1)))
    rec.actions = torch.from_numpy(actions.T.copy())
    rec.rewards = torch.from_numpy(np.tile(last_reward[None], (60, 1)))
    rec.visits = torch.from_numpy(np.tile(digest.view(np.int64)[None, :, None] % 251, (60, 1, 65)).astype(np.uint8))   # uint8 counts, as the engine records them
    rec.root_values = torch.full((60, n), float(rank))
    from muzero_amd.shard import record_bytes_per_game
    assert record_bytes_per_game(rec) <= 8 * 1024   # 4-8 KB per game on the wire
    parts = gather_game_records(rec)     # one {field: (T, n_r, ...)} per rank, in game-id order
    assert len(parts) == world and [p["actions"].shape[1] for p in parts] == [game_block(total, r, world)[1] for r in range(world)]
    full = {k: torch.cat([p[k] for p in parts], 1) for k in parts[0]}
    again = gather_game_records(rec)     # the receive buffers are reused
    assert again[0]["visits"].data_ptr() == parts[0]["visits"].data_ptr()
    t = max_over_ranks(1.0 + rank)
    dist.barrier()
    if rank == 0:
        np.savez(os.path.join(out_dir, "gathered.npz"), t=t, **{k: v.numpy() for k, v in full.items()})
    dist.destroy_process_group()


def test_two_rank_sharding_and_gather(tmp_path):
    from oracle import mz_oracle as O
    total, world = 301, 2  # uneven shards: 151 + 150
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(world, port, total, str(tmp_path)), nprocs=world, join=True)
    got = np.load(os.path.join(str(tmp_path), "gathered.npz"))
    e, digest, actions, last_reward = O.random_playout(1261, 0, total, 60, 0, want_actions=True)
    assert got["actions"].shape == (60, total)
    assert (got["actions"] == actions.T).all()
    assert (got["hist_white"][0].view(np.uint64) == e.white).all() and (got["hist_black"][60].view(np.uint64) == e.black).all()
    assert (got["rewards"][0] == last_reward).all()
    assert got["visits"].dtype == np.uint8 and (got["visits"][0, :, 0] == (digest.view(np.int64) % 251).astype(np.uint8)).all()
    assert (got["root_values"][0, :151] == 0).all() and (got["root_values"][0, 151:] == 1).all()
    assert float(got["t"]) == 2.0
