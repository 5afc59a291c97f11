"""Sharding of self-play across the GPUs of a node: one process per GPU, games partitioned by id.

Games and trees are independent (reference run_selfplay workers share only the weights,
main.py:59-63), so the data path needs NO collective: rank r plays the contiguous game-id block
``game_block(total, r, world)`` with the same network weights.  Collectives appear only at the
edges and are thin wrappers over torch.distributed (RCCL on the GPUs, gloo in the CPU tests):
``max_over_ranks`` for the bench's timing protocol and ``gather_game_records`` to bring the
fixed-size trajectory records of all ranks together for a replay buffer.
"""
import torch
import torch.distributed as dist


def game_block(total_games, rank, world):
    """(first game id, number of games) of `rank`: contiguous blocks, sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    base, rem = divmod(int(total_games), int(world))
    n = base + (1 if rank < rem else 0)
    g0 = rank * base + min(rank, rem)
    return g0, n


def max_over_ranks(seconds, device="cpu", group=None):
    """the slowest rank's time (bench.py contract); identity without an initialised process group"""
    if not (dist.is_available() and dist.is_initialized()):
        return float(seconds)
    t = torch.tensor([float(seconds)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())


RECORD_FIELDS = ("hist_white", "hist_black", "actions", "rewards", "visits", "root_values")


def gather_game_records(rec, group=None):
    """all_gather the step-major trajectory tensors of every rank along the game axis (dim 1).

    `rec` is a GameRecords (or any object with RECORD_FIELDS tensors shaped (T, n_local, ...)).
    Ranks may hold different n_local (game_block remainders): shorter shards are padded to the
    longest for the collective and trimmed afterwards.  Returns {field: tensor (T, total, ...)} in
    rank order = game-id order."""
    fields = {k: getattr(rec, k) for k in RECORD_FIELDS}
    if not (dist.is_available() and dist.is_initialized()):
        return fields
    world = dist.get_world_size(group)
    n_local = fields["actions"].shape[1]
    counts = [torch.zeros(1, dtype=torch.int64, device=fields["actions"].device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([n_local], dtype=torch.int64, device=fields["actions"].device), group=group)
    counts = [int(c.item()) for c in counts]
    n_max = max(counts)
    out = {}
    for k, t in fields.items():
        if n_local < n_max:
            pad = torch.zeros((t.shape[0], n_max - n_local) + tuple(t.shape[2:]), dtype=t.dtype, device=t.device)
            t = torch.cat([t, pad], 1)
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t.contiguous(), group=group)
        out[k] = torch.cat([p[:, :c] for p, c in zip(parts, counts)], 1)
    return out
