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


def owner_of_global_game(i, games_per_rank, world):
    """Sharded replay window: every rank keeps only the games it played; the GLOBAL window (what one process would hold)
    lists, per generation, rank 0's games, then rank 1's, ... (game-id order).  Global FIFO index i (0 = oldest) ->
    (owning rank, index in that rank's local FIFO).  All ranks hold the same number of generations."""
    per_gen = int(games_per_rank) * int(world)
    gen, within = divmod(int(i), per_gen)
    rank, j = divmod(within, int(games_per_rank))
    return rank, gen * int(games_per_rank) + j


RECORD_FIELDS = ("hist_white", "hist_black", "actions", "rewards", "visits", "root_values")

_GATHER_BUFFERS = {}   # (field, shape, dtype, device) -> preallocated (world, T, n, ...) receive buffer, reused every generation


def record_bytes_per_game(rec):
    """bytes one game contributes to the trajectory gather (SURVEY 8e budgets 4-8 KB)"""
    n = rec.actions.shape[1]
    return sum(getattr(rec, k).numel() * getattr(rec, k).element_size() for k in RECORD_FIELDS) // n


def gather_game_records(rec, group=None):
    """Bring the step-major trajectory tensors of every rank together: ONE all_gather_into_tensor per field, straight into
    a preallocated (world, T, n, ...) buffer -- no list of parts, no pad, no concatenation copy.

    `rec` is a GameRecords (or any object with RECORD_FIELDS tensors shaped (T, n_local, ...)).  Returns a list with one
    {field: tensor (T, n_r, ...)} per rank, in rank order = game-id order (views of the receive buffers; they stay valid
    until the next gather).  ReplayBuffer.save_games takes them one by one (mz_replay_save reads step-major batches with a
    stride, so no re-layout is needed).  Ranks may hold different n_local (game_block remainders): then every rank pads
    to the longest shard for the collective (one copy on the short ranks only) and the views are trimmed."""
    fields = {k: getattr(rec, k) for k in RECORD_FIELDS}
    if not (dist.is_available() and dist.is_initialized()):
        return [fields]
    world = dist.get_world_size(group)
    dev = fields["actions"].device
    n_local = fields["actions"].shape[1]
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, torch.tensor([n_local], dtype=torch.int64, device=dev), group=group)
    counts = [int(c) for c in counts.tolist()]
    n_max = max(counts)
    out = [dict() for _ in range(world)]
    for k, t in fields.items():
        if n_local < n_max:   # short rank: one cached pad buffer per field, its tail zeroed once (the head is overwritten every time)
            pkey = ("pad", k, t.shape[0], n_max, tuple(t.shape[2:]), t.dtype, t.device)
            pad = _GATHER_BUFFERS.get(pkey)
            if pad is None:
                pad = _GATHER_BUFFERS[pkey] = torch.zeros((t.shape[0], n_max) + tuple(t.shape[2:]), dtype=t.dtype, device=t.device)
            pad[:, :n_local] = t
            pad[:, n_local:] = 0   # (a rank's shard size never changes in a run; cheap insurance if it does)
            t = pad
        t = t.contiguous()
        key = (k, tuple(t.shape), t.dtype, t.device, world)
        buf = _GATHER_BUFFERS.get(key)
        if buf is None:
            buf = _GATHER_BUFFERS[key] = torch.empty((world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        # rank-major along dim 0.  The collective sees BYTES: RCCL has no 16-bit integer type (int16 visit counts, S > 255)
        # and a gather needs none
        dist.all_gather_into_tensor(buf.view(torch.uint8).view(world * t.shape[0], -1), t.view(torch.uint8).view(t.shape[0], -1), group=group)
        for r in range(world):
            out[r][k] = buf[r] if counts[r] == n_max else buf[r][:, :counts[r]]
    return out
