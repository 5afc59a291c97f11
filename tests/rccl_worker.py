"""Worker of tests/test_gpu_rccl.py: every collective of the N > 1 path on REAL RCCL (backend "nccl"), one process per GPU.
Launched by torch.distributed.run; a one-GPU box gives a single-rank communicator, which still runs each call through
ProcessGroupNCCL / RCCL (type checks, stream handling, the device_id binding) -- what the gloo tests cannot show."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", device_id=dev)       # as bench.py and muzero_amd.loop do
    from muzero_amd import train as T
    from muzero_amd.network import Network
    from muzero_amd.shard import gather_game_records, max_over_ranks, record_bytes_per_game
    dist.barrier()
    assert max_over_ranks(1.0 + rank, dev) == float(world)

    class Rec:
        pass

    n, Tm = 8 + rank, 60
    g = torch.Generator(device="cuda").manual_seed(5 + rank)
    for vis_dtype in (torch.uint8, torch.int16, torch.int32):   # <= 255 simulations / <= 32767 / more (selfplay.GameRecords)
        rec = Rec()
        rec.hist_white = torch.randint(-2**62, 2**62, (Tm + 1, n), device=dev, generator=g)
        rec.hist_black = torch.randint(-2**62, 2**62, (Tm + 1, n), device=dev, generator=g)
        rec.actions = torch.randint(0, 65, (Tm, n), device=dev, generator=g).to(torch.uint8)
        rec.rewards = torch.randint(-1, 2, (Tm, n), device=dev, generator=g).to(torch.int8)
        rec.visits = torch.randint(0, 120, (Tm, n, 65), device=dev, generator=g).to(vis_dtype)
        rec.root_values = torch.rand((Tm, n), device=dev, generator=g)
        parts = gather_game_records(rec)
        assert len(parts) == world
        mine = parts[rank]
        for k in ("hist_white", "hist_black", "actions", "rewards", "visits", "root_values"):
            assert mine[k].dtype == getattr(rec, k).dtype and torch.equal(mine[k], getattr(rec, k)), (k, vis_dtype)
        if vis_dtype == torch.uint8:
            assert record_bytes_per_game(rec) <= 8 * 1024

    # the training step's ONE flat-bucket all-reduce, weighted by the local batch share, and the weight broadcast
    torch.manual_seed(1261)
    net = Network().to(dev)
    net.train()
    T.broadcast_weights(net, src=0)
    opt = T.make_optimizer(net)
    bucket = T.FlatGradBucket(net.parameters())
    B, K = 8, 3
    batch = (torch.rand((B, 9, 8, 8), device=dev, generator=g).round(),
             torch.rand((K, B, 2, 8, 8), device=dev, generator=g).round(),
             torch.rand((K + 1, B), device=dev, generator=g) * 2 - 1,
             torch.softmax(torch.rand((K + 1, B, 65), device=dev, generator=g), -1))
    log, T.LOG = T.LOG, None
    try:
        p, v = T.update_weights(opt, net, batch, bucket, loss_scale=1.0)
        p2, v2 = T.update_weights(opt, net, None, bucket, loss_scale=0.0)     # a rank whose share of the global draw is empty
    finally:
        T.LOG = log
    assert torch.isfinite(torch.as_tensor([float(p), float(v)])).all()
    before = bucket.flat.clone()
    dist.all_reduce(bucket.flat, op=dist.ReduceOp.SUM)   # the bucket's collective itself (update_weights skips it in a world of one)
    if world == 1:
        assert torch.equal(bucket.flat, before)
    sums = torch.stack([q.detach().double().sum() for q in net.parameters()])
    ref = sums.clone()
    dist.broadcast(ref, src=0)
    assert torch.equal(sums, ref)        # replicas identical after the reduced step
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("RCCL_OK world=%d" % world, flush=True)


if __name__ == "__main__":
    main()
