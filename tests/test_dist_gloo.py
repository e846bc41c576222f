"""N > 1 path on CPU: two gloo ranks shard clips contiguously (no data-path collective), time with a
barrier on both sides and reduce MAX over ranks exactly like bench.py does on RCCL."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import utils


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_clips, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = utils.shard_range(n_clips, world, rank)
    mine = list(range(lo, hi))
    dist.barrier()
    dt = torch.tensor([0.1 * (rank + 1)], dtype=torch.float64)      # pretend per-rank elapsed time
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    gathered = [None] * world
    dist.all_gather_object(gathered, {"rank": rank, "clips": mine, "frames": 10 * len(mine)})
    dist.barrier()
    if rank == 0:
        out.put((float(dt.item()), gathered))
    dist.destroy_process_group()


def test_two_rank_clip_sharding_and_max_time():
    world, n_clips = 2, 7
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_clips, q)) for r in range(world)]
    for p in procs:
        p.start()
    dt, gathered = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert abs(dt - 0.2) < 1e-12                                   # MAX over ranks
    clips = [c for g in sorted(gathered, key=lambda g: g["rank"]) for c in g["clips"]]
    assert clips == list(range(n_clips))                           # every clip exactly once, contiguous blocks
    assert gathered[0]["clips"] == [0, 1, 2, 3] and gathered[1]["clips"] == [4, 5, 6]
    assert sum(g["frames"] for g in gathered) == 10 * n_clips      # whole-job aggregate = sum over ranks


# ------------------------------------------------------------------ data-parallel training step
def _dp_worker(rank, world, port, out):
    """Each rank: gradient of the mean loss over ITS batch (the float64 oracle stands in for the device
    gradient), then the one exchange of the training path (dp.mean_over_ranks_, gloo here / RCCL on the box)."""
    import numpy as np
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      SR_DIST_BACKEND="gloo")
    import dp
    from oracle import ref_numpy as rn
    from oracle import ref_train
    assert dp.init_if_needed() == world
    weights = rn.make_weights(seed=7)
    # weights differ per rank until rank 0's are broadcast (a fresh model is random per process)
    mine = [(n, a + np.float32(rank)) for n, a in weights.items()]
    mine = dp.broadcast_weights_(mine, device=0)
    assert all(np.array_equal(a, weights[n]) for n, a in mine)
    rng = np.random.default_rng(11)
    B, h, w = world, 8, 8
    data = [rng.random((B, h, w, 3), dtype=np.float32), rng.random((B, h, w, 3), dtype=np.float32),
            rng.random((B, 4 * h, 4 * w, 3), dtype=np.float32), rng.random((B, 4 * h, 4 * w, 3), dtype=np.float32)]
    shard = [a[rank:rank + 1] for a in data]
    losses, grads, _ = ref_train.loss_and_grads(weights, *shard)
    names = list(grads)
    flat = torch.from_numpy(np.concatenate([grads[n].ravel() for n in names]))
    dp.mean_over_ranks_(flat)
    lt = torch.tensor(losses, dtype=torch.float64)
    dp.mean_over_ranks_(lt)
    if rank == 0:
        want_l, want_g, _ = ref_train.loss_and_grads(weights, *data)      # the global batch in one process
        want = np.concatenate([want_g[n].ravel() for n in names])
        out.put((float(np.abs(flat.numpy() - want).max() / np.abs(want).max()), lt.tolist(), want_l))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_average_is_the_global_batch_gradient():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_dp_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    err, losses, want_l = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert err <= 1e-12                                   # float64: the average of shard gradients IS the batch gradient
    assert all(abs(a - b) <= 1e-12 for a, b in zip(losses, want_l))
