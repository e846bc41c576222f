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
