"""Data-parallel plumbing of the training step: one process per GPU and the single exchange the path has -- the sum of the
flat gradient vector (2,589,093 floats, 10.4 MB) over the ranks (SURVEY.md section 8 e).  Under the "nccl" backend (RCCL on
ROCm) the exchange is libsr_hip's own ncclAllReduce on its own communicator (include/sr_hip.h sr_rccl_*, sr_train_allreduce),
enqueued on the library's stream in front of the Adam step; torch.distributed only launches the ranks, carries the 128-byte
communicator id from rank 0 and broadcasts the initial weights.  SR_DP_DIRECT=0 sends the exchange through torch's all_reduce
instead, and the "gloo" backend (CPU rehearsal of several ranks on one GPU) stages it through the host.

The job's batch is the union of the ranks' equal-sized batches; each rank's gradient is that of the mean
loss over ITS batch, so the gradient of the mean over the global batch is the sum over ranks divided by
the world size -- which is what ``mean_over_ranks_`` leaves in place on every rank.
"""
import os


def world():
    """(rank, world_size, local_rank) from the launcher's environment (torchrun / bench.py contract)."""
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')),
            int(os.environ.get('LOCAL_RANK', '0')))


def active():
    """Is the data-parallel path on?  WORLD_SIZE > 1, or SR_DP_FORCE=1: a ONE-rank process group, which sends the gradient
    through the same RCCL all-reduce (the only way to execute that path on a box with a single GPU)."""
    return world()[1] > 1 or os.environ.get('SR_DP_FORCE') == '1'


def init_if_needed(backend=None):
    """Join the process group when the data-parallel path is on; returns the world size."""
    rank, n, _ = world()
    if not active():
        return 1
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        dist.init_process_group(backend or os.environ.get('SR_DIST_BACKEND', 'nccl'), rank=rank, world_size=n)
    return n


_comm = None


def direct():
    """Is the exchange on libsr_hip's own RCCL communicator?"""
    import torch.distributed as dist
    return (dist.is_available() and dist.is_initialized() and dist.get_backend() == 'nccl'
            and os.environ.get('SR_DP_DIRECT', '1') != '0')


def comm(device):
    """The process's RcclComm over all ranks, made on first use (collective: every rank calls it at the same point)."""
    global _comm
    if _comm is None:
        import torch
        import torch.distributed as dist
        import sr_hip
        rank, n = dist.get_rank(), dist.get_world_size()
        ident = sr_hip.RcclComm.unique_id() if rank == 0 else bytes(128)
        box = torch.tensor(list(ident), dtype=torch.uint8, device=f'cuda:{device}')
        dist.broadcast(box, src=0)
        _comm = sr_hip.RcclComm(bytes(box.cpu().tolist()), rank, n, device)
        import atexit
        atexit.register(shutdown)
    return _comm


def shutdown():
    """Destroy the library's communicator (before torch.distributed.destroy_process_group)."""
    global _comm
    if _comm is not None:
        _comm.close()
        _comm = None


def mean_over_ranks_(tensor):
    """In-place all-reduce(sum) / world size; identity for a single process."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return tensor
    dist.all_reduce(tensor, op=dist.ReduceOp.SUM)
    tensor /= dist.get_world_size()
    return tensor


def broadcast_weights_(tensors, device):
    """Every rank starts from rank 0's weights (a fresh model is initialised randomly per process).
    tensors: [(name, float32 array)]; returns the list with rank 0's values."""
    import numpy as np
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return tensors
    dev = f'cuda:{device}' if dist.get_backend() == 'nccl' else 'cpu'
    flat = torch.from_numpy(np.concatenate([a.ravel() for _, a in tensors]).astype(np.float32)).to(dev)
    dist.broadcast(flat, src=0)
    flat = flat.cpu().numpy()
    out, o = [], 0
    for name, a in tensors:
        out.append((name, flat[o:o + a.size].reshape(a.shape).copy()))
        o += a.size
    return out


class GradBuffer:
    """The flat gradient vector as a torch tensor on the rank's GPU (allocated once)."""

    def __init__(self, n, device, n_losses=3):
        import torch
        self.device = int(device)
        self.t = torch.empty(int(n), dtype=torch.float32, device=f'cuda:{device}')
        self.losses = torch.empty(int(n_losses), dtype=torch.float32, device=f'cuda:{device}')
        self.grad_scale = 1.0       # what the Adam step multiplies the reduced buffer by

    def ptr(self):
        return self.t.data_ptr()

    def reduce(self, losses, ctx=None):
        """Gradient and losses over the ranks; returns the averaged losses as floats.  On the direct path the buffer is left
        holding the SUM and grad_scale = 1 / ranks goes to the Adam step; on the torch paths the buffer holds the mean.
        ctx: the sr_hip.Context whose gradient (all of this buffer) it is -- the all-reduce then runs on that ctx's stream."""
        import torch
        import torch.distributed as dist
        self.losses.copy_(torch.tensor(losses, dtype=torch.float32))
        if direct():
            # the ctx stream is a blocking stream: its work and the null stream's (torch's copies, the second all-reduce)
            # are ordered as issued, and nothing here waits on the host until the losses are read back
            c = comm(self.device)
            if ctx is not None:
                ctx.train_allreduce(c, self.ptr())
            else:
                c.allreduce_f32(self.ptr(), self.t.numel())
            c.allreduce_f32(self.losses.data_ptr(), self.losses.numel())
            self.grad_scale = 1.0 / c.nranks
            return [float(v) / c.nranks for v in self.losses.cpu()]
        self.grad_scale = 1.0
        if dist.is_initialized() and dist.get_backend() == 'gloo':
            # rehearsal without RCCL (several ranks on one GPU, CPU-only collectives): stage through the host
            for t in (self.t, self.losses):
                h = t.cpu()
                mean_over_ranks_(h)
                t.copy_(h)
        else:
            mean_over_ranks_(self.t)
            mean_over_ranks_(self.losses)
        torch.cuda.synchronize(self.t.device)     # libsr_hip's stream is not torch's: finish before Adam reads
        return [float(v) for v in self.losses.cpu()]
