"""Data-parallel plumbing of the training step: one process per GPU, torch.distributed over RCCL
(backend "nccl" on ROCm) for the single exchange the path has -- the sum of the flat gradient vector
(2,589,093 floats, 10.4 MB) over the ranks (SURVEY.md section 8 e).  PyTorch is plumbing only: the
gradient is produced and consumed by libsr_hip.so; the torch tensor is the buffer RCCL reduces.

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
        self.t = torch.empty(int(n), dtype=torch.float32, device=f'cuda:{device}')
        self.losses = torch.empty(int(n_losses), dtype=torch.float32, device=f'cuda:{device}')

    def ptr(self):
        return self.t.data_ptr()

    def reduce(self, losses):
        """Sum-average gradient and losses over the ranks; returns the averaged losses as floats."""
        import torch
        import torch.distributed as dist
        self.losses.copy_(torch.tensor(losses, dtype=torch.float32))
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
