"""One-process-per-GPU plumbing for the data-parallel runs (bench.py, training scripts).

The PointConvFormer operator does not shard inside a scene: every rank owns whole
packed clouds and the only data-path collective is the gradient all-reduce that
DistributedDataParallel issues over RCCL ("nccl" backend on ROCm) -- xGMI links on one node.  This
module holds the small amount of glue around that: rendezvous from the torchrun environment, a
barrier+synchronise fence for timing, max-over-ranks of a wall time and the per-rank data seed.
It runs unchanged on the gloo backend (CPU), which is how the tests cover world_size 2.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def env_rank():
    return int(os.environ.get('RANK', 0)), int(os.environ.get('WORLD_SIZE', 1)), int(os.environ.get('LOCAL_RANK', 0))


def setup(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (no-op for one process).
    Returns (rank, world_size, local_rank, device)."""
    rank, world, local_rank = env_rank()
    # Rehearsal on a one-GPU box: PCF_DIST_REHEARSE=1 puts every rank on cuda:0 and uses gloo for the collectives
    # (RCCL refuses two ranks on one device).  Never set by bench.py or the tests' normal paths.
    rehearse = os.environ.get('PCF_DIST_REHEARSE') == '1'
    if rehearse:
        backend = 'gloo'
    use_gpu = torch.cuda.is_available() and (backend != 'gloo' or rehearse)
    dev = torch.device('cuda', 0 if rehearse else local_rank) if use_gpu else torch.device('cpu')
    if use_gpu:
        torch.cuda.set_device(dev)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        backend = backend or ('nccl' if use_gpu else 'gloo')
        kw = {'device_id': dev} if backend == 'nccl' else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local_rank, dev


def fence(dev=None):
    """Everything enqueued so far has finished on every rank."""
    if torch.cuda.is_available() and (dev is None or dev.type == 'cuda'):
        torch.cuda.synchronize()
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
        if torch.cuda.is_available() and (dev is None or dev.type == 'cuda'):
            torch.cuda.synchronize()


def max_over_ranks(seconds: float, dev) -> float:
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def data_seed(base: int, rank: int) -> int:
    """Each rank draws its own synthetic cloud (weak scaling: per-GPU work is fixed)."""
    return base + rank


def whole_job_rate(units_per_rank_per_step: int, steps: int, world: int, seconds: float) -> float:
    return world * units_per_rank_per_step * steps / seconds


def wrap_ddp(module, dev):
    if dist.is_initialized() and dist.get_world_size() > 1:
        ids = [dev.index] if dev.type == 'cuda' else None
        return torch.nn.parallel.DistributedDataParallel(module, device_ids=ids)
    return module


class GradBucket:
    """The gradient all-reduce of data-parallel training as ONE flat bucket, for steps replayed from a HIP graph.

    DistributedDataParallel's autograd hooks and bucket bookkeeping run on the host every step; the layer benchmark's
    step is ~190 kernels in 1.16 ms, so that cost (and the eager launches it forces) is what limits N > 1.  Here the
    gradients (13.7 k floats for the PCFLayer, 1.93 M for the 10cm-lite model) are packed into one contiguous buffer
    -- `pack()` is capturable: one concatenation, pre-scaled by 1/world -- summed over the ranks with a single
    RCCL all-reduce on the current stream, and `unpack()` copies the averages back into the .grad tensors with one
    multi-tensor kernel.  Gradients: the mean of the rank-local gradients, as DDP.  Buffers (BatchNorm running
    statistics and counters): DDP broadcasts rank 0's before every forward (`broadcast_buffers=True`); here they are
    broadcast at construction (`broadcast_parameters`) and whenever the caller asks (`sync_buffers()`: once per step
    to mirror DDP exactly, or before checkpointing from rank 0) -- the benchmark loops skip the per-step call because
    running statistics do not enter a training-mode forward."""

    def __init__(self, params, buffers=None):
        self.params = [p for p in params if p.requires_grad]
        self.buffers = [b for b in (buffers or []) if b is not None]
        n = sum(p.numel() for p in self.params)
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.flat = torch.zeros(n, dtype=torch.float32, device=self.params[0].device)
        self.views, o = [], 0
        for p in self.params:
            self.views.append(self.flat[o:o + p.numel()].view_as(p))
            o += p.numel()

    def broadcast_parameters(self):
        """Rank 0's parameters and buffers everywhere, as DDP does at construction."""
        if self.world > 1:
            for p in self.params:
                dist.broadcast(p.data, 0)
            self.sync_buffers()

    def sync_buffers(self):
        """Rank 0's buffers everywhere (DDP's broadcast_buffers)."""
        if self.world > 1:
            for b in self.buffers:
                dist.broadcast(b, 0)

    def pack(self):
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in self.params]
        torch.cat([g.reshape(-1) for g in grads], out=self.flat)
        if self.world > 1:
            self.flat.mul_(1.0 / self.world)

    def all_reduce(self):
        if self.world > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)

    def attach(self):
        """The averaged gradients without the copy back: every `.grad` becomes its view of the bucket (contiguous, at a fixed
        address -- what a captured optimizer step needs).  Parameters that had no gradient this step get the zeros `pack`
        wrote for them, as DDP hands the optimizer zeros for unused parameters."""
        for p, v in zip(self.params, self.views):
            p.grad = v

    def unpack(self):
        grads = [p.grad for p in self.params if p.grad is not None]
        views = [v for p, v in zip(self.params, self.views) if p.grad is not None]
        if grads:
            torch._foreach_copy_(grads, views)


def shutdown():
    if dist.is_initialized():
        dist.destroy_process_group()
