"""
Multi-GPU layer: trajectories (the leading batch axis of every tensor on the path) are independent, so they are
sharded contiguously over ranks and nothing per-time-step is ever communicated.  The only exchange is the sum that the
reference takes over its batch axis (`tf.reduce_sum`, kalman_filter.py:255; variational_cvi.py:402): one all-reduce of
the scalar ELBO (plus any per-step scalars / small gradient vectors) over RCCL ("nccl" backend on ROCm) or gloo on CPU.
"""
import os

import torch
import torch.distributed as dist


def shard_bounds(total, rank, world):
    """Contiguous, balanced shard [lo, hi) of `total` trajectories for `rank` (first `total % world` ranks get one more)."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of size {world}")
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run).  Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            local = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, **kw)
    return rank, world


def allreduce_sum_(t):
    """In-place sum over ranks (no-op without a process group): the ELBO / gradient reduction."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def allreduce_max_(t):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t
