"""Multi-GPU execution: one process per GPU, profiles sharded in contiguous ranges, no data exchange inside
an iteration.  The only collective is the all-reduce of the convergence counters (active, converged,
invalid, singular), a 32-byte message that is latency-bound; it runs over RCCL (torch.distributed backend
"nccl") on GPUs and over gloo in the CPU tests."""
from __future__ import annotations

import numpy as np


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT if a launcher set them.
    Returns (rank, world_size, local_rank)."""
    import os
    import torch
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', str(rank)))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def allreduce_counts(counts):
    """Sum the per-shard counters (a torch int64 tensor of 4 entries, on the GPU for nccl) over all ranks."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return counts


def allreduce_max(value):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(value, op=dist.ReduceOp.MAX)
    return value


def run_sharded_until_converged(step, read_counts, max_iter, check_every=1):
    """Generic sharded Picard loop: `step()` launches one iteration on the local shard, `read_counts()` returns
    the local (active, converged, invalid, singular) as a torch tensor; every rank leaves the loop on the same
    iteration, when no profile anywhere is active or max_iter is reached.  Returns (iterations, global counts)."""
    done = 0
    total = None
    while done < max_iter:
        k = min(check_every, max_iter - done)
        for _ in range(k):
            step()
        done += k
        total = allreduce_counts(read_counts())
        if int(total[0]) == 0:
            break
    return done, total
