"""Multi-GPU execution: one process per GPU, profiles sharded in contiguous ranges, no data exchange inside
an iteration.  The only collectives are the all-reduce of the convergence counters (active, converged,
invalid, singular: 32 bytes) and of the residual summary (sum of squared residuals, entry count, profiles
still iterating | largest rms, any invalid: 40 bytes), both latency-bound; they run over RCCL
(torch.distributed backend "nccl") on GPUs and over gloo in the CPU tests."""
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
        import datetime
        # a rank that has died must not keep the others waiting for half an hour in a 40-byte all-reduce
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=int(os.environ.get('TANGO_COLLECTIVE_TIMEOUT', '300'))))
    return rank, world, local


def _all_reduce(t, op):
    """In-place all-reduce of a small tensor.  RCCL (backend nccl) reduces device tensors on the current stream; gloo
    is handed a host copy (the call then waits for the stream: gloo is the rehearsal and test backend, and its own
    handling of device tensors stalled now and then with two ranks sharing a GPU)."""
    import torch.distributed as dist
    if t.is_cuda and dist.get_backend() != 'nccl':
        h = t.cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)


def allreduce_counts(counts):
    """Sum the per-shard counters (a torch int64 tensor of 4 entries, on the GPU for nccl) over all ranks."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        _all_reduce(counts, dist.ReduceOp.SUM)
    return counts


def allreduce_residual(summary):
    """Combine the per-shard residual summaries of tg_residual_reduce (a torch float64 tensor of 5 entries: sum of
    squared residuals, number of entries, profiles still iterating | largest rms, any invalid) over all ranks: the
    first three add up, the last two take the maximum.  Two collectives of 24 and 16 bytes, in place."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        _all_reduce(summary[:3], dist.ReduceOp.SUM)
        _all_reduce(summary[3:], dist.ReduceOp.MAX)
    return summary


def allreduce_max(value):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        _all_reduce(value, dist.ReduceOp.MAX)
    return value


class ConvergenceMonitor:
    """Convergence checks of a sharded ensemble that never make the host wait for the GPU.

    issue() queues, on the ensemble's stream: the device reductions of the shard (tg_status_reduce,
    tg_residual_reduce), the two all-reduces over the ranks and an asynchronous copy of the nine reduced numbers to
    pinned host memory.  previous() returns the numbers of the LAST issued check once the caller has queued more
    work behind it (by then the copy has landed long ago), as a dict, or None when no check is outstanding.  Every
    rank sees the same reduced numbers, so every rank takes the same decision on the same check.

    `ensemble` provides reduce_status(stream, counts_ptr) and reduce_residual(out_ptr, stream).  `stream` is a
    torch.cuda.Stream, the one the ensemble's steps are launched on; everything issue() queues (the library's
    reductions, torch's all-reduces, conversions and copies) goes to that stream, whatever torch's current stream is
    when it is called."""

    def __init__(self, ensemble, device, stream):
        import torch
        self.E, self.stream = ensemble, stream
        self.counts = torch.zeros(4, dtype=torch.int64, device=device)
        self.summary = torch.zeros(5, dtype=torch.float64, device=device)
        self.host = torch.zeros(9, dtype=torch.float64).pin_memory()
        self.landed = torch.cuda.Event()
        self.pending = False
        self.issued = self.consumed = 0

    def issue(self):
        import torch
        with torch.cuda.stream(self.stream):     # torch's collectives and copies follow its current stream
            self.E.reduce_status(self.stream.cuda_stream, self.counts.data_ptr())
            self.E.reduce_residual(self.summary.data_ptr(), self.stream.cuda_stream)
            allreduce_counts(self.counts)
            allreduce_residual(self.summary)
            self.host[:4].copy_(self.counts.to(torch.float64), non_blocking=True)
            self.host[4:].copy_(self.summary, non_blocking=True)
            self.landed.record(self.stream)
        self.pending = True
        self.issued += 1

    def previous(self):
        if not self.pending:
            return None
        self.landed.synchronize()
        self.pending = False
        self.consumed += 1
        h = self.host.numpy()
        return {'active': h[0], 'converged': h[1], 'invalid': h[2], 'singular': h[3], 'sum_sq': h[4], 'entries': h[5],
                'active_residual': h[6], 'max_rms': h[7], 'any_invalid': h[8],
                'rms_of_active': float(np.sqrt(h[4] / h[5])) if h[5] else None}


def run_sharded_until_converged(step, read_counts, max_iter, check_every=1, lagged=False):
    """Generic sharded Picard loop: `step()` launches one iteration on the local shard, `read_counts()` returns
    the local (active, converged, invalid, singular) as a torch tensor; every rank leaves the loop on the same
    iteration, when no profile anywhere is active or max_iter is reached.  Returns (iterations, global counts).

    lagged=False reads every check's result at once (the host waits for the GPU at each check; the loop stops at the
    first check that finds nothing active).  lagged=True looks at a check's result only after the next check_every
    iterations have been queued, so the host never waits; the loop then runs check_every iterations longer, which
    changes nothing: a profile that has converged is no longer stepped."""
    done = 0
    total = previous = None
    while done < max_iter:
        k = min(check_every, max_iter - done)
        for _ in range(k):
            step()
        done += k
        if lagged and previous is not None and int(previous[0]) == 0:
            total = previous
            break
        total = allreduce_counts(read_counts())
        if lagged:
            previous = total
        elif int(total[0]) == 0:
            break
    return done, total
