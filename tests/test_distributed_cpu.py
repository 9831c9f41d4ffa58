"""The multi-GPU control path on CPU: two gloo ranks shard an ensemble, all-reduce the convergence counters
and leave the Picard loop on the same iteration."""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip('torch')
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, nb, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), LOCAL_RANK=str(rank))
    from tango_amd import distributed as D
    from tango_amd.ensemble import shard_range
    r, w, _ = D.init_from_env(backend='gloo')
    lo, hi = shard_range(nb, r, w)
    # synthetic per-profile convergence iteration: profile b converges after 3 + (b % 11) iterations
    need = 3 + (np.arange(lo, hi) % 11)
    state = {'it': 0}

    def step():
        state['it'] += 1

    def counts():
        active = int(np.sum(need > state['it']))
        return torch.tensor([active, (hi - lo) - active, 0, 0], dtype=torch.int64)

    done, total = D.run_sharded_until_converged(step, counts, max_iter=50, check_every=1)
    mx = D.allreduce_max(torch.tensor([float(need.max())], dtype=torch.float64))
    out.put((r, lo, hi, done, [int(v) for v in total], float(mx[0])))
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_agree_on_termination():
    world, nb = 2, 37
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nb, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, d0, t0, m0), (r1, lo1, hi1, d1, t1, m1) = res
    assert lo0 == 0 and hi0 == lo1 and hi1 == nb
    assert d0 == d1 == 13                # slowest profile anywhere: 3 + 10
    assert t0 == t1 == [0, nb, 0, 0]
    assert m0 == m1 == 13.0
