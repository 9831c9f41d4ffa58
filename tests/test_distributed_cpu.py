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
    # the same loop with the verdict of a check read one check later (no host wait): one interval longer
    state['it'] = 0
    done_lag, total_lag = D.run_sharded_until_converged(step, counts, max_iter=50, check_every=2, lagged=True)
    # residual summary: sums add up over the shards, maxima are taken
    part = torch.tensor([float(hi - lo), 512.0 * (hi - lo), float(r + 1), 0.25 * (r + 1), float(r == 1)], dtype=torch.float64)
    summ = D.allreduce_residual(part.clone())
    out.put((r, lo, hi, done, [int(v) for v in total], float(mx[0]), done_lag, [int(v) for v in total_lag], [float(v) for v in summ]))
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
    (r0, lo0, hi0, d0, t0, m0, dl0, tl0, s0), (r1, lo1, hi1, d1, t1, m1, dl1, tl1, s1) = res
    assert lo0 == 0 and hi0 == lo1 and hi1 == nb
    assert d0 == d1 == 13                # slowest profile anywhere: 3 + 10
    assert t0 == t1 == [0, nb, 0, 0]
    assert m0 == m1 == 13.0
    assert dl0 == dl1 == 16              # checks after 14 iterations find nothing active; read after the next two
    assert tl0 == tl1 == [0, nb, 0, 0]
    assert s0 == s1 == [float(nb), 512.0 * nb, 3.0, 0.5, 1.0]


def test_bench_starts_its_own_ranks():
    """`bench.py --gpus 2` without a launcher starts two rank processes itself; on a machine without GPUs both stop
    with the product path's "needs a GPU" message and the parent reports the failure instead of printing a line."""
    import subprocess
    if torch.cuda.is_available():
        pytest.skip('a GPU is present: covered by the GPU tests')
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1'],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=280)
    assert r.returncode != 0
    assert not any(line.startswith('{') for line in r.stdout.splitlines())       # no result line
    assert (r.stderr + r.stdout).count('needs a GPU') == 2 and 'rank(s) failed' in r.stderr
