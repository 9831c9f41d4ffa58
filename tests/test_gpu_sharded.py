"""Two processes share the GPU, each stepping its contiguous shard of an ensemble (tango_amd.ensemble.shard_range);
the convergence counters are all-reduced (gloo here, RCCL in bench.py).  The union of the shards must be the
single-process result, and both ranks must leave the loop on the same iteration."""
import os
import socket
import sys

import numpy as np
import pytest

torch = pytest.importorskip('torch')
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NB, N, TOL, MAXIT, CHECK = 37, 256, 1e-9, 200, 4


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _run_shard(lo, hi, reduce_counts):
    from tango_amd.ensemble import OneFieldEnsemble, shestakov_ensemble
    E = OneFieldEnsemble(shestakov_ensemble(NB, N, lo=lo, hi=hi), tol=TOL, max_iter=MAXIT)

    def counts():
        E.reduce_status()
        return torch.from_numpy(E.counts.to_host().astype(np.int64))
    done = 0
    while done < MAXIT:
        E.step(iters=CHECK)
        done += CHECK
        if int(reduce_counts(counts())[0]) == 0:
            break
    return done, E.u.to_host(), E.iters.to_host(), E.status.to_host()


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), LOCAL_RANK='0')
    from tango_amd import distributed as D
    from tango_amd.ensemble import shard_range
    r, w, _ = D.init_from_env(backend='gloo')
    lo, hi = shard_range(NB, r, w)
    done, u, iters, status = _run_shard(lo, hi, D.allreduce_counts)
    np.savez(os.path.join(outdir, f'rank{r}.npz'), lo=lo, hi=hi, done=done, u=u, iters=iters, status=status)
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_one_gpu_equal_one_process(tmp_path):
    import tango_amd
    assert tango_amd.device_count() > 0
    ctx = mp.get_context('spawn')
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=150)
        assert p.exitcode == 0
    done1, u1, it1, st1 = _run_shard(0, NB, lambda c: c)
    parts = [np.load(tmp_path / f'rank{r}.npz') for r in range(2)]
    assert int(parts[0]['lo']) == 0 and int(parts[0]['hi']) == int(parts[1]['lo']) and int(parts[1]['hi']) == NB
    assert int(parts[0]['done']) == int(parts[1]['done']) == done1
    assert np.array_equal(np.concatenate([p['iters'] for p in parts]), it1)
    assert np.array_equal(np.concatenate([p['status'] for p in parts]), st1)
    assert np.array_equal(np.concatenate([p['u'] for p in parts]), u1)        # same kernels on the same rows: bit for bit
    assert np.all(st1 & 1) and len(set(it1.tolist())) > 2
