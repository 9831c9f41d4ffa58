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


def test_residual_summary_matches_numpy():
    """tg_residual_reduce against the same sums formed on the host from the residual history."""
    from tango_amd import _hip as H
    from tango_amd.ensemble import OneFieldEnsemble, shestakov_ensemble
    nb, n = 301, 256
    E = OneFieldEnsemble(shestakov_ensemble(nb, n), tol=3e-6, max_iter=40)
    for _ in range(5):
        E.step(iters=4)           # 20 iterations: some profiles have converged, some have not
    out = H.DeviceArray((5,))
    E.reduce_residual(out.ptr)
    got = out.to_host()
    status, iters, err = E.status.to_host(), E.iters.to_host(), E.err.to_host()
    last = err[np.arange(nb), iters - 1]
    active = (status & 3) == 0
    assert 0 < active.sum() < nb
    assert got[2] == active.sum() and got[1] == n * active.sum()
    assert abs(got[0] - np.sum(last[active] ** 2 * n)) <= 1e-12 * got[0]
    assert got[3] == last.max() and got[4] == float(np.any(status & 2))


def test_lagged_convergence_checks_change_nothing():
    """ConvergenceMonitor (verdict of a check read one check later, no host wait) ends one interval later than a loop
    that reads every check at once, with the same final state bit for bit."""
    from tango_amd import distributed as D
    from tango_amd import _hip as H
    from tango_amd.ensemble import OneFieldEnsemble, shestakov_ensemble
    dev = torch.device('cuda', 0)
    stream = torch.cuda.current_stream()
    previous = H.stream()
    H.set_stream(stream.cuda_stream)
    try:
        res = []
        for lagged in (False, True):
            E = OneFieldEnsemble(shestakov_ensemble(NB, N), tol=TOL, max_iter=MAXIT)
            mon = D.ConvergenceMonitor(E, dev, stream)
            done = 0
            while done < MAXIT:
                E.step(stream.cuda_stream, iters=CHECK)
                done += CHECK
                if lagged:
                    r = mon.previous()
                    if r is not None and r['active'] == 0:
                        break
                    mon.issue()
                else:
                    mon.issue()
                    if mon.previous()['active'] == 0:
                        break
            res.append((done, E.u.to_host(), E.iters.to_host(), E.status.to_host()))
        assert res[1][0] == res[0][0] + CHECK
        for a, b in zip(res[0][1:], res[1][1:]):
            assert np.array_equal(a, b)
    finally:
        H.set_stream(previous)


def _worker_rccl(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                      LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY='0')
    import torch as t
    from tango_amd import _hip as H
    from tango_amd import distributed as D
    from tango_amd.ensemble import OneFieldEnsemble, shard_range, shestakov_ensemble
    r, w, local = D.init_from_env(backend='nccl')
    t.cuda.set_device(local)
    dev = t.device('cuda', local)
    stream = t.cuda.current_stream()
    H.set_stream(stream.cuda_stream)
    lo, hi = shard_range(NB, r, w)
    E = OneFieldEnsemble(shestakov_ensemble(NB, N, lo=lo, hi=hi), tol=TOL, max_iter=MAXIT)
    mon = D.ConvergenceMonitor(E, dev, stream)
    done = 0
    while done < MAXIT:
        E.step(stream.cuda_stream, iters=CHECK)
        done += CHECK
        mon.issue()
        res = mon.previous()
        if res['active'] == 0:
            break
    np.savez(os.path.join(outdir, f'rccl{r}.npz'), lo=lo, hi=hi, done=done, u=E.u.to_host(), iters=E.iters.to_host(),
             converged=res['converged'], max_rms=res['max_rms'])
    import torch.distributed as dist
    dist.barrier(device_ids=[local])
    dist.destroy_process_group()


def test_two_ranks_on_two_gpus_over_rccl(tmp_path):
    """The RCCL leg: two ranks, one GPU each, counters and residual summary all-reduced with backend nccl."""
    if torch.cuda.device_count() < 2:
        pytest.skip('needs two GPUs')
    ctx = mp.get_context('spawn')
    port = _free_port()
    procs = [ctx.Process(target=_worker_rccl, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=170)
        assert p.exitcode == 0
    done1, u1, it1, st1 = _run_shard(0, NB, lambda c: c)
    parts = [np.load(tmp_path / f'rccl{r}.npz') for r in range(2)]
    assert int(parts[0]['done']) == int(parts[1]['done']) == done1
    assert float(parts[0]['converged']) == float(parts[1]['converged']) == NB
    assert float(parts[0]['max_rms']) == float(parts[1]['max_rms'])
    assert np.array_equal(np.concatenate([p['u'] for p in parts]), u1)
    assert np.array_equal(np.concatenate([p['iters'] for p in parts]), it1)


@pytest.mark.parametrize('extra', [['--iters-per-launch', '8'], ['--iters-per-launch', '1'], ['--config', 'c4', '--profiles-per-gpu', '64'],
                                   ['--config', 'c3', '--profiles-per-gpu', '512'], ['--iters-per-launch', '64', '--check-every', '64'],
                                   ['--config', 'c5', '--iters-per-launch', '8', '--theta', 'convective']])
def test_bench_line(extra):
    """bench.py end to end on one GPU at a reduced ensemble size: one JSON line whose launch shape is the one that ran
    and whose roofline fractions are fractions."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '24', '--warmup', '8', '--check-every', '8',
           '--no-cpu-baseline'] + (extra if '--profiles-per-gpu' in extra else extra + ['--profiles-per-gpu', '2048'])
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out['n_gpus'] == 1 and out['steps'] == 24 and out['value'] > 0 and out['invalid_profiles'] == 0
    shape = out['config']['launch_shape']
    k = shape['iterations_per_launch']
    assert shape['launches_in_timed_region'] == -(-24 // k) and shape['all_reduce_rounds_in_timed_region'] == 24 // shape['check_every']
    roof = out['roofline']
    # one iteration per launch crosses HBM every iteration, and so does the segmented form's chain of kernels (c4); only the
    # register-resident multi-iteration kernel is bound by fp64 issue
    assert roof['bound'] == ('hbm' if (k == 1 or 'c4' in extra) else 'valu_fp64')
    if 'c4' in extra:
        assert k == 16 and roof['iterations_per_launch'] == 12 and 'hbm' in roof
    assert roof['frac'] is None or 0 < roof['frac'] <= 1
    assert 0 < out['hbm_frac_headline'] <= 1
    if k > 1:
        assert roof['hbm']['bound'] == 'hbm' and roof['hbm']['steps'] >= 100 
        assert out['hbm_frac_headline'] == (roof['frac'] if roof['bound'] == 'hbm' else roof['hbm']['frac'])
    assert out['config']['world_size'] == 1 and out['config']['theta_params']['name'] == ('convective' if '--theta' in extra else 'example')
    assert out['convergence_summary_last_check']['active'] == out['config']['profiles_total']


def test_bench_two_ranks_share_the_gpu():
    """bench.py --gpus 2 --backend gloo: the multi-rank path of the benchmark (rank processes started by bench.py itself,
    contiguous shards of the two-shard ensemble, barriers, all-reduces inside the timed region, maximum over the ranks)
    rehearsed with both ranks on this box's one GPU.  With the driver's own arguments (20 steps, 64 iterations per
    launch) the timed region still holds one all-reduce round."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '20', '--warmup', '5', '--backend', 'gloo',
           '--config', 'c5', '--profiles-per-gpu', '4096', '--clock-warmup-ms', '5']
    env['TANGO_COLLECTIVE_TIMEOUT'] = '60'
    env['TANGO_BENCH_WATCHDOG'] = '100'       # a stuck rank reports where it stands and ends; bench.py then ends the other
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=150)
    except subprocess.TimeoutExpired as e:
        err = e.stderr.decode(errors='replace') if isinstance(e.stderr, bytes) else (e.stderr or '')
        raise AssertionError('bench.py --gpus 2 did not finish: ' + err[-3000:])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    out = json.loads(lines[0])
    cfg = out['config']
    assert out['n_gpus'] == 2 and cfg['world_size'] == 2 and cfg['backend'] == 'gloo' and cfg['profiles_total'] == 8192
    assert cfg['launch_shape']['all_reduce_rounds_in_timed_region'] == 1
    assert out['convergence_summary_last_check']['active'] == 8192 and out['invalid_profiles'] == 0
    assert 'cpu_baseline' not in out and out['value'] > 0
