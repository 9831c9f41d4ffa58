#!/usr/bin/env python3
"""Benchmark of the fused Picard iteration on the synthetic Shestakov ensemble.

    python bench.py --gpus N --steps K --warmup W

One "step" = one Picard iteration (EWMA -> flux -> split -> EWMA -> assembly -> residual -> tridiagonal
solve -> validity check) of every profile of the ensemble, all profiles forced active.  --iters-per-launch k
(default 64) runs k consecutive steps in one kernel launch with every profile held in registers in between: the
arithmetic, the per-iteration convergence test and the residual history are those of k single launches, only
the HBM round trip of the state between iterations is gone (k = 1 gives the one-launch-per-iteration form).  Each GPU (one
process per GPU; for N > 1 launch through torch.distributed.run) holds 65 536 profiles x 512 grid points,
fp64, resident in HBM before the timed region; the batch shards with no data exchange, the only collective is
the all-reduce of the convergence counters (RCCL), issued every --check-every steps inside the timed region.

Rank 0 prints ONE JSON line.  `value` is in units of "Picard iterations per second of a 65 536-profile
ensemble": (profiles on all GPUs / 65 536) x steps / time, so that at N = 1 it is literally the iteration rate
of the 64K-profile ensemble the metric names and at N GPUs the aggregate of N such shards (weak scaling).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E nominal peak
BYTES_PER_POINT = 72           # 5 arrays read + 4 written per point per iteration, fp64
PROFILES_PER_GPU = 65536
N_GRID = 512


def cpu_baseline(n, seconds=12.0):
    """The CPU restatement of the same iteration (oracle/tango_oracle.c, OpenMP over profiles) on a 4096-profile
    subsample of the same ensemble, timed on this host.  Checker code used as a reported baseline only."""
    from oracle import c_oracle
    from tango_amd.ensemble import shestakov_ensemble
    sub = 4096
    ens = shestakov_ensemble(PROFILES_PER_GPU, n, lo=0, hi=sub)
    threads = min(16, os.cpu_count() or 1)
    E = c_oracle.Ensemble1F(ens, fixed_count=True, max_iter=1)
    used = E.step(threads)                       # warm-up, also reports the thread count
    t0 = time.perf_counter()
    steps = 0
    while True:
        E.step(threads)
        steps += 1
        el = time.perf_counter() - t0
        if el >= seconds:
            break
    rate = steps * sub / PROFILES_PER_GPU / el
    faithful = cpu_faithful(n)
    return {'value': rate, 'unit': 'iters/s (65536-profile ensemble equivalents)', 'cores': int(used), 'kind': 'port',
            'sample': f'{steps} iterations of the first {sub} of the 65536 profiles, N={n}, {el:.1f} s wall, '
                      f'C restatement of the reference path (oracle/tango_oracle.c), OpenMP {used} threads',
            'profile_iterations_per_s': steps * sub / el, 'faithful_numpy': faithful}


def cpu_faithful(n, seconds=3.0):
    """The reference's own shape of the computation: one profile at a time on one core, NumPy expressions and
    SciPy's banded solver (oracle/tango_oracle.py), for a few seconds on member 0 of the ensemble."""
    from oracle import tango_oracle as orc
    from tango_amd.ensemble import shestakov_ensemble
    ens = shestakov_ensemble(PROFILES_PER_GPU, n, lo=0, hi=1)
    st = orc.State1F(ens['u0'][0])
    H1 = np.ones(n)
    H7 = ens['S0'][0] * (ens['x'] < ens['delta'][0])
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < seconds:
        orc.picard_step_1f(st, 1e4, ens['dx'], ens['nL'][0], H1, H7, 0.3, 1.0)
        steps += 1
    el = time.perf_counter() - t0
    return {'profile_iterations_per_s': steps / el, 'cores': 1,
            'sample': f'{steps} iterations of member 0, N={n}, {el:.1f} s, NumPy + scipy.linalg.solve_banded one profile at a time'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=192)
    ap.add_argument('--warmup', type=int, default=64)
    ap.add_argument('--iters-per-launch', type=int, default=64,
                    help='Picard iterations per kernel launch (profiles stay in registers in between); 1 = one launch per step')
    ap.add_argument('--check-every', type=int, default=64, help='steps between convergence-counter all-reduces')
    ap.add_argument('--profiles-per-gpu', type=int, default=PROFILES_PER_GPU)
    ap.add_argument('--n', type=int, default=N_GRID)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import tango_amd
    from tango_amd import _hip as H
    from tango_amd import distributed as D
    from tango_amd.ensemble import OneFieldEnsemble, shard_range, shestakov_ensemble

    rank, world, local = D.init_from_env()
    if world != args.gpus and rank == 0:
        print(f'warning: --gpus {args.gpus} but WORLD_SIZE={world}', file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the product path has no CPU fallback')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    stream = torch.cuda.current_stream()
    H.set_stream(stream.cuda_stream)

    nb_total = args.profiles_per_gpu * world
    lo, hi = shard_range(nb_total, rank, world)
    ens = shestakov_ensemble(nb_total, args.n, lo=lo, hi=hi)
    alloc = lambda nbytes: torch.empty(nbytes, dtype=torch.uint8, device=dev)     # torch owns the state pool
    E = OneFieldEnsemble(ens, a_turb=0.3, a_prof=1.0, tol=1e-11, fixed_count=True, keep_err=False, alloc=alloc)
    counts = torch.zeros(4, dtype=torch.int64, device=dev)      # all-reduced over ranks by RCCL

    kper = max(1, args.iters_per_launch)
    check = max(kper, args.check_every - args.check_every % kper)

    def run(nsteps):
        """nsteps Picard iterations: launches of kper iterations (a shorter one at the end if needed), the
        convergence counters reduced on the device and all-reduced over the ranks every `check` steps."""
        done = 0
        while done < nsteps:
            k = min(kper, nsteps - done)
            E.step(stream.cuda_stream, iters=k)
            done += k
            if done % check == 0:
                E.reduce_status(stream.cuda_stream, counts.data_ptr())
                D.allreduce_counts(counts)

    def fence():
        if world > 1:
            dist.barrier(device_ids=[local])
        torch.cuda.synchronize()

    run(args.warmup)
    E.reduce_status(stream.cuda_stream, counts.data_ptr())      # whatever the warm-up length: the collective has run once
    D.allreduce_counts(counts)
    fence()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    run(args.steps)
    e1.record(stream)
    fence()
    wall = time.perf_counter() - t0
    ev_ms = e0.elapsed_time(e1)

    # the same step with one iteration per launch (every iteration crosses HBM), reported beside the headline
    k1_steps = max(8, args.steps // 4)
    for _ in range(3):
        E.step(stream.cuda_stream, iters=1)
    fence()
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s0.record(stream)
    for _ in range(k1_steps):
        E.step(stream.cuda_stream, iters=1)
    s1.record(stream)
    fence()
    k1_ms = s0.elapsed_time(s1) / k1_steps

    t = torch.tensor([wall, ev_ms, k1_ms], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall, ev_ms, k1_ms = float(t[0]), float(t[1]), float(t[2])
    bad = int(np.count_nonzero(E.status.to_host() & 6))

    if rank == 0:
        ms_per_step = wall * 1e3 / args.steps
        value = (nb_total / PROFILES_PER_GPU) * args.steps / wall
        launches = (args.steps + kper - 1) // kper
        bytes_per_launch = (hi - lo) * args.n * BYTES_PER_POINT * args.steps / launches
        kernel_ms = ev_ms / launches            # launches are back to back on one stream: events bracket them
        achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
        traffic = traffic1 = None
        tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tpath):        # HBM bytes per launch from the committed rocprofv3 counter passes
            try:
                tj = json.load(open(tpath))
                traffic = tj.get(f'picard_1f_{args.profiles_per_gpu}x{args.n}_k{kper}')
                traffic1 = tj.get(f'picard_1f_{args.profiles_per_gpu}x{args.n}_k1')
            except Exception:
                traffic = traffic1 = None
        k1_bytes = (hi - lo) * args.n * BYTES_PER_POINT
        k1_achieved = k1_bytes / (k1_ms * 1e-3) / 1e9
        out = {
            'metric': 'Picard iters/sec, 64K-profile Shestakov ensemble N=512 fp64; % of HBM peak',
            'value': value,
            'unit': 'iters/s (65536-profile ensemble equivalents)',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': f'one-field Shestakov ensemble, {args.profiles_per_gpu} profiles x N={args.n} per GPU, '
                                   f'EWMA 0.3/1.0, dt=1e4, fused assemble+solve, all profiles active, seed 12345',
                       'profiles_total': nb_total, 'grid_points': args.n, 'check_every': check,
                       'iters_per_launch': kper,
                       'parallelism': f'batch sharded over {world} GPU(s), RCCL all-reduce of 4 int64 counters'},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic,
                         'kernel': 'picard_1f_kernel', 'kernel_ms': kernel_ms, 'iterations_per_launch': kper,
                         'algorithmic_bytes_per_launch': bytes_per_launch,
                         'note': 'algorithmic bytes = 72 B/point/iteration x iterations per launch; with more than one '
                                 'iteration per launch the profiles stay in registers between iterations, so HBM traffic '
                                 '(see traffic) is below the algorithmic bytes and frac can exceed what HBM alone allows'},
            'single_iteration_launch': {'kernel_ms': k1_ms, 'iters_per_s': (nb_total / PROFILES_PER_GPU) * 1e3 / k1_ms,
                                        'achieved': k1_achieved, 'frac': k1_achieved / HBM_PEAK_GBS, 'unit': 'GB/s',
                                        'traffic': traffic1, 'steps': k1_steps,
                                        'note': 'one Picard iteration per launch: the state crosses HBM every iteration'},
            'invalid_profiles': bad,
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.n, args.cpu_seconds)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier(device_ids=[local])
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
