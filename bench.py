#!/usr/bin/env python3
"""Benchmark of the fused Picard iteration on the synthetic ensembles of the benchmark set.

    python bench.py --gpus N --steps K --warmup W [--config metric|c2|c3|c4|c5] [--iters-per-launch k] [--theta convective]

One "step" = one Picard iteration (EWMA -> flux -> split -> EWMA -> assembly -> residual -> solve -> validity
check) of every profile of the ensemble, all profiles forced active.  Configurations (per GPU):
  metric  one field, 65 536 profiles x 512 points, EWMA 0.3/1.0, dt = 1e4       (the headline workload)
  c2      one field, 65 536 x 256
  c3      three fields n, pi<->pe with nonlinear nu, 16 384 x 512, EWMA 0.3/0.3, dt = 1e6
  c4      one field, 4 096 x 8192, EWMA 0.3/0.3 (segmented form of the step)
  c5      one field, 131 072 x 512 per GPU: with --gpus 8 the 1 048 576-profile ensemble (rank r holds the r-th eighth)
--theta convective replaces the example's thetaParams (Dmax = 1e13: the whole flux is diffusive, c = 0) by Dmin = 0.1,
Dmax = 10, dpdxThreshold = 1, with which part of the flux of most profiles is convective (c != 0): the upwinding
weights and the EWMA of c then work on numbers, not on zeros.
--iters-per-launch k (one-field configurations up to 2048 points; default 64 for metric and c2) runs k consecutive
steps in one kernel launch with every profile held in registers in between: the arithmetic, the per-iteration
convergence test and the residual history are those of k single launches, only the HBM round trip of the state
between iterations is gone.  For c4 (default 16) the k iterations of a call are a chain of kernels in which the second
half of an iteration and the first half of the next share a kernel: every iteration still crosses HBM, with 78 instead
of 109 bytes per point.

N > 1: one process per GPU.  Started by a launcher (RANK / WORLD_SIZE in the environment, e.g.
torch.distributed.run) every process is one rank; started plainly with --gpus N this process starts the N ranks
itself, before it touches any GPU, and relays rank 0's line.  The ensemble shards with no data exchange
(weak scaling: every GPU holds a full-size shard); the only collectives are the all-reduces of the convergence
counters and of the residual summary (RCCL), issued every --check-every steps inside the timed region (and once at
its end when it is shorter than that: a timed region never goes without a check); their result, read one check later so
that the host never waits for the GPU, decides whether the loop goes on.

Rank 0 prints ONE JSON line.  `value` is in "Picard iterations per second of a full-size ensemble": (profiles on
all GPUs / profiles per GPU of the configuration) x steps / time.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E nominal peak
FP64_VECTOR_PEAK_TFLOPS = 78.6   # 256 CUs x 4 SIMDs x 16 lanes x 2 flop (fma) x 2.4 GHz
CLOCK_HZ = 2.4e9
SIMDS = 1024
CONFIGS = {
    #          fields  profiles  n     a_prof  dt    bytes/point  default k
    'metric': (1,      65536,    512,  1.0,    1e4,  72,          64),
    'c2':     (1,      65536,    256,  1.0,    1e4,  72,          64),
    'c3':     (3,      16384,    512,  0.3,    1e6,  216,         1),
    'c4':     (1,      4096,     8192, 0.3,    1e4,  72,          16),
    'c5':     (1,      131072,   512,  1.0,    1e4,  72,          64),
}
THETAS = {'example': dict(Dmin=1e-5, Dmax=1e13, dpdxThreshold=10.0), 'convective': dict(Dmin=0.1, Dmax=10.0, dpdxThreshold=1.0)}
K1_MIN_LAUNCHES = 100            # launches of the single-iteration leg, whatever --steps
METRIC = 'Picard iters/sec, 64K-profile Shestakov ensemble N=512 fp64; % of HBM peak'


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def spawn_ranks(n):
    """Start n fresh rank processes of this script (no GPU call has been made in this process) and relay rank 0's
    JSON line.  Returns the exit code for this process."""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    # A rank that dies leaves the others waiting in a collective for as long as the backend's timeout lasts: watch them
    # all, and when one has failed end the others (these exact processes) instead of waiting with them.
    import threading
    out = []
    reader = threading.Thread(target=lambda: out.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    codes = [None] * n
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
        if any(c not in (None, 0) for c in codes):
            time.sleep(2.0)      # let the others fail by themselves if they are about to
            for r, p in enumerate(procs):
                if p.poll() is None:
                    p.kill()
                codes[r] = p.wait()
            break
        time.sleep(0.05)
    reader.join(timeout=5)
    for line in ''.join(x or '' for x in out).splitlines():
        print(line, flush=True)
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        print(f'bench.py: rank(s) failed: {bad}', file=sys.stderr)
        return 1
    return 0


def moved_into(struct, lo, nbytes, delta):
    """A copy of a ctypes parameter block in which every device pointer into [lo, lo + nbytes) is moved by delta: the
    same block for a byte-for-byte copy of that memory range."""
    import ctypes as C
    new = type(struct).from_buffer_copy(struct)
    for name, ctype in struct._fields_:
        if ctype is C.c_void_p:
            v = getattr(struct, name)
            if v and lo <= v < lo + nbytes:
                setattr(new, name, v + delta)
        elif isinstance(ctype, type) and issubclass(ctype, C.Array) and ctype._type_ is C.c_void_p:
            arr = getattr(new, name)
            for k, v in enumerate(getattr(struct, name)):
                if v and lo <= v < lo + nbytes:
                    arr[k] = v + delta
    return new


def cpu_model():
    try:
        for line in open('/proc/cpuinfo'):
            if line.startswith('model name'):
                return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def cpu_threads():
    """The cores this process may use: the share the scheduler of the machine gave it (OMP_NUM_THREADS when the
    environment sets it, as a box that hands out a slice of a larger host does; else the affinity mask).  os.cpu_count()
    is the whole host, which on a shared machine is not ours to fill."""
    env = os.environ.get('OMP_NUM_THREADS', '')
    if env.isdigit() and int(env) > 0:
        return int(env), 'OMP_NUM_THREADS'
    try:
        return len(os.sched_getaffinity(0)), 'affinity mask'
    except AttributeError:
        return os.cpu_count() or 1, 'os.cpu_count()'


def cpu_baseline_1f(nprof, n, a_prof, seconds, theta):
    """The CPU restatement of the same iteration (oracle/tango_oracle.c, OpenMP over profiles) on a subsample of the
    same ensemble, timed on this host.  Checker code used as a reported baseline only."""
    from oracle import c_oracle
    from tango_amd.ensemble import shestakov_ensemble
    sub = max(16, min(nprof, (4096 * 512) // n))
    ens = shestakov_ensemble(nprof, n, lo=0, hi=sub)
    threads, threads_from = cpu_threads()
    E = c_oracle.Ensemble1F(ens, a_prof=a_prof, fixed_count=True, max_iter=1, Dmin=theta['Dmin'], Dmax=theta['Dmax'], thr=theta['dpdxThreshold'])
    used = E.step(threads)                       # warm-up, also reports the thread count
    t0 = time.perf_counter()
    steps = 0
    while True:
        E.step(threads)
        steps += 1
        el = time.perf_counter() - t0
        if el >= seconds:
            break
    rate = steps * sub / nprof / el
    return {'value': rate, 'unit': f'iters/s ({nprof}-profile ensemble equivalents)', 'cores': int(used), 'kind': 'port',
            'sample': f'{steps} iterations of the first {sub} of the {nprof} profiles, N={n}, {el:.1f} s wall, '
                      f'C restatement of the reference path (oracle/tango_oracle.c), OpenMP {used} threads',
            'cpu_model': cpu_model(), 'host_cpus_visible': os.cpu_count(), 'threads_from': threads_from,
            'profile_iterations_per_s': steps * sub / el, 'faithful_numpy': cpu_faithful_1f(nprof, n, a_prof, theta)}


def cpu_faithful_1f(nprof, n, a_prof, theta, seconds=3.0):
    """The reference's own shape of the computation: one profile at a time on one core, NumPy expressions and
    SciPy's banded solver (oracle/tango_oracle.py), for a few seconds on member 0 of the ensemble."""
    from oracle import tango_oracle as orc
    from tango_amd.ensemble import shestakov_ensemble
    ens = shestakov_ensemble(nprof, n, lo=0, hi=1)
    st = orc.State1F(ens['u0'][0])
    H1 = np.ones(n)
    H7 = ens['S0'][0] * (ens['x'] < ens['delta'][0])
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < seconds:
        orc.picard_step_1f(st, 1e4, ens['dx'], ens['nL'][0], H1, H7, 0.3, a_prof, theta)
        steps += 1
    el = time.perf_counter() - t0
    return {'profile_iterations_per_s': steps / el, 'cores': 1,
            'sample': f'{steps} iterations of member 0, N={n}, {el:.1f} s, NumPy + scipy.linalg.solve_banded one profile at a time'}


def cpu_baseline_3f(nprof, n, seconds):
    """Three fields: the NumPy restatement (oracle/tango_oracle.py: two LAPACK band solves per iteration, as the
    reference) on member 0, one core."""
    from oracle import tango_oracle as orc
    from tango_amd.ensemble import threefield_ensemble
    ens = threefield_ensemble(nprof, n, lo=0, hi=1)
    st = orc.State3F(ens['u0'][0], ens['u0'][0], ens['u0'][0])
    t0 = time.perf_counter()
    steps = 0
    while time.perf_counter() - t0 < seconds:
        orc.picard_step_3f(st, 1e6, ens['dx'], ens['bc'][0], ens['nu0'][0], ens['x'], 0.3, 0.3)
        steps += 1
    el = time.perf_counter() - t0
    return {'value': steps / el / nprof, 'unit': f'iters/s ({nprof}-profile ensemble equivalents)', 'cores': 1, 'kind': 'port',
            'sample': f'{steps} iterations of member 0 of the three-field ensemble, N={n}, {el:.1f} s wall, NumPy restatement '
                      f'(oracle/tango_oracle.py) with scipy.linalg.solve_banded, one profile at a time on one core',
            'profile_iterations_per_s': steps / el}


def committed(key):
    """A number from the committed rocprofv3 counter passes (profiles/traffic.json), or None."""
    try:
        return json.load(open(os.path.join(ROOT, 'profiles', 'traffic.json'))).get(key)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=192)
    ap.add_argument('--warmup', type=int, default=64)
    ap.add_argument('--config', choices=sorted(CONFIGS), default='metric')
    ap.add_argument('--iters-per-launch', type=int, default=0,
                    help='Picard iterations per kernel launch (profiles stay in registers in between); 0 = the configuration\'s default')
    ap.add_argument('--check-every', type=int, default=64, help='steps between convergence all-reduces')
    ap.add_argument('--theta', choices=sorted(THETAS), default='example',
                    help='thetaParams of the flux split: the example\'s (c = 0 everywhere) or a set that makes part of the flux convective')
    ap.add_argument('--clock-warmup-ms', type=float, default=50.0,
                    help='untimed launches before the warm-up steps, until the GPU has been busy this long (0: none)')
    ap.add_argument('--backend', choices=['nccl', 'gloo'], default='nccl',
                    help='torch.distributed backend of the all-reduces: nccl (= RCCL, one GPU per rank) or gloo (a rehearsal of the '
                         'multi-rank path on fewer GPUs than ranks: rank r uses GPU r modulo the visible ones)')
    ap.add_argument('--profiles-per-gpu', type=int, default=0)
    ap.add_argument('--n', type=int, default=0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=12.0)
    args = ap.parse_args()
    if args.gpus < 1 or args.steps < 1 or args.warmup < 0:
        raise SystemExit('bench.py: --gpus and --steps must be positive, --warmup non-negative')

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ and 'RANK' not in os.environ:
        sys.exit(spawn_ranks(args.gpus))         # nothing in this process has touched a GPU
    if os.environ.get('TANGO_BENCH_WATCHDOG', '').isdigit():
        # a run that has not finished after this many seconds writes where every thread stands to stderr and ends
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['TANGO_BENCH_WATCHDOG']), exit=True)

    import torch
    import torch.distributed as dist
    import tango_amd
    from tango_amd import _hip as H
    from tango_amd import distributed as D
    from tango_amd.ensemble import OneFieldEnsemble, ThreeFieldEnsemble, shard_range, shestakov_ensemble, threefield_ensemble

    rank, world, local = D.init_from_env(backend=args.backend if torch.cuda.is_available() else 'gloo')
    if world != args.gpus:
        if rank == 0:
            print(f'bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks', file=sys.stderr)
        sys.exit(2)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU: the product path has no CPU fallback')
    if args.backend == 'gloo' and world > 1:
        local %= torch.cuda.device_count()
    if torch.cuda.device_count() <= local:
        raise SystemExit(f'bench.py: rank {rank} needs GPU {local} but {torch.cuda.device_count()} are visible')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    stream = torch.cuda.current_stream()
    H.set_stream(stream.cuda_stream)

    fields, per_gpu, n, a_prof, dt, bpp, kdef = CONFIGS[args.config]
    per_gpu = args.profiles_per_gpu or per_gpu
    n = args.n or n
    kper = max(1, args.iters_per_launch or kdef)
    if fields == 3:
        kper = 1                                   # the three-field kernel runs one iteration per launch
    check = max(kper, args.check_every - args.check_every % kper)
    k1_steps = max(K1_MIN_LAUNCHES, args.steps // 4) if kper > 1 else 0
    theta = THETAS[args.theta]
    if fields == 3 and args.theta != 'example':
        raise SystemExit('bench.py: --theta applies to the one-field configurations')
    rows = args.warmup + args.steps + k1_steps + 8  # residual history columns: every iteration of this run
    nb_total = per_gpu * world
    lo, hi = shard_range(nb_total, rank, world)
    alloc = lambda nbytes: torch.empty(nbytes, dtype=torch.uint8, device=dev)     # torch owns the state pool
    if fields == 1:
        ens = shestakov_ensemble(nb_total, n, lo=lo, hi=hi)
        E = OneFieldEnsemble(ens, dt=dt, a_turb=0.3, a_prof=a_prof, tol=1e-11, theta=theta, fixed_count=True, keep_err=True, max_iter=rows, alloc=alloc)
    else:
        ens = threefield_ensemble(nb_total, n, lo=lo, hi=hi)
        E = ThreeFieldEnsemble(ens, dt=dt, a_turb=0.3, a_prof=a_prof, tol=1e-9, fixed_count=True, keep_err=True, max_iter=rows, alloc=alloc)
    mon = D.ConvergenceMonitor(E, dev, stream)     # device reductions + RCCL all-reduces + asynchronous copy to the host

    def nothing_left():
        """The previous check's verdict (all ranks see the same reduced numbers): True when no profile anywhere
        still iterates.  Its copy landed long ago: a whole check interval of launches has been queued behind it."""
        res = mon.previous()
        return res is not None and res['active'] == 0.0

    def run(nsteps):
        """Up to nsteps Picard iterations: launches of kper iterations (a shorter one at the end if needed), a
        convergence check every `check` steps; the loop ends early when a check found nothing left to iterate.  With
        several ranks a run shorter than the check interval still ends with one check (device reductions + all-reduces
        + copy), so that no timed region, however few its steps, goes without the collective."""
        done = 0
        issued = mon.issued
        while done < nsteps:
            k = min(kper, nsteps - done)
            E.step(stream.cuda_stream, iters=k)
            done += k
            if done % check == 0:
                if nothing_left():
                    break
                mon.issue()
        if world > 1 and mon.issued == issued and nsteps > 0:
            nothing_left()
            mon.issue()
        return done

    def barrier():
        if world > 1:
            if args.backend == 'nccl':
                dist.barrier(device_ids=[local])
            else:
                torch.cuda.synchronize()
                dist.barrier()

    def fence():
        barrier()
        torch.cuda.synchronize()

    run(args.warmup)
    mon.issue()                  # whatever the warm-up length: both collectives have run once before the timed region
    mon.previous()
    # A GPU that has just been idle, or busy for a millisecond only, runs its next few milliseconds at a lower clock: the
    # same 20-iteration launch (22 403 vector instructions per wavefront either way, counted with rocprofv3) takes 4.0 ms
    # after 5 warm-up steps, 3.5 ms after 128 and 3.25 ms after 256 or more.  The requested warm-up can be that short
    # (5 steps = 1 ms), so the clock is brought up by further untimed launches right before the timed region; they are
    # neither warm-up steps nor timed steps, are reported as clock_warmup_iterations, and touch nothing of the ensemble:
    # they step a COPY of the whole state pool (profiles, EWMA memories, status, iteration counters, residual history, work
    # space; the same kernels on the same numbers), so the timed steps are iterations warmup .. warmup+steps-1 of the
    # ensemble's own trajectory.
    clock_warmup = 0
    if args.clock_warmup_ms > 0:
        barrier()                                # start together, so that the ranks also reach the timed region's barrier together
        twin = torch.empty(E.pool.nbytes, dtype=torch.uint8, device=dev)
        H.check(H.lib.tg_copy_d2d(twin.data_ptr(), E.pool.ptr, E.pool.nbytes, stream.cuda_stream))
        real = (E.state, E.params)
        E.state, E.params = (moved_into(v, E.pool.ptr, E.pool.nbytes, twin.data_ptr() - E.pool.ptr) for v in real)
        try:
            t_up = time.perf_counter()
            while (time.perf_counter() - t_up) * 1e3 < args.clock_warmup_ms:
                for _ in range(4):
                    E.step(stream.cuda_stream, iters=kper)
                    clock_warmup += kper
                torch.cuda.synchronize()
            for _ in range(2):
                E.step(stream.cuda_stream, iters=kper)
                clock_warmup += kper
        finally:
            E.state, E.params = real
    fence()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    issued_before = mon.issued
    t0 = time.perf_counter()
    e0.record(stream)
    done = run(args.steps)
    e1.record(stream)
    fence()
    wall = time.perf_counter() - t0
    ev_ms = e0.elapsed_time(e1)
    if done != args.steps:
        raise SystemExit(f'bench.py: the timed region ended after {done} of {args.steps} steps')
    timed_checks = mon.issued - issued_before

    # the same step with one iteration per launch (every iteration crosses HBM): the HBM roofline of the path
    k1_ms = None
    if kper > 1:
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

    t = torch.tensor([wall, ev_ms, k1_ms or 0.0], dtype=torch.float64, device=dev)
    if world > 1:
        D.allreduce_max(t)
    wall, ev_ms, k1_max = float(t[0]), float(t[1]), float(t[2])
    bad = int(np.count_nonzero(E.status.to_host() & 6))
    mon.issue()
    red = mon.previous()

    if rank == 0:
        shard = hi - lo
        ms_per_step = wall * 1e3 / args.steps
        value = (nb_total / per_gpu) * args.steps / wall
        launches = (args.steps + kper - 1) // kper
        kernel_ms = ev_ms / launches            # launches are back to back on one stream: events bracket them
        bytes_per_iter = shard * n * bpp
        tag = f'{args.config}_{per_gpu}x{n}'
        hbm_k1 = None
        segmented = fields == 1 and (n >= 8192 or (n == 4096 and kper == 1))
        if kper == 1 or segmented:
            # every iteration crosses HBM (the segmented form's k iterations per call are a chain of kernels)
            per_launch = args.steps / launches
            achieved = bytes_per_iter * per_launch / (kernel_ms * 1e-3) / 1e9
            if fields == 3:
                kernel = 'picard_3f_kernel'
            elif segmented and kper > 1:
                kernel = (f'one call = seg_halo_kernel, picard_seg_kernel<0>, then {kper - 1} x (picard_seg_reduce_kernel, '
                          f'picard_seg_kernel<2>), picard_seg_reduce_kernel, picard_seg_kernel<1>')
            elif segmented:
                kernel = 'one call = seg_halo_kernel, picard_seg_kernel<0>, picard_seg_reduce_kernel, picard_seg_kernel<1>'
            else:
                kernel = 'picard_1f_kernel'
            roofline = {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS,
                        'traffic': committed(f'{tag}_k{kper}_hbm_bytes_per_launch'), 'kernel': kernel,
                        'kernel_ms': kernel_ms, 'iterations_per_launch': per_launch, 'algorithmic_bytes_per_launch': bytes_per_iter * per_launch,
                        'note': f'algorithmic bytes = {bpp} B/point/iteration x {shard} profiles x {n} points x iterations per launch'}
            if kper > 1:
                k1_achieved = bytes_per_iter / (k1_max * 1e-3) / 1e9
                hbm_k1 = {'bound': 'hbm', 'achieved': k1_achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': k1_achieved / HBM_PEAK_GBS,
                          'traffic': committed(f'{tag}_k1_hbm_bytes_per_launch'), 'kernel_ms': k1_max, 'steps': k1_steps,
                          'algorithmic_bytes_per_launch': bytes_per_iter, 'note': 'the same step, one Picard iteration per call'}
        else:
            # k iterations per launch: the state stays in registers, HBM is idle most of the time and the kernel is bound by
            # fp64 vector issue.  Executed fp64 operations per wavefront-iteration come from the committed rocprofv3 pass
            # (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 over a launch of the same shape; an fma counts two).
            flops_wave_iter = committed(f'{tag}_fp64_flops_per_wave_iteration')
            valu_wave_iter = committed(f'{tag}_valu_insts_per_wave_iteration')
            waves = -(-shard * n // 512)              # one wavefront per 512 grid points
            roofline = {'bound': 'valu_fp64', 'achieved': None, 'peak': FP64_VECTOR_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': None,
                        'traffic': committed(f'{tag}_k{kper}_hbm_bytes_per_launch'), 'kernel': 'picard_1f_kernel (multi-iteration form)',
                        'kernel_ms': kernel_ms, 'iterations_per_launch': args.steps / launches}
            if flops_wave_iter:
                tf = flops_wave_iter * 64 * waves * (args.steps / launches) / (kernel_ms * 1e-3) / 1e12
                roofline.update(achieved=tf, frac=tf / FP64_VECTOR_PEAK_TFLOPS, fp64_flops_per_wave_iteration=flops_wave_iter)
            if valu_wave_iter:
                # every VALU instruction occupies its SIMD for at least 4 cycles (64 lanes over 16 ALUs)
                roofline['valu_issue_frac'] = valu_wave_iter * 4 * waves * (args.steps / launches) / (SIMDS * kernel_ms * 1e-3 * CLOCK_HZ)
            roofline['note'] = ('k iterations per launch keep the profiles in registers: HBM traffic per launch (traffic) is that of ONE '
                                'iteration, so the HBM roofline does not bound this kernel; achieved = executed fp64 operations / kernel time')
            roofline['counts_source'] = ('instruction counts per wavefront and iteration and HBM bytes per launch: rocprofv3 counter passes over '
                                         'this command, committed in profiles/traffic.json (not collected in this run); kernel_ms: HIP events, this run')
            k1_achieved = bytes_per_iter / (k1_max * 1e-3) / 1e9
            hbm_k1 = {'bound': 'hbm', 'achieved': k1_achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': k1_achieved / HBM_PEAK_GBS,
                      'traffic': committed(f'{tag}_k1_hbm_bytes_per_launch'), 'kernel': 'picard_1f_kernel (single-iteration form)',
                      'kernel_ms': k1_max, 'iters_per_s': (nb_total / per_gpu) * 1e3 / k1_max, 'steps': k1_steps,
                      'algorithmic_bytes_per_launch': bytes_per_iter,
                      'note': 'the same step, one Picard iteration per launch: the state crosses HBM every iteration (72 B/point)'}
        what = {1: f'one-field Shestakov ensemble, {per_gpu} profiles x N={n} per GPU'
                   + (f' ({nb_total} profiles in all: rank r holds the r-th contiguous range)' if world > 1 else '')
                   + f', EWMA 0.3/{a_prof}, dt={dt:g}, fused assemble+solve',
                3: f'three-field (n, pi<->pe, nonlinear nu) ensemble, {per_gpu} profiles x N={n} per GPU, EWMA 0.3/{a_prof}, dt={dt:g}, '
                   f'fused assemble + scalar and 2x2 block solves'}[fields]
        out = {
            'metric': METRIC if args.config == 'metric' else f'Picard iters/sec, configuration {args.config}; % of HBM peak',
            'value': value,
            'unit': f'iters/s ({per_gpu}-profile ensemble equivalents)',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': what + ', all profiles active, seeds as in tango_amd.ensemble',
                       'name': args.config, 'profiles_total': nb_total, 'grid_points': n, 'theta_params': dict(theta, name=args.theta),
                       'world_size': dist.get_world_size() if dist.is_initialized() else 1,
                       'backend': dist.get_backend() if dist.is_initialized() else None,
                       'launch_shape': {'iterations_per_launch': kper, 'launches_in_timed_region': launches,
                                        'iterations_in_last_launch': args.steps - (launches - 1) * kper,
                                        'clock_warmup_iterations': clock_warmup,
                                        'check_every': check, 'all_reduce_rounds_in_timed_region': timed_checks,
                                        'reduced_results_consumed': mon.consumed},
                       'parallelism': f'batch sharded over {world} rank(s), contiguous ranges, no data exchange; '
                                      f'{"RCCL" if args.backend == "nccl" else args.backend} all-reduce of 4 int64 counters + 5 fp64 residual '
                                      f'summary values per check' if world > 1 else 'one GPU (no collective traffic)'},
            'roofline': roofline,
            'convergence_summary_last_check': {k: (float(v) if v is not None else None) for k, v in red.items()},
            'invalid_profiles': bad,
        }
        if hbm_k1:
            # the metric's "% of HBM peak": the same step with one iteration per launch, timed in this run
            roofline['hbm'] = hbm_k1
        out['hbm_frac_headline'] = roofline['frac'] if roofline['bound'] == 'hbm' else hbm_k1['frac']
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline_3f(per_gpu, n, args.cpu_seconds) if fields == 3 else \
                cpu_baseline_1f(per_gpu, n, a_prof, args.cpu_seconds, theta)
        print(json.dumps(out), flush=True)
    if world > 1:
        barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
