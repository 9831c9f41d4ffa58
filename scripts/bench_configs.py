"""Throughput of the fused kernels on every single-GPU configuration of the benchmark set, plus time to
convergence with the per-profile masks on.  Prints one JSON line per measurement (HIP events on the launch
stream).  usage: python scripts/bench_configs.py [c1] [c2] [metric] [c3] [c4] [conv]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tango_amd as tango
from tango_amd import _hip as H
from tango_amd.ensemble import OneFieldEnsemble, shestakov_ensemble, threefield_ensemble

PEAK = 8000.0


def timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def report(name, nb, n, fields_bytes, ms_per_iter, **kw):
    gbs = nb * n * fields_bytes / (ms_per_iter * 1e-3) / 1e9
    out = {'config': name, 'profiles': nb, 'N': n, 'ms_per_iteration': round(ms_per_iter, 5), 'iters_per_s': round(1e3 / ms_per_iter, 1),
           'algorithmic_GBps': round(gbs, 1), 'frac_of_8TBps': round(gbs / PEAK, 4)}
    out.update(kw)
    print(json.dumps(out), flush=True)


def one_field(name, nb, n, a_prof, ks, reps=24):
    st = torch.cuda.current_stream().cuda_stream
    E = OneFieldEnsemble(shestakov_ensemble(nb, n), a_turb=0.3, a_prof=a_prof, fixed_count=True, keep_err=False)
    for k in ks:
        for _ in range(3):
            E.step(st, iters=k)
        ms = timed(lambda: E.step(st, iters=k), max(4, reps // k)) / k
        bad = int(np.count_nonzero(E.status.to_host() & 6))
        report(name, nb, n, 72, ms, iters_per_launch=k, invalid=bad)
    del E


def three_field(nb=16384, n=512, reps=24):
    ens = threefield_ensemble(nb, n)
    T = tango.extras.threefield
    dx, x = ens['dx'], ens['x']
    th = {'Dmin': 1e-5, 'Dmax': 1e13, 'dpdxThreshold': 10}
    nu0, bc = ens['nu0'], ens['bc']
    cbs = [T.ComputeAllH_n(), T.ComputeAllH_pi(nu0), T.ComputeAllH_pe(nu0)]
    lms = [tango.lodestro_method.lm(0.3, 0.3, th) for _ in range(3)]
    F = tango.multifield.Field
    ic = ens['u0']
    fields = [F(label='n', rightBC=bc[:, 0].copy(), profile_mminus1=1.0 * ic, compute_all_H=cbs[0], lodestroMethod=lms[0]),
              F(label='pi', rightBC=bc[:, 1].copy(), profile_mminus1=1.0 * ic, compute_all_H=cbs[1], lodestroMethod=lms[1], coupledTo='pe'),
              F(label='pe', rightBC=bc[:, 2].copy(), profile_mminus1=1.0 * ic, compute_all_H=cbs[2], lodestroMethod=lms[2], coupledTo='pi')]
    tango.multifield.check_fields_initialize(fields)
    call = tango.multifield.ComputeAllHAllFields(fields, tango.lodestro_method.TurbulenceHandler(dx, x, T.ShestakovThreeFieldFluxModel(dx)))
    s = tango.solver.Solver(1, x, np.array([0, 1e6]), 400, 1e-9, call, fields, iterationsPerCheck=8)
    fz = s._fused
    assert fz is not None
    # throughput: all profiles forced active
    s.m, s.dt = 1, 1e6
    q = fz.params(s)
    q.fixed_count = 1
    st = H.stream()
    step = lambda: H.check(H.lib.tg_picard_3f_step(C.byref(fz.state), C.byref(q), st))
    for _ in range(3):
        step()
    ms = timed(step, reps)
    report('c3 three-field', nb, n, 216, ms, invalid=int(np.count_nonzero(fz.status.to_host() & 6)))
    # time to convergence, masks on (fresh solver)
    lms2 = [tango.lodestro_method.lm(0.3, 0.3, th) for _ in range(3)]
    for f, m_ in zip(fields, lms2):
        f.lodestroMethod = m_
    s2 = tango.solver.Solver(1, x, np.array([0, 1e6]), 400, 1e-9, call, fields, iterationsPerCheck=8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s2.take_timestep()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    it = s2.iterationsPerProfile
    print(json.dumps({'config': 'c3 three-field to convergence (tol 1e-9, masks on)', 'profiles': nb, 'N': n, 'wall_s': round(wall, 4),
                      'converged': bool(s2.converged), 'iterations_min_max': [int(it.min()), int(it.max())]}), flush=True)


def single_profile(n=500):
    """The reference's own example (examples/shestakov_basic.py: one profile, N = 500, tol 1e-11) through the Solver
    API: wall time to convergence, with the convergence test on the host every iteration and every 8 iterations."""
    sh = tango.extras.shestakov_nonlinear_diffusion
    dx = 1.0 / (n - 1)
    x = np.arange(n) * dx
    for k in (1, 8):
        walls = []
        for rep in range(3):
            lm = tango.lodestro_method.lm(0.3, 1.0, {'Dmin': 1e-5, 'Dmax': 1e13, 'dpdxThreshold': 10})
            f = tango.multifield.Field(label='n', rightBC=1e-2, profile_mminus1=1 - 0.5 * x, compute_all_H=sh.ComputeAllH(1.0, 0.1),
                                       lodestroMethod=lm)
            tango.multifield.check_fields_initialize([f])
            call = tango.multifield.ComputeAllHAllFields([f], tango.lodestro_method.TurbulenceHandler(dx, x, sh.AnalyticFluxModel(dx)))
            s = tango.solver.Solver(1, x, np.array([0, 1e4]), 1000, 1e-11, call, [f], iterationsPerCheck=k)
            assert s._fused is not None
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            s.take_timestep()
            torch.cuda.synchronize()
            walls.append(time.perf_counter() - t0)
        its = len(s.errHistoryFinal)
        print(json.dumps({'config': 'c1 single profile through Solver (tol 1e-11)', 'profiles': 1, 'N': n, 'iterationsPerCheck': k,
                          'iterations': its, 'wall_ms': round(min(walls) * 1e3, 3), 'us_per_iteration': round(min(walls) * 1e6 / its, 1),
                          'converged': bool(s.converged)}), flush=True)


def single_profile_three_field(n=500):
    """examples/shestakov_threefield_nonlinearnu.py as one profile through the Solver API (tol 1e-9): wall time."""
    T = tango.extras.threefield
    dx = 1.0 / (n - 1)
    x = np.arange(n) * dx
    th = {'Dmin': 1e-5, 'Dmax': 1e13, 'dpdxThreshold': 10}
    walls = []
    for rep in range(3):
        lms = [tango.lodestro_method.lm(0.3, 0.3, th) for _ in range(3)]
        F = tango.multifield.Field
        ic = 1 - 0.5 * x
        fields = [F(label='n', rightBC=2.0, profile_mminus1=1.0 * ic, compute_all_H=T.ComputeAllH_n(), lodestroMethod=lms[0]),
                  F(label='pi', rightBC=0.1, profile_mminus1=1.0 * ic, compute_all_H=T.ComputeAllH_pi(2.2), lodestroMethod=lms[1], coupledTo='pe'),
                  F(label='pe', rightBC=0.3, profile_mminus1=1.0 * ic, compute_all_H=T.ComputeAllH_pe(2.2), lodestroMethod=lms[2], coupledTo='pi')]
        tango.multifield.check_fields_initialize(fields)
        call = tango.multifield.ComputeAllHAllFields(fields, tango.lodestro_method.TurbulenceHandler(dx, x, T.ShestakovThreeFieldFluxModel(dx)))
        s = tango.solver.Solver(1, x, np.array([0, 1e6]), 400, 1e-9, call, fields)
        assert s._fused is not None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.take_timestep()
        torch.cuda.synchronize()
        walls.append(time.perf_counter() - t0)
    its = len(s.errHistoryFinal)
    print(json.dumps({'config': 'three-field example, single profile through Solver (tol 1e-9)', 'profiles': 1, 'N': n, 'iterations': its,
                      'wall_ms': round(min(walls) * 1e3, 3), 'us_per_iteration': round(min(walls) * 1e6 / its, 1), 'converged': bool(s.converged)}),
          flush=True)


def convergence(nb=65536, n=512, k=64):
    st = torch.cuda.current_stream().cuda_stream
    E = OneFieldEnsemble(shestakov_ensemble(nb, n), a_turb=0.3, a_prof=1.0, tol=1e-11, fixed_count=False, keep_err=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    launches = 0
    while True:
        E.step(st, iters=k)
        launches += 1
        E.reduce_status(st)
        if int(E.counts.to_host()[0]) == 0 or launches * k >= 1000:
            break
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    it = E.iters.to_host()
    status = E.status.to_host()
    print(json.dumps({'config': f'metric config to convergence (tol 1e-11, masks on, {k} iterations per launch)', 'profiles': nb, 'N': n,
                      'wall_s': round(wall, 4), 'launches': launches, 'iterations_min_max': [int(it.min()), int(it.max())],
                      'profile_iterations': int(it.sum()), 'converged': int(np.count_nonzero(status & 1)),
                      'invalid': int(np.count_nonzero(status & 6))}), flush=True)


if __name__ == '__main__':
    what = sys.argv[1:] or ['c1', 'c2', 'metric', 'c3', 'c4', 'conv']
    H.set_stream(torch.cuda.current_stream().cuda_stream)
    if 'c1' in what:
        single_profile()
        single_profile_three_field()
    if 'c2' in what:
        one_field('c2 one-field', 65536, 256, 1.0, (1, 8))
    if 'metric' in what:
        one_field('metric one-field', 65536, 512, 1.0, (1, 4, 8, 16))
    if 'c4' in what:
        one_field('c4 one-field fine grid, EWMA 0.3/0.3', 4096, 8192, 0.3, (1,), reps=12)
    if 'c3' in what:
        three_field()
    if 'conv' in what:
        convergence()
