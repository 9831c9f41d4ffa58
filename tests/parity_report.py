"""Measured parity figures (not bars) of the fused one-field step against the reference's arithmetic, by grid size.

    python tests/parity_report.py [out.json]          (needs a GPU; a script, not collected by pytest)

For N in 256 ... 8192 the example problem of examples/shestakov_basic.py (EWMA 0.3/0.3, dt = 1e4) is stepped 50
Picard iterations three ways: by the fused HIP kernel; by the CPU restatement with the reference's own solve (SciPy ->
LAPACK dgtsv); and by the CPU restatement with every tridiagonal system solved in extended precision (x87 long double
Thomas sweeps), i.e. the trajectory the reference would follow if its solver made no rounding error.  Reported per
iterate (0, 10, last) for the profile u and the EWMA memories D, c:
  max   = max|a - b| / max|b|                    (the norm of the 1e-10 parity bar)
  elem  = max over points of |a - b| / |b|       (element-wise; points with |b| < 1e-300 left out)
  tail  = max|a - b| / max|b| over the last quarter of the grid (where the profile is 100 times smaller than at x = 0)
for the pairs gpu/lapack, gpu/exact and lapack/exact.  Also: how many members of the seeded 32-profile ensemble and
of a 16-member three-field ensemble stop on exactly the reference's iteration."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle import tango_oracle as orc          # noqa: E402  (checker)


def thomas_ld(A, B, C, D):
    A, B, C, D = (np.asarray(v, dtype=np.longdouble) for v in (A, B, C, D))
    n = len(B)
    cp = np.empty(n, np.longdouble)
    dp = np.empty(n, np.longdouble)
    cp[0] = A[0] / B[0]
    dp[0] = D[0] / B[0]
    for j in range(1, n):
        den = B[j] - C[j] * cp[j - 1]
        cp[j] = A[j] / den
        dp[j] = (D[j] - C[j] * dp[j - 1]) / den
    u = np.empty(n, np.longdouble)
    u[-1] = dp[-1]
    for j in range(n - 2, -1, -1):
        u[j] = dp[j] - cp[j] * u[j + 1]
    return u.astype(np.float64)


def errs(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    d = np.abs(a - b)
    ok = np.abs(b) > 1e-300
    q = len(b) - len(b) // 4
    return {'max': float(d.max() / np.abs(b).max()), 'elem': float((d[ok] / np.abs(b[ok])).max()),
            'tail': float(d[q:].max() / np.abs(b[q:]).max())}


def one_field(N, its=50):
    from tango_amd.ensemble import OneFieldEnsemble
    x = np.arange(N) * (1 / (N - 1))
    ens = dict(x=x, dx=1 / (N - 1), u0=(1 - 0.5 * x)[None], nL=np.array([1e-2]), S0=np.array([1.0]), delta=np.array([0.1]))
    E = OneFieldEnsemble(ens, a_prof=0.3, max_iter=its, fixed_count=True)
    H1, H7 = np.ones(N), orc.shestakov_source(x)
    lap, exa = orc.State1F(ens['u0'][0]), orc.State1F(ens['u0'][0])
    out = {}
    for it in range(its):
        E.step()
        orc.picard_step_1f(lap, 1e4, ens['dx'], 1e-2, H1, H7, 0.3, 0.3)
        orc.picard_step_1f(exa, 1e4, ens['dx'], 1e-2, H1, H7, 0.3, 0.3, solve=thomas_ld)
        if it in (0, 10, its - 1):
            g = {'u': E.u.to_host()[0], 'D': E.DE.to_host()[0], 'c': E.cE.to_host()[0]}
            row = {}
            for k, (l, e) in {'u': (lap.u, exa.u), 'D': (lap.DE, exa.DE), 'c': (lap.cE, exa.cE)}.items():
                row[k] = {'gpu_vs_lapack': errs(g[k], l), 'gpu_vs_exact': errs(g[k], e), 'lapack_vs_exact': errs(l, e)}
            out[f'iterate_{it}'] = row
    return out


def ensemble32():
    import tango_amd as tango
    from test_gpu_picard import make_solver
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g10_ensemble32.npz'))
    ens = orc.shestakov_ensemble(32, 512)
    s, _ = make_solver(tango, 512, ens['nL'], ens['S0'], ens['delta'], ens['u0'], 0.3, 1.0, 1e-11, 100, iterationsPerCheck=7)
    s.take_timestep()
    got = np.asarray(s.iterationsPerProfile, int)
    ref = np.asarray(g['iters'], int)
    worst = max(errs(s.profiles['n'][b], g['finals'][b])['elem'] for b in range(32))
    return {'members': 32, 'stop_on_the_reference_iteration': int(np.sum(got == ref)), 'differences': (got - ref)[got != ref].tolist(),
            'worst_elementwise_error_of_final_profiles': worst}


def three_field():
    import tango_amd as tango
    from test_gpu_coupled import make_3f_solver
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g9_threefield.npz'))
    out = {}
    for name, inner, key in (('example', 0, 'N500_iters'), ('example_inner_iteration', 2, 'inner_iters')):
        s, _ = make_3f_solver(tango, 500, 2.2, (2, 0.1, 0.3), inner=inner)
        while s.ok:
            s.take_timestep()
        pre = 'N500' if inner == 0 else 'inner'
        out[name] = {'iterations': len(s.errHistoryFinal), 'reference_iterations': int(g[key]),
                     'final_profiles': {lab: errs(s.profiles[lab], g[f'{pre}_{lab}']) for lab in ('n', 'pi', 'pe')}}
    N, nb = 512, 16
    ens = orc.threefield_ensemble(nb, N)
    nu0, bc = ens['nu0'], ens['bc']
    s, _ = make_3f_solver(tango, N, nu0, (bc[:, 0].copy(), bc[:, 1].copy(), bc[:, 2].copy()), ic=ens['u0'], iterationsPerCheck=5)
    s.take_timestep()
    diffs, worst = [], 0.0
    for b in range(nb):
        (n, pi, pe), it, err = orc.run_3f(ens['u0'][b], ens['u0'][b], ens['u0'][b], 1e6, ens['dx'], tuple(bc[b]), nu0[b], ens['x'], 0.3, 0.3, 1e-9, 1000)
        diffs.append(int(s.iterationsPerProfile[b]) - it)
        if diffs[-1] == 0:
            worst = max([worst] + [errs(s.profiles[lab][b], ref)['elem'] for lab, ref in (('n', n), ('pi', pi), ('pe', pe))])
    out['ensemble16'] = {'members': nb, 'stop_on_the_reference_iteration': int(np.sum(np.array(diffs) == 0)), 'differences': [d for d in diffs if d],
                         'worst_elementwise_error_of_final_profiles_same_stop': worst}
    return out


def main():
    import tango_amd
    assert tango_amd.device_count() > 0
    rep = {'one_field_by_grid_size': {str(N): one_field(N) for N in (256, 512, 1024, 2048, 4096, 8192)},
           'ensemble32': ensemble32(), 'three_field': three_field()}
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'gpurun_out', 'parity_report.json')
    json.dump(rep, open(path, 'w'), indent=1)
    for N, r in rep['one_field_by_grid_size'].items():
        last = r[sorted(r, key=lambda k: int(k.split('_')[1]))[-1]]
        print(f"N={N:>5}  last iterate  u: gpu/lapack {last['u']['gpu_vs_lapack']['max']:.1e} (elem {last['u']['gpu_vs_lapack']['elem']:.1e}, "
              f"tail {last['u']['gpu_vs_lapack']['tail']:.1e})  gpu/exact {last['u']['gpu_vs_exact']['max']:.1e}  lapack/exact "
              f"{last['u']['lapack_vs_exact']['max']:.1e} | D: gpu/lapack {last['D']['gpu_vs_lapack']['max']:.1e} elem "
              f"{last['D']['gpu_vs_lapack']['elem']:.1e} | c elem {last['c']['gpu_vs_lapack']['elem']:.1e}")
    print(rep['ensemble32'])
    print(json.dumps(rep['three_field']))


if __name__ == '__main__':
    main()
