"""Generate the golden vectors under tests/golden/ by importing the reference itself.

Run in the build container only (needs the read-only reference checkout):

    python tests/golden/make_golden.py [/root/reference]

The reference is pure Python; it is imported from where it lies, nothing is copied.  The
package's __init__ pulls in HDF5 I/O modules that need h5py, which this image lacks; the
hot-path modules do not use it, so an empty module object named h5py is registered before the
import (no HDF5 call is ever made here).  Every .npz holds inputs AND the reference's outputs,
so the tests that consume them never need the reference.
"""
import os
import sys
import types

import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.modules.setdefault('h5py', types.ModuleType('h5py'))
sys.path.insert(0, REF)

import tango  # noqa: E402
from tango import (tridiag, HToMatrixFD, JK_to_matrix_fd, lodestro_method, derivatives,  # noqa: E402
                   smoother, multifield, fieldgroups, solver as tsolver)
from tango.extras import shestakov_nonlinear_diffusion as shest  # noqa: E402



def save(name, **kw):
    np.savez_compressed(os.path.join(OUT, name + '.npz'), **kw)
    print('wrote', name, {k: np.shape(v) for k, v in kw.items()})


# ---------------------------------------------------------------- G1 tridiagonal solves
def g1():
    out = {}
    N = 12
    B = -np.ones(N); A = 0.1 * np.ones(N); C = 0.1 * np.ones(N)
    A[-1] = 0; C[0] = 0
    D = np.linspace(0, 0.22, N)
    out.update(t12_A=A, t12_B=B, t12_C=C, t12_D=D, t12_u=tridiag.solve(A, B, C, D),
               t12_u_py=tridiag.solve_python(A, B, C, D))
    rng = np.random.default_rng(1)
    for n in (8, 500, 8192):
        A = rng.uniform(-1, 1, n); C = rng.uniform(-1, 1, n)
        B = np.abs(A) + np.abs(C) + rng.uniform(0.01, 1.0, n)
        B *= rng.choice([-1.0, 1.0])
        A[-1] = 0; C[0] = 0
        D = rng.normal(size=n)
        out.update({f'r{n}_A': A, f'r{n}_B': B, f'r{n}_C': C, f'r{n}_D': D,
                    f'r{n}_u': tridiag.solve_with_scipy(A, B, C, D),
                    f'r{n}_u_py': tridiag.solve_python(A, B, C, D)})
    save('g1_tridiag', **out)


# ---------------------------------------------------------------- G2 assembly
def g2():
    out = {}
    # the four setups of tests/test_HToMatrixFD.py
    N = 500; dx = 1 / (N - 1); x = np.arange(N) * dx
    U = np.sin(np.pi * x) + 0.2 * x
    H1 = np.ones_like(x); H2 = 0.8 * np.ones_like(x); H7 = 1 - x ** 2
    A, B, C, D = HToMatrixFD.H_to_matrix(1e5, dx, 0.2, U, H1=H1, H2=H2, H7=H7)
    out.update(cart_dt=1e5, cart_dx=dx, cart_UL=0.2, cart_U=U, cart_H1=H1, cart_H2=H2, cart_H7=H7,
               cart_A=A, cart_B=B, cart_C=C, cart_D=D, cart_u=HToMatrixFD.solve(A, B, C, D))
    dr = 1 / (N - 0.5); r = np.linspace(dr / 2, 1, N)
    U = 0.3 * np.sin(np.pi * r) + 0.03 * r
    A, B, C, D = HToMatrixFD.H_to_matrix(1e6, dr, 0.03, U, H1=r, H2=1.3 * r, H7=r ** 2)
    out.update(polar_dt=1e6, polar_dx=dr, polar_UL=0.03, polar_U=U, polar_H1=r, polar_H2=1.3 * r,
               polar_H7=r ** 2, polar_A=A, polar_B=B, polar_C=C, polar_D=D,
               polar_u=HToMatrixFD.solve(A, B, C, D))
    N3 = 300; dr = 1 / (N3 - 0.5); r = np.linspace(dr / 2, 1, N3)
    U = 0.5 * np.ones_like(r) + 0.01 * np.cos(3 * r)
    H3 = -0.7 * r ** 2
    H7 = r / 1.3 * 0.9 * np.exp(-0.9 * 0.05) * (1 - r ** 5) + 25 * r ** 4 * (1 - np.exp(-0.9 * 0.05))
    A, B, C, D = HToMatrixFD.H_to_matrix(1e-2, dr, 0.5, U, H1=r, H2=1.3 * r, H3=H3, H7=H7)
    out.update(conv_dt=1e-2, conv_dx=dr, conv_UL=0.5, conv_U=U, conv_H1=r, conv_H2=1.3 * r,
               conv_H3=H3, conv_H7=H7, conv_A=A, conv_B=B, conv_C=C, conv_D=D,
               conv_u=HToMatrixFD.solve(A, B, C, D))
    U = np.sin(np.pi * x) + 2.2 * x
    H4 = -np.sin(x)
    A, B, C, D = HToMatrixFD.H_to_matrix(1e9, dx, 2.2, U, H1=np.ones_like(x), H2=np.ones_like(x),
                                         H4=H4, H7=1 - x ** 2)
    out.update(h4_dt=1e9, h4_dx=dx, h4_UL=2.2, h4_U=U, h4_H1=np.ones_like(x), h4_H2=np.ones_like(x),
               h4_H4=H4, h4_H7=1 - x ** 2, h4_A=A, h4_B=B, h4_C=C, h4_D=D,
               h4_u=HToMatrixFD.solve(A, B, C, D))
    # seeded random coefficients hitting every upwinding branch, all H present
    rng = np.random.default_rng(2)
    n = 257; dxr = 0.013
    H1 = rng.uniform(0.5, 2, n)
    H2 = 10.0 ** rng.uniform(-8, 2, n)
    H3 = rng.normal(size=n) * 10.0 ** rng.uniform(-3, 2, n)
    H3[40:60] = 0.0
    H3[100:110] = -H3[100:110]
    H4 = rng.normal(size=n); H6 = -rng.uniform(0, 3, n); H7 = rng.normal(size=n)
    U = rng.uniform(0.1, 2, n)
    A, B, C, D = HToMatrixFD.H_to_matrix(0.37, dxr, 1.7, U, H1, H2, H3, H4, H6, H7)
    out.update(rnd_dt=0.37, rnd_dx=dxr, rnd_UL=1.7, rnd_U=U, rnd_H1=H1, rnd_H2=H2, rnd_H3=H3,
               rnd_H4=H4, rnd_H6=H6, rnd_H7=H7, rnd_A=A, rnd_B=B, rnd_C=C, rnd_D=D)
    H2h = HToMatrixFD._return_average_on_half_integer_grid(H2)
    H3h = HToMatrixFD._return_average_on_half_integer_grid(H3)
    out.update(rnd_H2h=H2h, rnd_H3h=H3h, rnd_beta=HToMatrixFD._fbeta(H2h, H3h, dxr))
    # H3 without H2 (H2 defaults to zero inside the upwinding)
    A, B, C, D = HToMatrixFD.H_to_matrix(0.37, dxr, 1.7, U, H1, H3=H3, H7=H7)
    out.update(h3only_A=A, h3only_B=B, h3only_C=C, h3only_D=D)
    save('g2_assembly', **out)


# ---------------------------------------------------------------- G3 pair-coupled
def two_field_setup():
    N = 500; dx = 1 / (N - 0.5); x = np.linspace(dx / 2, 1, N)
    DU, DW, cU, cW, nu, alpha, beta, UL, WL = 1.6, 2.4, 0.3, 9, 1.3, 1.9, 0.09, 0.75, 3.33
    J2 = DU * x; K2 = DW * x; J3 = -cU * x ** 2; K3 = -cW * x ** 2
    J6 = x; K6 = x ** 2; J8 = nu * x; K8 = -nu * x
    pi, sin, cos = np.pi, np.sin, np.cos
    J7 = (9 * alpha * DU * x ** 2 + 2 * cU * x * UL + cU * alpha * (2 * x - 5 * x ** 4)
          - x * UL - alpha * (x - x ** 4) - nu * x * WL - nu * beta * x * cos(pi * x / 2))
    K7 = (pi / 2 * DW * beta * (sin(pi * x / 2) + pi * x / 2 * cos(pi * x / 2)) + 2 * cW * x * WL
          + cW * beta * (2 * x * cos(pi * x / 2) - pi * x ** 2 / 2 * sin(pi * x / 2)) - x ** 2 * WL
          - beta * x ** 2 * cos(pi * x / 2) + nu * x * UL + nu * alpha * (x - x ** 4))
    J1 = np.ones_like(x); K1 = np.ones_like(x)
    return dict(dt=1e9, dx=dx, UL=UL, WL=WL, x=x, J1=J1, J2=J2, J3=J3, J6=J6, J7=J7, J8=J8,
                K1=K1, K2=K2, K3=K3, K6=K6, K7=K7, K8=K8,
                Uan=UL + alpha * (1 - x ** 3), Wan=WL + beta * cos(pi * x / 2))


def g3():
    s = two_field_setup()
    z = np.zeros_like(s['x'])
    Ue, We = JK_to_matrix_fd.JK_to_matrix(s['dt'], s['dx'], (s['UL'], s['WL']), (z, z),
                                          J1=s['J1'], J2=s['J2'], J3=s['J3'], J6=s['J6'], J7=s['J7'], J8=s['J8'],
                                          K1=s['K1'], K2=s['K2'], K3=s['K3'], K6=s['K6'], K7=s['K7'], K8=s['K8'])
    U, W = JK_to_matrix_fd.solve(Ue, We)
    # residual of a perturbed pair with the same matrix (G11, pair part)
    pfg = fieldgroups.PairCoupledFieldGroup('a', 'b')
    meq = fieldgroups.PairCoupledMatrixEqn(Ue, We)
    prof = {'a': U * (1 + 0.01 * np.sin(7 * s['x'])), 'b': W * (1 - 0.02 * np.cos(5 * s['x']))}
    res = pfg.calculate_residual(meq, prof)
    save('g3_pair', **s, AU=Ue.A, BU=Ue.B, CU=Ue.C, DU=Ue.D, GW=Ue.G,
         AW=We.A, BW=We.B, CW=We.C, DW=We.D, GU=We.G, U=U, W=W,
         res_U_in=prof['a'], res_W_in=prof['b'], res_U=res['a'], res_W=res['b'])


# ---------------------------------------------------------------- G4 flux split
def g4():
    out = {}
    N = 500; dx = 1 / (N - 1); x = np.arange(N) * dx
    th = {'Dmin': 1e-5, 'Dmax': 1e13, 'dpdxThreshold': 10}
    fs = lodestro_method.FluxSplit(th)
    for tag, p in (('it0', 1 - 0.5 * x), ('it20', shest.steady_state_solution(x, 1e-2) * (1 + 0.05 * np.sin(9 * x)))):
        flux = shest.get_flux(p, dx)
        D, c, d = fs.flux_to_transport_coeffs(flux, p, dx)
        out.update({f'{tag}_p': p, f'{tag}_flux': flux, f'{tag}_D': D, f'{tag}_c': c,
                    f'{tag}_DHat': d['DHat'], f'{tag}_cHat': d['cHat'], f'{tag}_theta': d['theta']})
    out.update(shest_dx=dx, shest_Dmin=1e-5, shest_Dmax=1e13, shest_thr=10.0)
    # synthetic inputs reaching every theta branch and dpdx == 0
    rng = np.random.default_rng(4)
    n = 200; dxs = 0.05
    p = np.cumsum(rng.normal(size=n) * 0.3) + 5
    p[50:54] = p[50]            # flat stretch -> dpdx == 0 at 51, 52
    flux = rng.normal(size=n) * 10.0 ** rng.uniform(-3, 2, n)
    th2 = {'Dmin': 0.1, 'Dmax': 10.0, 'dpdxThreshold': 1.0}
    gxx = rng.uniform(0.5, 2, n); gradx = rng.uniform(0.5, 2, n)
    for tag, kw in (('syn', {}), ('geo', dict(gxxAvg=gxx, gradxAvg=gradx))):
        fs2 = lodestro_method.FluxSplit(th2, **kw)
        with np.errstate(all='ignore'):
            D, c, d = fs2.flux_to_transport_coeffs(flux, p, dxs)
        out.update({f'{tag}_D': D, f'{tag}_c': c, f'{tag}_DHat': d['DHat'], f'{tag}_cHat': d['cHat'],
                    f'{tag}_theta': d['theta']})
    out.update(syn_p=p, syn_flux=flux, syn_dx=dxs, syn_Dmin=0.1, syn_Dmax=10.0, syn_thr=1.0,
               geo_gxx=gxx, geo_gradx=gradx)
    save('g4_fluxsplit', **out)


# ---------------------------------------------------------------- G5-G7 small known answers
def g567():
    e = lodestro_method.EWMA(0.12)
    seq = [1.2, 2.2, -0.3, 3.4]
    outs = [e.next_ewma_iterate(v) for v in seq]
    rng = np.random.default_rng(6)
    xs = rng.normal(size=64)
    sm = {f'sm_w{w}': smoother.Smoother(w).smooth(xs) for w in (1, 3, 4, 5, 11)}
    x9 = np.array([1., 2., 9., 4., 10., 3., 1., 7., 5.])
    u7 = np.array([2, 3, 7, 5, -3, 2, 8], dtype=float)
    save('g567_small', ewma_param=0.12, ewma_in=np.array(seq), ewma_out=np.array(outs),
         ewma_single=lodestro_method.EWMA._compute_next_ewma(37, 7.1, 0.12),
         sm_x=xs, **sm, sm9_x=x9, sm9_w5=smoother.moving_average(x9, 5),
         d_u=u7, d_dx=0.4, d_edge1=derivatives.dx_centered_difference_edge_first_order(u7, 0.4),
         d_edge2=derivatives.dx_centered_difference(u7, 0.4))


# ---------------------------------------------------------------- G8/G10 one-field trajectories
class ComputeAllH1F(object):
    def __init__(self, S0, delta):
        self.S0, self.delta = S0, delta

    def __call__(self, t, x, profiles, HCoeffsTurb):
        H1 = np.ones_like(x)
        H7 = shest.GetSource(x, S0=self.S0, delta=self.delta)
        return multifield.HCoefficients(H1=H1, H7=H7) + HCoeffsTurb


def run_ref_1f(N, nL, S0, delta, n0, a_turb, a_prof, tol, maxit, dt=1e4, keep=(), **solver_kw):
    dx = 1 / (N - 1); x = np.arange(N) * dx
    th = {'Dmin': 1e-5, 'Dmax': 1e13, 'dpdxThreshold': 10}
    fm = shest.AnalyticFluxModel(dx)
    turb = lodestro_method.TurbulenceHandler(dx, x, fm)
    lm = lodestro_method.lm(a_turb, a_prof, th)
    f0 = multifield.Field(label='n', rightBC=nL, profile_mminus1=n0.copy(), compute_all_H=ComputeAllH1F(S0, delta),
                          lodestroMethod=lm)
    fields = [f0]
    multifield.check_fields_initialize(fields)
    call = multifield.ComputeAllHAllFields(fields, turb)
    s = tsolver.Solver(1, x, np.array([0, dt]), maxit, tol, call, fields, **solver_kw)
    snaps = {}
    k = 0
    # drive the reference's own loop one iteration at a time so that EWMA states can be recorded
    s.m += 1; s.converged = False; (s.t, s.dt) = s._get_tdt(s.tArray, s.m)
    s.errHistory[:] = 0
    s.profilesAllIterations = {'n': np.zeros((maxit, N))}
    s.fluxesAllIterations = {'n': np.zeros((maxit, N))}
    while s.iterationNumber < s.maxIterations and not s.converged and not s.solutionError:
        s.compute_next_iteration()
        if k in keep:
            snaps[k] = (s.profiles['n'].copy(), lm._EWMATurbD._yEWMA_lminus1.copy(),
                        lm._EWMATurbc._yEWMA_lminus1.copy(), lm._EWMAProfile._yEWMA_lminus1.copy())
        k += 1
    last = (s.profiles['n'].copy(), lm._EWMATurbD._yEWMA_lminus1.copy(),
            lm._EWMATurbc._yEWMA_lminus1.copy(), lm._EWMAProfile._yEWMA_lminus1.copy())
    return s.profiles['n'].copy(), k, s.errHistory[:k].copy(), snaps, last, bool(s.converged)


def g8():
    out = {}
    for N in (256, 500, 512):
        for tag, ap in (('p1', 1.0), ('p03', 0.3)):
            x = np.arange(N) * (1 / (N - 1))
            u, k, err, snaps, last, conv = run_ref_1f(N, 1e-2, 1.0, 0.1, 1 - 0.5 * x, 0.3, ap, 1e-11, 1000,
                                                      keep=(0, 1, 2, 10))
            key = f'N{N}_{tag}'
            out.update({f'{key}_iters': k, f'{key}_err': err, f'{key}_final': u, f'{key}_conv': conv})
            for it, (p, D, c, pe) in snaps.items():
                out.update({f'{key}_it{it}_u': p, f'{key}_it{it}_DE': D, f'{key}_it{it}_cE': c,
                            f'{key}_it{it}_uE': pe})
            out.update({f'{key}_last_DE': last[1], f'{key}_last_cE': last[2], f'{key}_last_uE': last[3]})
    # fine grid, fixed 50 iterations, EWMA 0.3/0.3
    N = 8192
    x = np.arange(N) * (1 / (N - 1))
    u, k, err, snaps, last, conv = run_ref_1f(N, 1e-2, 1.0, 0.1, 1 - 0.5 * x, 0.3, 0.3, 1e-30, 50, keep=(0, 10))
    out.update(N8192_iters=k, N8192_err=err, N8192_final=u, N8192_it0_u=snaps[0][0], N8192_it10_u=snaps[10][0],
               N8192_last_DE=last[1], N8192_last_cE=last[2], N8192_last_uE=last[3])
    save('g8_onefield', **out)


def g10():
    B, N = 32, 512
    rng = np.random.default_rng(12345)
    nL = rng.uniform(0.005, 0.02, size=B); S0 = rng.uniform(0.5, 2.0, size=B)
    delta = rng.uniform(0.05, 0.2, size=B); a = rng.uniform(0.8, 1.2, size=B)
    s = rng.uniform(0.3, 0.6, size=B) * a
    nL[0], S0[0], delta[0], a[0], s[0] = 1e-2, 1.0, 0.1, 1.0, 0.5
    x = np.arange(N) * (1 / (N - 1))
    finals = np.zeros((B, N)); iters = np.zeros(B, dtype=np.int64); errs = np.zeros((B, 100))
    for b in range(B):
        u, k, err, _, _, conv = run_ref_1f(N, nL[b], S0[b], delta[b], a[b] - s[b] * x, 0.3, 1.0, 1e-11, 1000)
        assert conv and k <= 100
        finals[b] = u; iters[b] = k; errs[b, :k] = err
    save('g10_ensemble32', nL=nL, S0=S0, delta=delta, a=a, s=s, finals=finals, iters=iters, errs=errs)


def g12():
    """Pseudo-timestep control (examples/shestakov_treadlightly.py, shestakov_treadinitially.py of the reference):
    nL = 1, n0 = -0.2 (x - 1) + 1.2, EWMA 0.1/1, tol 1e-10; every profile iterate is kept."""
    N = 500
    x = np.arange(N) * (1 / (N - 1))
    n0 = -0.2 * (x - 1) + 1.2
    out = dict(N=N, nL=1.0, n0=n0)
    cases = {'lightly': dict(useTreadLightly=True, treadLightlyParams={'maxFractionalChange': 0.2}),
             'lightly_tight': dict(useTreadLightly=True, treadLightlyParams={'maxFractionalChange': 0.05, 'innerFraction': 0.8}),
             'initially': dict(useTreadInitially=True, treadInitiallyParams={'numInitialSteps': 6, 'dtau': 0.3}),
             'both': dict(useTreadLightly=True, treadLightlyParams={'maxFractionalChange': 0.1},
                          useTreadInitially=True, treadInitiallyParams={'numInitialSteps': 4, 'dtau': 0.5})}
    for tag, kw in cases.items():
        keep = tuple(range(0, 40))
        u, k, err, snaps, last, conv = run_ref_1f(N, 1.0, 1.0, 0.1, n0, 0.1, 1.0, 1e-10, 1000, keep=keep, **kw)
        out.update({f'{tag}_iters': k, f'{tag}_err': err, f'{tag}_final': u, f'{tag}_conv': conv,
                    f'{tag}_first40': np.stack([snaps[i][0] for i in sorted(snaps)]),
                    f'{tag}_last_DE': last[1], f'{tag}_last_cE': last[2]})
        print(tag, 'iterations', k, 'converged', conv)
    save('g12_tread', **out)


def g13():
    """Several implicit timesteps of unequal length driven by the reference's own loop (tango/solver.py:117-186,
    :533-548): EWMA memories persist from one timestep to the next, u_{m-1} is the previous converged profile."""
    N = 128
    dx = 1 / (N - 1); x = np.arange(N) * dx
    th = {'Dmin': 1e-5, 'Dmax': 1e13, 'dpdxThreshold': 10}
    tArray = np.array([0.0, 0.02, 0.05, 0.1, 0.3])
    n0 = 1 - 0.5 * x
    lm = lodestro_method.lm(0.3, 1.0, th)
    f0 = multifield.Field(label='n', rightBC=1e-2, profile_mminus1=n0.copy(), compute_all_H=ComputeAllH1F(1.0, 0.1), lodestroMethod=lm)
    multifield.check_fields_initialize([f0])
    call = multifield.ComputeAllHAllFields([f0], lodestro_method.TurbulenceHandler(dx, x, shest.AnalyticFluxModel(dx)))
    s = tsolver.Solver(1, x, tArray, 400, 1e-10, call, [f0])
    iters, errs, profs = [], [], []
    while s.ok:
        s.take_timestep()
        iters.append(len(s.errHistoryFinal)); errs.append(s.errHistoryFinal.copy()); profs.append(s.profiles['n'].copy())
    E = np.zeros((len(errs), max(iters)))
    for k, e in enumerate(errs):
        E[k, :len(e)] = e
    print('timesteps', len(iters), 'iterations', iters, 'reachedEnd', s.reachedEnd)
    save('g13_timesteps', N=N, tArray=tArray, n0=n0, iters=np.array(iters), errs=E, profiles=np.stack(profs),
         reachedEnd=bool(s.reachedEnd), last_DE=lm._EWMATurbD._yEWMA_lminus1, last_uE=lm._EWMAProfile._yEWMA_lminus1)


class PowerLawFlux(object):
    """Flux model in the manner of examples/stripped_down_tango_example.py:55-65: Gamma = -D0 |n'/n|^p n'."""
    def __init__(self, dx, D0=0.7, p=1.5):
        self.dx, self.D0, self.p = dx, D0, p

    def get_flux(self, profiles):
        n = profiles['n']
        dn = derivatives.dx_centered_difference_edge_first_order(n, self.dx)
        return {'n': -self.D0 * np.abs(dn / n) ** self.p * dn}


class ComputeAllHGeom(object):
    def __init__(self, Vp, S):
        self.Vp, self.S = Vp, S

    def __call__(self, t, x, profiles, HCoeffsTurb):
        return multifield.HCoefficients(H1=1.0 * self.Vp, H7=self.Vp * self.S) + HCoeffsTurb


def g14():
    """One field in non-Cartesian geometry with a flux model supplied as a Python object and a smoothed flux: V',
    <g^xx>, <|grad x|> on the transport grid (TurbulenceHandler) and on the turbulence grid (lm), H1 = V',
    H7 = V' S (tango/lodestro_method.py:280-408,540-638)."""
    N = 200
    dx = 1 / (N - 1); x = np.arange(N) * dx
    Vp = 1.0 + x; gxx = 1.0 + 0.2 * x ** 2; gradx = np.sqrt(gxx)
    S = 1.0 * (x < 0.15)
    th = {'Dmin': 1e-5, 'Dmax': 1e13, 'dpdxThreshold': 10}
    out = dict(N=N, Vp=Vp, gxx=gxx, gradx=gradx, S=S)
    for tag, window in (('plain', None), ('smooth', 7)):
        lm = lodestro_method.lm(0.2, 0.6, th, gxxAvgTurb=gxx, gradxAvgTurb=gradx)
        sm = smoother.Smoother(window) if window else None
        turb = lodestro_method.TurbulenceHandler(dx, x, PowerLawFlux(dx), VprimeTango=Vp, fluxSmoother=sm, gxxAvgTango=gxx,
                                                 gradxAvgTango=gradx)
        f0 = multifield.Field(label='n', rightBC=0.05, profile_mminus1=1 - 0.6 * x, compute_all_H=ComputeAllHGeom(Vp, S), lodestroMethod=lm)
        multifield.check_fields_initialize([f0])
        # the smoothed flux does not settle below 1e-2 on this problem: compare a fixed 40 iterations of it
        s = tsolver.Solver(1, x, np.array([0, 50.0]), 40 if window else 600, 1e-10, multifield.ComputeAllHAllFields([f0], turb), [f0])
        while s.ok:
            s.take_timestep()
        k = len(s.errHistoryFinal)
        print(tag, 'iterations', k, 'converged', s.converged)
        out.update({f'{tag}_iters': k, f'{tag}_err': s.errHistoryFinal.copy(), f'{tag}_final': s.profiles['n'].copy(),
                    f'{tag}_it3': s.profilesAllIterations['n'][3].copy(), f'{tag}_last_DE': lm._EWMATurbD._yEWMA_lminus1,
                    f'{tag}_last_cE': lm._EWMATurbc._yEWMA_lminus1, f'{tag}_conv': bool(s.converged)})
    save('g14_geometry', **out)


def g15():
    """Flux-model decorators (tango/extras/noisyflux.py, bufferzone.py) with numpy's global generator seeded.  The
    reference's bufferzone calls scipy.signal.tukey, which this SciPy only has as scipy.signal.windows.tukey: the
    window itself is recorded."""
    import scipy.signal.windows
    from tango.extras import noisyflux
    out = {}
    np.random.seed(11)
    out['ar1_300_7p5'] = noisyflux.ar1noise(300, 7.5)
    out['ar1_64_3'] = noisyflux.ar1noise(64, 3.0)
    v = np.linspace(1.0, 2.0, 200) ** 2
    out['v'] = v
    out['dampen_default'] = noisyflux.dampen_sides(v)
    out['dampen_10'] = noisyflux.dampen_sides(v, 10)
    np.random.seed(12)
    out['noisy'] = noisyflux.NoisyFlux._add_noise(v, 0.1, 7.5)
    out['uniform'] = noisyflux.UniformNoisyFlux._add_noise(v, 0.2)
    out['tukey_200_0p3'] = scipy.signal.windows.tukey(200, 0.3)
    out['tukey_51_0p5'] = scipy.signal.windows.tukey(51, 0.5)
    out['tukey_40_1'] = scipy.signal.windows.tukey(40, 1.0)
    save('g15_decorators', **out)


def g16():
    """Grid mapping with the transport grid outside the turbulence grid (tango/interfacegrids_gene.py:35-87): the
    class's three maps on sample data, and a one-field run whose flux model lives on a finer, shorter grid."""
    from tango import interfacegrids_gene as ig
    xT = np.linspace(0.0, 1.0, 60)
    xG = np.linspace(0.13, 0.81, 150)
    gm = ig.GridInterfaceTangoOutside(xT, xG)
    prof = 1.2 - 0.7 * xT ** 2 + 0.05 * np.sin(9 * xT)
    Dg = 0.3 * np.exp(-((xG - 0.45) / 0.2) ** 2) - 0.004
    cg = 0.1 * np.sin(7 * xG) * (xG - 0.13) * (0.81 - xG)
    out = dict(xT=xT, xG=xG, prof=prof, Dg=Dg, cg=cg, prof_on_G=gm.map_profile_onto_turb_grid(prof))
    out['D_on_T'], out['c_on_T'] = gm.map_transport_coeffs_onto_transport_grid(Dg, cg)
    xn = np.sort(np.concatenate(([0.0, 1.0], np.random.default_rng(3).uniform(0, 1, 38))))      # non-uniform grid
    yn = np.cos(3 * xn) + xn
    xq = np.linspace(-0.05, 1.05, 97)
    import scipy.interpolate
    out.update(xn=xn, yn=yn, xq=xq, yq=scipy.interpolate.InterpolatedUnivariateSpline(xn, yn)(xq))
    ge = ig.TangoOutsideExtrapCoeffs(xT, xG, 0.55, 0.7, 2)
    out['ext_prof_on_G'] = ge.map_profile_onto_turb_grid(prof)
    out['ext_D_on_T'], out['ext_c_on_T'] = ge.map_transport_coeffs_onto_transport_grid(Dg, cg)
    # a run: transport grid 0..1 (N=120), turbulence grid 0.1..0.9 (N=300); Shestakov flux on the turbulence grid
    N, NG = 120, 300
    x = np.linspace(0, 1, N); dx = x[1] - x[0]
    xg = np.linspace(0.1, 0.9, NG); dxg = xg[1] - xg[0]
    th = {'Dmin': 1e-5, 'Dmax': 1e13, 'dpdxThreshold': 10}
    lm = lodestro_method.lm(0.3, 1.0, th)
    f0 = multifield.Field(label='n', rightBC=0.05, profile_mminus1=1 - 0.5 * x, compute_all_H=ComputeAllH1F(1.0, 0.3),
                          gridMapper=ig.GridInterfaceTangoOutside(x, xg), lodestroMethod=lm)
    multifield.check_fields_initialize([f0])
    call = multifield.ComputeAllHAllFields([f0], lodestro_method.TurbulenceHandler(dxg, x, shest.AnalyticFluxModel(dxg)))
    s = tsolver.Solver(1, x, np.array([0, 0.5]), 25, 1e-14, call, [f0])
    s.take_timestep()
    out.update(run_N=N, run_NG=NG, run_err=s.errHistoryFinal.copy(), run_final=s.profiles['n'].copy(),
               run_it2=s.profilesAllIterations['n'][2].copy(), run_DE=lm._EWMATurbD._yEWMA_lminus1.copy())
    print('run iterations', len(s.errHistoryFinal), 'last err', s.errHistoryFinal[-1])
    save('g16_gridmap', **out)


def g17():
    """The remaining grid mappers (tango/interfacegrids_gene.py:181-245, :300-443, :528-692): extrapolation of the
    coefficients at both boundaries, and the turbulence grid reaching beyond the outer end of the transport grid.
    make_extrapolator_fixed_slope uses the alias numpy.float, which numpy 2 no longer has; it is put back (as the
    builtin float, what it always was) for the duration of this function."""
    from tango import interfacegrids_gene as ig
    had = hasattr(np, 'float')
    if not had:
        np.float = float
    try:
        xT = np.linspace(0.0, 1.0, 60)
        xG = np.linspace(0.13, 0.81, 150)
        Dg = 0.3 * np.exp(-((xG - 0.45) / 0.2) ** 2) - 0.004
        cg = 0.1 * np.sin(7 * xG) * (xG - 0.13) * (0.81 - xG)
        prof = 1.2 - 0.7 * xT ** 2 + 0.05 * np.sin(9 * xT)
        gb = ig.TangoOutsideExtrapCoeffsBothSides(xT, xG, 0.2, 0.32, 0.55, 0.7, 2)
        out = dict(xT=xT, xG=xG, Dg=Dg, cg=cg, prof=prof, both_prof_on_G=gb.map_profile_onto_turb_grid(prof))
        out['both_D_on_T'], out['both_c_on_T'] = gb.map_transport_coeffs_onto_transport_grid(Dg, cg)
        # transport grid 0..0.8, turbulence grid 0.1..1: it ends beyond the transport grid
        xTi = np.linspace(0.0, 0.8, 50)
        xGi = np.linspace(0.1, 1.0, 140)
        profi = 1.1 - 0.9 * xTi ** 2 + 0.03 * np.cos(11 * xTi)
        steep = 2.0 - 2.6 * xTi                      # continued outward it turns negative: exercises enforcePositive
        Dgi = 0.25 * np.exp(-((xGi - 0.5) / 0.25) ** 2) - 0.01
        cgi = 0.08 * np.sin(6 * xGi) * (xGi - 0.1)
        gi = ig.GridInterfaceTangoInside(xTi, xGi)
        out.update(xTi=xTi, xGi=xGi, profi=profi, steep=steep, Dgi=Dgi, cgi=cgi,
                   in_prof_on_G=gi.map_profile_onto_turb_grid(profi), in_steep_on_G=gi.map_profile_onto_turb_grid(steep))
        out['in_D_on_T'], out['in_c_on_T'] = gi.map_transport_coeffs_onto_transport_grid(Dgi, cgi)
        gf = ig.GridInterfaceTangoInsideFixedOutside(xTi, xGi, -1.7)
        out['fix_prof_on_G'] = gf.map_profile_onto_turb_grid(profi)
        out['fix_steep_on_G'] = gf.map_profile_onto_turb_grid(steep)
        out['fix_D_on_T'], out['fix_c_on_T'] = gf.map_transport_coeffs_onto_transport_grid(Dgi, cgi)
        out['slope'] = np.array(ig.least_squares_slope(xTi[-10:], profi[-10:], xTi[-1], profi[-1]))
        out['reg_left'] = ig.extrap1d_constrained_linear_regression(xTi, profi, np.linspace(-0.2, 0.1, 7), side='left', numPts=6)
        out['ext_left'] = ig.make_extrapolator(xTi, profi, 'left', 6)(np.linspace(-0.2, 0.5, 23))
    finally:
        if not had:
            del np.float
    save('g17_gridmap_more', **out)


# ---------------------------------------------------------------- G9 three-field trajectories
def g19():
    """One-field runs whose coefficients are extrapolated beyond the trusted part of the turbulence grid
    (tango/interfacegrids_gene.py:90-178 at the outer boundary, :181-245 at both): transport grid 0..1 (N=120),
    turbulence grid 0.1..0.9 (N=300), Shestakov flux on the turbulence grid, as in g16."""
    from tango import interfacegrids_gene as ig
    N, NG = 120, 300
    x = np.linspace(0, 1, N)
    xg = np.linspace(0.1, 0.9, NG); dxg = xg[1] - xg[0]
    th = {'Dmin': 1e-5, 'Dmax': 1e13, 'dpdxThreshold': 10}
    out = dict(run_N=N, run_NG=NG)
    for tag, mapper in (('outer', ig.TangoOutsideExtrapCoeffs(x, xg, 0.6, 0.8, 1)),
                        ('both', ig.TangoOutsideExtrapCoeffsBothSides(x, xg, 0.2, 0.35, 0.6, 0.8, 2))):
        lm = lodestro_method.lm(0.3, 1.0, th)
        f0 = multifield.Field(label='n', rightBC=0.05, profile_mminus1=1 - 0.5 * x, compute_all_H=ComputeAllH1F(1.0, 0.3),
                              gridMapper=mapper, lodestroMethod=lm)
        multifield.check_fields_initialize([f0])
        call = multifield.ComputeAllHAllFields([f0], lodestro_method.TurbulenceHandler(dxg, x, shest.AnalyticFluxModel(dxg)))
        s = tsolver.Solver(1, x, np.array([0, 0.5]), 25, 1e-14, call, [f0])
        s.take_timestep()
        out[tag + '_err'] = s.errHistoryFinal.copy()
        out[tag + '_final'] = s.profiles['n'].copy()
        out[tag + '_it2'] = s.profilesAllIterations['n'][2].copy()
        out[tag + '_DE'] = lm._EWMATurbD._yEWMA_lminus1.copy()
        print(tag, 'iterations', len(s.errHistoryFinal), 'last err', s.errHistoryFinal[-1])
    save('g19_gridmap_extrap_runs', **out)


def g20():
    """A one-field run with a turbulence grid that reaches beyond the outer end of the transport grid
    (tango/interfacegrids_gene.py:367-417): transport grid 0..0.8 (N=100), turbulence grid 0.1..1 (N=280), Shestakov
    flux on the turbulence grid; and the profile map alone for a profile whose continuation turns negative."""
    from tango import interfacegrids_gene as ig
    N, NG = 100, 280
    x = np.linspace(0, 0.8, N)
    xg = np.linspace(0.1, 1.0, NG); dxg = xg[1] - xg[0]
    th = {'Dmin': 1e-5, 'Dmax': 1e13, 'dpdxThreshold': 10}
    lm = lodestro_method.lm(0.3, 1.0, th)
    mapper = ig.GridInterfaceTangoInside(x, xg)
    f0 = multifield.Field(label='n', rightBC=0.25, profile_mminus1=1 - 0.9 * x, compute_all_H=ComputeAllH1F(1.0, 0.3),
                          gridMapper=mapper, lodestroMethod=lm)
    multifield.check_fields_initialize([f0])
    call = multifield.ComputeAllHAllFields([f0], lodestro_method.TurbulenceHandler(dxg, x, shest.AnalyticFluxModel(dxg)))
    s = tsolver.Solver(0.8, x, np.array([0, 0.5]), 25, 1e-14, call, [f0])
    s.take_timestep()
    print('iterations', len(s.errHistoryFinal), 'last err', s.errHistoryFinal[-1])
    steep = 2.0 - 2.4 * x
    # the same problem with an imposed slope of the profile beyond the transport grid (:419-443); numpy.float as in g17
    had = hasattr(np, 'float')
    if not had:
        np.float = float
    try:
        lm2 = lodestro_method.lm(0.3, 1.0, th)
        mapper2 = ig.GridInterfaceTangoInsideFixedOutside(x, xg, -1.2)
        f2 = multifield.Field(label='n', rightBC=0.25, profile_mminus1=1 - 0.9 * x, compute_all_H=ComputeAllH1F(1.0, 0.3),
                              gridMapper=mapper2, lodestroMethod=lm2)
        multifield.check_fields_initialize([f2])
        call2 = multifield.ComputeAllHAllFields([f2], lodestro_method.TurbulenceHandler(dxg, x, shest.AnalyticFluxModel(dxg)))
        s2 = tsolver.Solver(0.8, x, np.array([0, 0.5]), 25, 1e-14, call2, [f2])
        s2.take_timestep()
        print('fixed slope: iterations', len(s2.errHistoryFinal), 'last err', s2.errHistoryFinal[-1])
        fixed = dict(fix_err=s2.errHistoryFinal.copy(), fix_final=s2.profiles['n'].copy(), fix_it2=s2.profilesAllIterations['n'][2].copy(),
                     fix_DE=lm2._EWMATurbD._yEWMA_lminus1.copy(), fix_uE=lm2._EWMAProfile._yEWMA_lminus1.copy(),
                     fix_steep_on_G=mapper2.map_profile_onto_turb_grid(steep))
    finally:
        if not had:
            del np.float
    save('g20_gridmap_inside_run', **fixed, run_N=N, run_NG=NG, run_err=s.errHistoryFinal.copy(), run_final=s.profiles['n'].copy(),
         run_it2=s.profilesAllIterations['n'][2].copy(), run_DE=lm._EWMATurbD._yEWMA_lminus1.copy(),
         run_uE=lm._EWMAProfile._yEWMA_lminus1.copy(), steep=steep, steep_on_G=mapper.map_profile_onto_turb_grid(steep))


def run_ref_3f(N, nu0, BCs, a_turb, a_prof, tol, maxit, inner=0, dt=1e6, keep=(), ic=None, **solver_kw):
    dx = 1 / (N - 1); x = np.arange(N) * dx
    th = {'Dmin': 1e-5, 'Dmax': 1e13, 'dpdxThreshold': 10}

    class Flux3(object):
        def get_flux(self, profiles):
            n, pi, pe = profiles['n'], profiles['pi'], profiles['pe']
            d = derivatives.dx_centered_difference_edge_first_order
            dn, dpi, dpe = d(n, dx), d(pi, dx), d(pe, dx)
            D = dpi ** 2 / pi ** 2
            return {'n': -D * dn, 'pi': -D * dpi, 'pe': -D * dpe}

    def src(S0, delta, lin=False):
        S = np.zeros_like(x); ind = x < delta
        S[ind] = S0 * x[ind] if lin else S0
        return S

    def nu_of(p):
        return nu0 / ((p['pi'] / p['n'] + p['pe'] / p['n']) ** (3 / 2))

    class Hn(object):
        def __call__(self, t, xx, profiles, HT):
            return multifield.HCoefficients(H1=np.ones_like(xx), H7=src(8, 0.3)) + HT

    class Hp(object):
        def __init__(self, S):
            self.S = S

        def __call__(self, t, xx, profiles, HT):
            nu = nu_of(profiles)
            return multifield.HCoefficients(H1=np.ones_like(xx), H6=-nu, H7=self.S, H8=nu) + HT

    ic = (1 - 0.5 * x) if ic is None else ic
    lms = [lodestro_method.lm(a_turb, a_prof, th) for _ in range(3)]
    f0 = multifield.Field(label='n', rightBC=BCs[0], profile_mminus1=1.0 * ic, compute_all_H=Hn(), lodestroMethod=lms[0])
    f1 = multifield.Field(label='pi', rightBC=BCs[1], profile_mminus1=1.0 * ic, compute_all_H=Hp(src(1, 0.1)),
                          lodestroMethod=lms[1], coupledTo='pe')
    f2 = multifield.Field(label='pe', rightBC=BCs[2], profile_mminus1=1.0 * ic, compute_all_H=Hp(src(3, 0.4, True)),
                          lodestroMethod=lms[2], coupledTo='pi')
    fields = [f0, f1, f2]
    multifield.check_fields_initialize(fields)
    call = multifield.ComputeAllHAllFields(fields, lodestro_method.TurbulenceHandler(dx, x, Flux3()))
    s = tsolver.Solver(1, x, np.array([0, dt]), maxit, tol, call, fields,
                       useInnerIteration=inner > 0, innerIterationMaxCount=max(inner, 1), **solver_kw)
    s.m += 1; s.converged = False; (s.t, s.dt) = s._get_tdt(s.tArray, s.m)
    s.errHistory[:] = 0
    s.profilesAllIterations = {l: np.zeros((maxit, N)) for l in ('n', 'pi', 'pe')}
    s.fluxesAllIterations = {l: np.zeros((maxit, N)) for l in ('n', 'pi', 'pe')}
    snaps = {}
    k = 0
    while s.iterationNumber < s.maxIterations and not s.converged and not s.solutionError:
        s.compute_next_iteration()
        if k in keep:
            snaps[k] = tuple(s.profiles[l].copy() for l in ('n', 'pi', 'pe'))
        k += 1
    fin = tuple(s.profiles[l].copy() for l in ('n', 'pi', 'pe'))
    ew = tuple(np.array([m._EWMATurbD._yEWMA_lminus1, m._EWMATurbc._yEWMA_lminus1, m._EWMAProfile._yEWMA_lminus1])
               for m in lms)
    return fin, k, s.errHistory[:k].copy(), snaps, ew, bool(s.converged)


def g9():
    out = {}
    fin, k, err, snaps, ew, conv = run_ref_3f(500, 2.2, (2, 0.1, 0.3), 0.3, 0.3, 1e-9, 1000, keep=(0, 1, 10))
    out.update(N500_iters=k, N500_err=err, N500_conv=conv, N500_n=fin[0], N500_pi=fin[1], N500_pe=fin[2],
               N500_ewma_n=ew[0], N500_ewma_pi=ew[1], N500_ewma_pe=ew[2])
    for it, p in snaps.items():
        out.update({f'N500_it{it}_n': p[0], f'N500_it{it}_pi': p[1], f'N500_it{it}_pe': p[2]})
    fin, k, err, snaps, ew, conv = run_ref_3f(500, 2.2, (2, 0.1, 0.3), 0.3, 0.3, 1e-9, 1000, inner=2, keep=(0,))
    out.update(inner_iters=k, inner_err=err, inner_conv=conv, inner_n=fin[0], inner_pi=fin[1], inner_pe=fin[2],
               inner_it0_n=snaps[0][0], inner_it0_pi=snaps[0][1], inner_it0_pe=snaps[0][2])
    fin, k, err, snaps, ew, conv = run_ref_3f(512, 3.1, (2.3, 0.09, 0.33), 0.3, 0.3, 1e-9, 1000, keep=(0,))
    out.update(N512_nu0=3.1, N512_bc=np.array((2.3, 0.09, 0.33)), N512_iters=k, N512_err=err, N512_conv=conv,
               N512_n=fin[0], N512_pi=fin[1], N512_pe=fin[2])
    save('g9_threefield', **out)


def g18():
    """Pseudo-timestep control on the three-field problem (tango/solver.py:198-216, :311-372 with three fields: a step is
    redone with a halved pseudo-timestep when ANY of the three profiles moved too far): the example's setup from a steeper
    start so that the control acts; every iterate of the first 30 is kept."""
    N = 500
    x = np.arange(N) * (1 / (N - 1))
    ic = 1.6 - 1.2 * x
    out = dict(N=N, nu0=2.2, bc=np.array((2.0, 0.1, 0.3)), ic=ic)
    cases = {'lightly': dict(useTreadLightly=True, treadLightlyParams={'maxFractionalChange': 0.2}),
             'lightly_tight': dict(useTreadLightly=True, treadLightlyParams={'maxFractionalChange': 0.05, 'innerFraction': 0.8}),
             'initially': dict(useTreadInitially=True, treadInitiallyParams={'numInitialSteps': 6, 'dtau': 0.05}),
             'both': dict(useTreadLightly=True, treadLightlyParams={'maxFractionalChange': 0.1},
                          useTreadInitially=True, treadInitiallyParams={'numInitialSteps': 4, 'dtau': 0.1})}
    for tag, kw in cases.items():
        fin, k, err, snaps, ew, conv = run_ref_3f(N, 2.2, (2, 0.1, 0.3), 0.3, 0.3, 1e-9, 400, keep=tuple(range(30)), ic=ic, **kw)
        out.update({f'{tag}_iters': k, f'{tag}_err': err, f'{tag}_conv': conv, f'{tag}_n': fin[0], f'{tag}_pi': fin[1], f'{tag}_pe': fin[2],
                    f'{tag}_first30': np.stack([np.stack(snaps[i]) for i in sorted(snaps)]), f'{tag}_ewma_pi': ew[1]})
        print(tag, 'iterations', k, 'converged', conv)
    save('g18_threefield_tread', **out)


# ---------------------------------------------------------------- G11 residual, uncoupled
def g11():
    rng = np.random.default_rng(11)
    n = 300
    A = rng.normal(size=n); B = rng.normal(size=n) + 4; C = rng.normal(size=n); f = rng.normal(size=n) * 3
    A[-1] = 0; C[0] = 0
    u = rng.normal(size=n)
    ufg = fieldgroups.UncoupledFieldGroup('n')
    r = ufg.calculate_residual(fieldgroups.UncoupledMatrixEqn(A, B, C, f), {'n': u})['n']
    save('g11_residual', A=A, B=B, C=C, f=f, u=u, r=r, rms=np.sqrt(1 / n * np.sum(r ** 2)))


if __name__ == '__main__':
    which = sys.argv[2:] or ['g1', 'g2', 'g3', 'g4', 'g567', 'g8', 'g9', 'g10', 'g11', 'g12', 'g13', 'g14', 'g15', 'g16', 'g17', 'g18', 'g19', 'g20']
    for w in which:
        globals()[w]()
