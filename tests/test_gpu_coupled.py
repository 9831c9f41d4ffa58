"""Coupled-field kernels: the pair-coupled (2x2 block) solve, the 3x3 block-tridiagonal solve and the fused
three-field Picard iteration, against the reference's golden vectors and the CPU oracle."""
import os

import numpy as np
import pytest

from conftest import relerr
from oracle import tango_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-10


@pytest.fixture(scope='module')
def tango():
    import tango_amd
    assert tango_amd.device_count() > 0
    return tango_amd


def test_pair_golden(tango, golden):
    """tests/test_JK_to_matrix_fd.py:10-76 and tests/test_fieldgroups.py:68-114 of the reference."""
    g = golden('g3_pair')
    z = np.zeros_like(g['x'])
    kw = {f'J{i}': g[f'J{i}'] for i in (1, 2, 3, 6, 7, 8)}
    kw.update({f'K{i}': g[f'K{i}'] for i in (1, 2, 3, 6, 7, 8)})
    Ue, We = tango.JK_to_matrix_fd.JK_to_matrix(float(g['dt']), float(g['dx']), (float(g['UL']), float(g['WL'])), (z, z), **kw)
    for got, k in zip(Ue, ('AU', 'BU', 'CU', 'DU', 'GW')):
        assert relerr(got, g[k]) < 1e-14, k
    for got, k in zip(We, ('AW', 'BW', 'CW', 'DW', 'GU')):
        assert relerr(got, g[k]) < 1e-14, k
    U, W = tango.JK_to_matrix_fd.solve(Ue, We)
    assert relerr(U, g['U']) < TOL and relerr(W, g['W']) < TOL
    assert np.allclose(U, g['Uan'], rtol=1e-4, atol=1e-4) and np.allclose(W, g['Wan'], rtol=1e-4, atol=1e-4)
    with pytest.raises(AssertionError):
        tango.JK_to_matrix_fd.JK_to_matrix(1.0, 0.1, (1, 1), (z, z), J2=g['J2'], K1=g['K1'])
    # field-group chain and residual (golden: perturbed profiles with the same matrix)
    pfg = tango.fieldgroups.PairCoupledFieldGroup('a', 'b')
    JK = tango.fieldgroups.JKCoefficients(**kw)
    meq = pfg.Hcoeffs_to_matrix_eqn(float(g['dt']), float(g['dx']), (float(g['UL']), float(g['WL'])), (z, z), JK)
    sol = pfg.solve_matrix_eqn(meq)
    assert relerr(sol['a'], g['U']) < TOL and relerr(sol['b'], g['W']) < TOL
    res = pfg.calculate_residual(meq, {'a': g['res_U_in'], 'b': g['res_W_in']})
    # the residual cancels terms of size max|B| max|u| down to max|res|: its rounding bound, not 1e-10, is the bar
    for k, u, B, ref in (('a', 'res_U_in', 'BU', 'res_U'), ('b', 'res_W_in', 'BW', 'res_W')):
        bound = 4 * np.finfo(float).eps * np.max(np.abs(g[B])) * np.max(np.abs(g[u])) / np.max(np.abs(g[ref]))
        assert relerr(res[k], g[ref]) < bound, (k, bound)


@pytest.mark.parametrize('n', [1, 2, 5, 16, 100, 255, 256, 500, 1000, 2048, 4096])
def test_pair_solve_random(tango, n):
    rng = np.random.default_rng(100 + n)
    nb = 3
    def dom():
        A = rng.uniform(-1, 1, (nb, n)); C = rng.uniform(-1, 1, (nb, n)); G = rng.uniform(-0.5, 0.5, (nb, n))
        B = np.abs(A) + np.abs(C) + np.abs(G) + rng.uniform(0.1, 1, (nb, n))
        return A, B, C, rng.normal(size=(nb, n)), G
    U, W = dom(), dom()
    ru, rw = orc.pair_solve(U, W)
    gu, gw = tango.JK_to_matrix_fd.solve(tango.JK_to_matrix_fd.MatEqnCoeffs(*U), tango.JK_to_matrix_fd.MatEqnCoeffs(*W))
    assert relerr(gu, ru) < 1e-12 and relerr(gw, rw) < 1e-12


@pytest.mark.parametrize('n', [1, 3, 4, 17, 64, 200, 256, 700, 2048])
def test_block3_solve(tango, n):
    """3x3 block-tridiagonal solve against a dense solve; also the reference's structure (n uncoupled, pi<->pe
    coupled only through the diagonal block) as a special case."""
    from tango_amd import _ops
    rng = np.random.default_rng(7 * n)
    nb = 2 if n > 500 else 4
    Cb = rng.uniform(-1, 1, (nb, n, 3, 3)); Ab = rng.uniform(-1, 1, (nb, n, 3, 3))
    Bb = rng.uniform(-1, 1, (nb, n, 3, 3)) + 8 * np.eye(3)
    d = rng.normal(size=(nb, n, 3))
    x = _ops.block3_solve(Cb, Bb, Ab, d)
    for b in range(nb):
        ref = orc.block3_dense_solve(Cb[b], Bb[b], Ab[b], d[b])
        assert relerr(x[b], ref) < 1e-11
    # reference structure: scalar field 0, pair (1,2) with diagonal off-diagonal blocks
    a = rng.uniform(-1, 1, (3, n)); c = rng.uniform(-1, 1, (3, n)); gg = rng.uniform(-0.5, 0.5, (2, n))
    bd = np.abs(a) + np.abs(c) + 1.0
    rhs = rng.normal(size=(3, n))
    Cb1 = np.zeros((n, 3, 3)); Ab1 = np.zeros((n, 3, 3)); Bb1 = np.zeros((n, 3, 3))
    for k in range(3):
        Cb1[:, k, k] = c[k]; Ab1[:, k, k] = a[k]; Bb1[:, k, k] = bd[k]
    Bb1[:, 1, 2] = gg[0]; Bb1[:, 2, 1] = gg[1]
    x1 = _ops.block3_solve(Cb1, Bb1, Ab1, rhs.T.copy())
    u0 = orc.tridiag_solve(a[0], bd[0], c[0], rhs[0])
    u1, u2 = orc.pair_solve((a[1], bd[1], c[1], rhs[1], gg[0]), (a[2], bd[2], c[2], rhs[2], gg[1]))
    assert relerr(x1[:, 0], u0) < 1e-12 and relerr(x1[:, 1], u1) < 1e-12 and relerr(x1[:, 2], u2) < 1e-12


def make_3f_solver(tango, N, nu0, BCs, tol=1e-9, maxit=1000, inner=0, fused=True, ic=None, ewma=(0.3, 0.3), dt=1e6, ewma_iterates=None,
                   external=False, order=(0, 1, 2), **kw):
    T = tango.extras.threefield
    dx = 1 / (N - 1); x = np.arange(N) * dx
    th = {'Dmin': 1e-5, 'Dmax': 1e13, 'dpdxThreshold': 10}
    ic = (1 - 0.5 * x) if ic is None else ic
    cbs = [T.ComputeAllH_n(), T.ComputeAllH_pi(nu0), T.ComputeAllH_pe(nu0)]
    model = T.ShestakovThreeFieldFluxModel(dx)
    if not fused or external:     # plain callables, as a user's script has them (examples/shestakov_threefield_nonlinearnu.py:52-149)
        cbs = [(lambda cb: (lambda t, xx, p, HT: cb(t, xx, p, HT)))(cb) for cb in cbs]
    if external:                  # ... and a flux model the package knows nothing about

        class UserFluxModel(object):
            def __init__(self, inner):
                self.calls, self.inner = 0, inner

            def get_flux(self, profiles):
                self.calls += 1
                return self.inner.get_flux(profiles)
        model = UserFluxModel(model)
    if not fused:
        kw.setdefault('fusedPaths', False)      # the generic path: one kernel per reference function
    lms = [tango.lodestro_method.lm(ewma[0], ewma[1], th) for _ in range(3)]
    if ewma_iterates is not None:      # continuing an interrupted run: (profile, D, c) memories of the three fields
        for m_, its_ in zip(lms, ewma_iterates):
            m_.set_ewma_iterates(*its_)
    F = tango.multifield.Field
    fields = [F(label='n', rightBC=BCs[0], profile_mminus1=1.0 * ic, compute_all_H=cbs[0], lodestroMethod=lms[0]),
              F(label='pi', rightBC=BCs[1], profile_mminus1=1.0 * ic, compute_all_H=cbs[1], lodestroMethod=lms[1], coupledTo='pe'),
              F(label='pe', rightBC=BCs[2], profile_mminus1=1.0 * ic, compute_all_H=cbs[2], lodestroMethod=lms[2], coupledTo='pi')]
    fields = [fields[k] for k in order]
    tango.multifield.check_fields_initialize(fields)
    call = tango.multifield.ComputeAllHAllFields(fields, tango.lodestro_method.TurbulenceHandler(dx, x, model))
    return tango.solver.Solver(1, x, np.array([0, dt]), maxit, tol, call, fields, useInnerIteration=inner > 0,
                               innerIterationMaxCount=max(inner, 1), **kw), lms


def test_threefield_fused_matches_reference(tango, golden):
    g = golden('g9_threefield')
    s, lms = make_3f_solver(tango, 500, 2.2, (2, 0.1, 0.3))
    assert s._fused is not None
    while s.ok:
        s.take_timestep()
    assert s.reachedEnd and s.converged
    assert len(s.errHistoryFinal) == int(g['N500_iters'])       # stops on the reference's iteration
    k = min(len(s.errHistoryFinal), int(g['N500_iters']))
    assert np.allclose(s.errHistoryFinal[:k], g['N500_err'][:k], rtol=1e-6, atol=5e-12)   # rounding floor of the rms residual
    for lab in ('n', 'pi', 'pe'):
        assert relerr(s.profiles[lab], g[f'N500_{lab}']) < TOL
        assert relerr(s.profilesAllIterations[lab][0], g[f'N500_it0_{lab}']) < TOL
        assert relerr(s.profilesAllIterations[lab][10], g[f'N500_it10_{lab}']) < TOL
    # values quoted for this problem
    assert abs(s.profiles['n'][0] - 3.943314) < 1e-6 and abs(s.profiles['pi'][0] - 0.325379) < 1e-6
    if len(s.errHistoryFinal) == int(g['N500_iters']):
        assert relerr(lms[1]._EWMATurbD._yEWMA_lminus1, g['N500_ewma_pi'][0]) < TOL


@pytest.mark.parametrize('order', [(0, 1, 2), (2, 0, 1)])
def test_threefield_outside_inputs_match_reference(tango, golden, order):
    """The three-field example with everything problem-specific in Python, as the reference's script has it: a flux model
    object with get_flux and plain compute_all_H functions that form the collision frequency from the current profiles.
    The run takes the fused step with outside inputs (flux arrays, H6/H7/H8 arrays; one launch per iteration) and must
    follow the reference's trajectory: the same 65 iterations, residual history, iterates and final profiles.  The fields
    may be listed in any order (the pair is found by coupledTo, tango/fieldgroups.py:275-331)."""
    from tango_amd.solver import _FusedExternal3F
    g = golden('g9_threefield')
    s, lms = make_3f_solver(tango, 500, 2.2, (2, 0.1, 0.3), external=True, order=order)
    assert isinstance(s._fused, _FusedExternal3F) and not s._fused.static
    while s.ok:
        s.take_timestep()
    assert s.reachedEnd and s.converged and not s.solutionError
    assert len(s.errHistoryFinal) == int(g['N500_iters'])       # stops on the reference's iteration
    assert s.compute_all_H_all_fields.turbhandler.fluxModel.calls == int(g['N500_iters'])
    assert np.allclose(s.errHistoryFinal, g['N500_err'], rtol=1e-6, atol=5e-12)
    for lab in ('n', 'pi', 'pe'):
        assert relerr(s.profiles[lab], g[f'N500_{lab}']) < TOL and s.profiles[lab].shape == (500,)
        assert relerr(s.profilesAllIterations[lab][0], g[f'N500_it0_{lab}']) < TOL
        assert relerr(s.profilesAllIterations[lab][10], g[f'N500_it10_{lab}']) < TOL
    assert relerr(lms[1]._EWMATurbD._yEWMA_lminus1, g['N500_ewma_pi'][0]) < TOL


def test_threefield_outside_inputs_inner_iteration_and_ensemble(tango, golden):
    """Outside inputs with inner iterations (the callbacks are called again with every new profile), and an ensemble whose
    members have their own collision amplitude and boundary values: the same numbers as the built-in closure gives."""
    from tango_amd.solver import _FusedExternal3F
    g = golden('g9_threefield')
    s, _ = make_3f_solver(tango, 500, 2.2, (2, 0.1, 0.3), inner=2, external=True)
    assert isinstance(s._fused, _FusedExternal3F)
    while s.ok:
        s.take_timestep()
    assert s.converged and len(s.errHistoryFinal) == int(g['inner_iters'])
    for lab in ('n', 'pi', 'pe'):
        assert relerr(s.profiles[lab], g[f'inner_{lab}']) < TOL
        assert relerr(s.profilesAllIterations[lab][0], g[f'inner_it0_{lab}']) < TOL
    N, nb = 256, 5
    ens = orc.threefield_ensemble(nb, N)
    bcs = tuple(ens['bc'][:, k].copy() for k in range(3))
    runs = []
    for external in (False, True):
        s, _ = make_3f_solver(tango, N, ens['nu0'], bcs, ic=ens['u0'], external=external, iterationsPerCheck=4)
        assert s.batched and isinstance(s._fused, _FusedExternal3F) == external
        s.take_timestep()
        assert s.converged
        runs.append(s)
    a, b = runs
    assert np.array_equal(a.iterationsPerProfile, b.iterationsPerProfile) and len(set(a.iterationsPerProfile.tolist())) > 1
    for lab in ('n', 'pi', 'pe'):
        assert relerr(b.profiles[lab], a.profiles[lab]) < TOL
    assert np.allclose(b.errHistory, a.errHistory, rtol=1e-6, atol=5e-12)


def test_threefield_inner_iteration(tango, golden):
    g = golden('g9_threefield')
    s, _ = make_3f_solver(tango, 500, 2.2, (2, 0.1, 0.3), inner=2)
    assert s._fused is not None
    while s.ok:
        s.take_timestep()
    assert s.converged and len(s.errHistoryFinal) == int(g['inner_iters'])
    for lab in ('n', 'pi', 'pe'):
        assert relerr(s.profiles[lab], g[f'inner_{lab}']) < TOL
        assert relerr(s.profilesAllIterations[lab][0], g[f'inner_it0_{lab}']) < TOL


@pytest.mark.parametrize('nb', [1, 3])
def test_threefield_interrupt_and_continue(tango, nb):
    """A three-field run stopped after 9 iterations and continued in a new Solver from the profiles and the EWMA memories
    (tango/restart.py:90-98, tango/solver.py:109) follows the uninterrupted run; the iteration number carries on."""
    N = 256
    if nb == 1:
        nu0, BCs, ic = 2.2, (2, 0.1, 0.3), None
    else:
        ens = orc.threefield_ensemble(nb, N)
        nu0, BCs, ic = ens['nu0'], tuple(ens['bc'][:, k].copy() for k in range(3)), ens['u0']
    whole, _ = make_3f_solver(tango, N, nu0, BCs, tol=1e-30, maxit=20, ic=ic)
    whole.take_timestep()
    part, lms = make_3f_solver(tango, N, nu0, BCs, tol=1e-30, maxit=9, ic=ic)
    part.take_timestep()
    assert part.countLastTimestep == 9
    memories = [tuple(np.array(v) for v in (m._EWMAProfile._yEWMA_lminus1, m._EWMATurbD._yEWMA_lminus1, m._EWMATurbc._yEWMA_lminus1))
                for m in lms]
    cont, _ = make_3f_solver(tango, N, nu0, BCs, tol=1e-30, maxit=20, ic=ic, ewma_iterates=memories,
                             profiles={k: np.array(v) for k, v in part.profiles.items()}, startIterationNumber=9)
    assert cont._fused is not None
    cont.take_timestep()
    assert cont.countLastTimestep == 20 and np.shape(cont.errHistoryFinal)[-1] == 11
    for lab in ('n', 'pi', 'pe'):
        assert relerr(cont.profiles[lab], whole.profiles[lab]) < 1e-12
    assert np.allclose(cont.errHistoryFinal, whole.errHistoryFinal[..., 9:20], rtol=1e-9, atol=1e-15)


TREAD3_CASES = {'lightly': dict(useTreadLightly=True, treadLightlyParams={'maxFractionalChange': 0.2}),
                'lightly_tight': dict(useTreadLightly=True, treadLightlyParams={'maxFractionalChange': 0.05, 'innerFraction': 0.8}),
                'initially': dict(useTreadInitially=True, treadInitiallyParams={'numInitialSteps': 6, 'dtau': 0.05}),
                'both': dict(useTreadLightly=True, treadLightlyParams={'maxFractionalChange': 0.1},
                             useTreadInitially=True, treadInitiallyParams={'numInitialSteps': 4, 'dtau': 0.1})}


@pytest.mark.parametrize('tag', sorted(TREAD3_CASES))
@pytest.mark.parametrize('fused', [True, False])
def test_threefield_pseudo_timestep_control(tango, golden, tag, fused):
    """useTreadLightly / useTreadInitially on the three-field problem (tango/solver.py:198-216, :311-372): inside the fused
    kernel and on the generic path, against the reference's own runs (g18: every one of the first 30 iterates)."""
    g = golden('g18_threefield_tread')
    N = int(g['N'])
    maxit = 400 if fused else 30
    s, lms = make_3f_solver(tango, N, float(g['nu0']), tuple(g['bc']), ic=g['ic'], fused=fused, maxit=maxit, **TREAD3_CASES[tag])
    assert (s._fused is not None) == fused
    s.take_timestep()
    k = int(g[f'{tag}_iters'])
    m = min(k, len(s.errHistoryFinal))
    assert np.allclose(s.errHistoryFinal[:m], g[f'{tag}_err'][:m], rtol=1e-6, atol=5e-12)
    for q, lab in enumerate(('n', 'pi', 'pe')):
        assert relerr(s.profilesAllIterations[lab][:30], g[f'{tag}_first30'][:, q]) < TOL, lab
    if fused:
        assert s.converged and len(s.errHistoryFinal) == k
        for lab in ('n', 'pi', 'pe'):
            assert relerr(s.profiles[lab], g[f'{tag}_{lab}']) < TOL
        assert relerr(lms[1]._EWMATurbD._yEWMA_lminus1, g[f'{tag}_ewma_pi'][0]) < TOL


def test_threefield_pseudo_timestep_control_ensemble(tango):
    """Per-profile halving inside one launch: an ensemble whose members need different numbers of halvings (several of
    them share a wavefront at N = 64), each against the oracle run alone."""
    for N, nb in ((64, 9), (512, 4)):
        ens = orc.threefield_ensemble(nb, N)
        nu0, bc = ens['nu0'], ens['bc']
        ic = (1.2 + 0.5 * np.arange(nb)[:, None] / nb) - (0.8 + 0.4 * np.arange(nb)[:, None] / nb) * ens['x'][None, :]
        tl = {'maxFractionalChange': 0.1}
        s, _ = make_3f_solver(tango, N, nu0, tuple(bc[:, k].copy() for k in range(3)), ic=ic, maxit=25, tol=1e-30, useTreadLightly=True,
                              treadLightlyParams=tl, iterationsPerCheck=5)
        assert s._fused is not None and s.batched
        s.take_timestep()
        halved = 0
        for b in range(nb):
            (n, pi, pe), it, err, hist, halv = orc.run_3f_tread(ic[b], ic[b], ic[b], 1e6, ens['dx'], tuple(bc[b]), nu0[b], ens['x'], 0.3, 0.3,
                                                                1e-30, 25, tread_lightly=tl)
            halved += int(halv.max() > 0)
            for lab, ref in (('n', n), ('pi', pi), ('pe', pe)):
                assert relerr(s.profiles[lab][b], ref) < TOL, (N, b, lab)
            assert np.allclose(s.errHistory[b, :25], err, rtol=1e-6, atol=5e-12)
        assert halved >= 2


def test_threefield_generic_path(tango, golden):
    """User-style callbacks: n through the scalar kernels, (pi, pe) through JK_to_matrix + the pair solve."""
    g = golden('g9_threefield')
    s, _ = make_3f_solver(tango, 500, 2.2, (2, 0.1, 0.3), fused=False, maxit=12, tol=1e-30)
    assert s._fused is None
    s.take_timestep()
    for lab in ('n', 'pi', 'pe'):
        assert relerr(s.profilesAllIterations[lab][0], g[f'N500_it0_{lab}']) < TOL
        assert relerr(s.profilesAllIterations[lab][10], g[f'N500_it10_{lab}']) < TOL
    assert np.allclose(s.errHistoryFinal, g['N500_err'][:12], rtol=1e-6, atol=1e-12)


def test_threefield_ensemble(tango, golden):
    """An ensemble with per-profile nu0 and boundary values: member 0 is the example problem, member 1 the
    second golden trajectory (N = 512), the others are checked against the CPU oracle."""
    g = golden('g9_threefield')
    N, nb = 512, 6
    ens = orc.threefield_ensemble(nb, N)
    nu0 = ens['nu0'].copy(); bc = ens['bc'].copy()
    nu0[1] = float(g['N512_nu0']); bc[1] = g['N512_bc']
    s, _ = make_3f_solver(tango, N, nu0, (bc[:, 0].copy(), bc[:, 1].copy(), bc[:, 2].copy()), ic=ens['u0'], iterationsPerCheck=5)
    assert s._fused is not None and s.batched
    s.take_timestep()
    assert s.converged
    assert int(s.iterationsPerProfile[1]) == int(g['N512_iters'])
    for lab in ('n', 'pi', 'pe'):
        assert relerr(s.profiles[lab][1], g[f'N512_{lab}']) < TOL
    for b in (0, 3, 5):
        (n, pi, pe), it, err = orc.run_3f(ens['u0'][b], ens['u0'][b], ens['u0'][b], 1e6, ens['dx'], tuple(bc[b]), nu0[b], ens['x'],
                                          0.3, 0.3, 1e-9, 1000)
        assert int(s.iterationsPerProfile[b]) == it
        assert relerr(s.profiles['n'][b], n) < TOL and relerr(s.profiles['pi'][b], pi) < TOL and relerr(s.profiles['pe'][b], pe) < TOL


@pytest.mark.parametrize('seed', range(int(os.environ.get('TANGO_RANDOM_SEEDS_3F', '10'))))      # more seeds for a sweep by hand
def test_threefield_random_problems_vs_oracle(tango, seed):
    """Randomly drawn grids, collision frequencies, boundary values, EWMA parameters and timesteps, a fixed number of
    iterations: the fused three-field kernel (one wavefront or one workgroup per profile, full and ragged grids)
    against the CPU oracle, member by member."""
    rng = np.random.default_rng(500 + seed)
    N = int(rng.choice([64, 100, 255, 256, 500, 512, 1024]))
    nb = int(rng.integers(1, 5))
    its = int(rng.integers(3, 10))
    a_turb, a_prof = float(rng.uniform(0.1, 0.5)), float(rng.choice([1.0, rng.uniform(0.1, 1.0)]))
    dt = float(10 ** rng.uniform(2, 6))
    ens = orc.threefield_ensemble(nb + 1, N, seed=seed)
    nu0, bc, ic = ens['nu0'][1:].copy(), ens['bc'][1:].copy(), ens['u0'][1:].copy()
    s, _ = make_3f_solver(tango, N, nu0, (bc[:, 0].copy(), bc[:, 1].copy(), bc[:, 2].copy()), ic=ic, tol=1e-30, maxit=its,
                          ewma=(a_turb, a_prof), dt=dt, iterationsPerCheck=int(rng.integers(1, 4)))
    assert s._fused is not None and s.batched
    s.take_timestep()
    assert not s.solutionError and np.all(s.iterationsPerProfile == its)
    bar = TOL if N <= 512 else 3e-10
    for b in range(nb):
        (n, pi, pe), it, err = orc.run_3f(ic[b], ic[b], ic[b], dt, ens['dx'], tuple(bc[b]), nu0[b], ens['x'], a_turb, a_prof, 1e-30, its,
                                          fixed_count=True)
        for lab, ref in (('n', n), ('pi', pi), ('pe', pe)):
            assert relerr(s.profiles[lab][b], ref) < bar, (seed, b, lab, N, dt)
        assert np.allclose(s.errHistory[b, :its], err[:its], rtol=1e-6, atol=1e-11), (seed, b)


def test_threefield_full_size_properties(tango):
    """16 384 x 512 (the three-field benchmark size): valid results, a scattered subset stepped alone reproduces its
    members of the full ensemble bit for bit (batch independence), and members against the CPU oracle."""
    from tango_amd.ensemble import ThreeFieldEnsemble, threefield_ensemble
    nb, N, its = 16384, 512, 4
    ens = threefield_ensemble(nb, N)
    E = ThreeFieldEnsemble(ens, max_iter=its, fixed_count=True)
    E.step(iters=its)
    u = [a.to_host() for a in E.u]
    assert all(np.all(np.isfinite(a)) and np.all(a >= 0) for a in u) and not np.any(E.status.to_host() & 6)
    assert np.all(E.iters.to_host() == its)
    sel = np.array([0, 1, 127, 8192, 16383])
    sub = {k: (v[sel] if k in ('u0', 'nu0', 'bc') else v) for k, v in ens.items()}
    E2 = ThreeFieldEnsemble(sub, max_iter=its, fixed_count=True)
    E2.step(iters=its)
    for k in range(3):
        assert np.array_equal(E2.u[k].to_host(), u[k][sel])
        assert np.array_equal(E2.DE[k].to_host(), E.DE[k].to_host()[sel])
    assert np.array_equal(E2.err.to_host(), E.err.to_host()[sel])
    for i in (0, 2, 4):
        b = sel[i]
        ic = ens['u0'][b]
        (n, pi, pe), it, err = orc.run_3f(ic, ic, ic, 1e6, ens['dx'], tuple(ens['bc'][b]), ens['nu0'][b], ens['x'], 0.3, 0.3, 1e-30, its,
                                          fixed_count=True)
        for k, ref in enumerate((n, pi, pe)):
            assert relerr(u[k][b], ref) < TOL, (b, k)
        assert np.allclose(E.err.to_host()[b, :its], err[:its], rtol=1e-6, atol=1e-11)


@pytest.mark.parametrize('N,nb', [(512, 3), (100, 4), (1024, 2), (64, 5)])
def test_threefield_step_with_outside_inputs_vs_oracle(tango, N, nb):
    """tg_picard_3f_step with TG_FLUX_EXTERNAL through the C ABI: arbitrary flux arrays, an H6 on every field, H8 that differ
    between the two fields of the pair (so that the orientation of the coupling is pinned: H8 of the first multiplies the
    second one's profile in the first one's equation, tango/JK_to_matrix_fd.py:97-112), H7 once per profile and once
    shared, against the same iteration put together from the CPU oracle's pieces; two iterations, the second with
    seeded memories, then an inner pass with the turbulent coefficients frozen."""
    import ctypes as C
    from tango_amd import _hip as H
    from tango_amd.ensemble import ThreeFieldEnsemble
    rng = np.random.default_rng(5 + N)
    ens = orc.threefield_ensemble(nb, N)
    x, dx, dt, a_turb, a_prof = ens['x'], ens['dx'], 1e6, 0.3, 0.4
    theta = dict(Dmin=1e-5, Dmax=1e13, dpdxThreshold=10.0)
    E = ThreeFieldEnsemble(ens, dt=dt, a_turb=a_turb, a_prof=a_prof, max_iter=4, fixed_count=True)
    q = E.params
    q.flux_model = H.FLUX_EXTERNAL
    D = H.DeviceArray
    u0 = [ens['u0'] * (1.0 + 0.1 * k) for k in range(3)]
    for k in range(3):
        E.u[k].set(u0[k]); E.um1[k].set(u0[k])
    H7 = [rng.uniform(0, 2, (nb, N)), rng.uniform(0, 2, N), None]
    H6 = [-rng.uniform(0, 0.5, N), -rng.uniform(1, 3, (nb, N)), -rng.uniform(1, 3, (nb, N))]
    H8 = [None, 0.7 * -H6[1], rng.uniform(0.2, 0.9, N)[None, :] * -H6[2]]
    keep = []
    for k in range(3):
        for name, arrs, ptrs, strides in (('H6', H6, q.H6, q.h6_stride), ('H7', H7, q.H7, q.h7_stride), ('H8', H8, q.H8, q.h8_stride)):
            ptrs[k], strides[k] = None, 0
            if arrs[k] is not None:
                d = D.from_host(arrs[k]); keep.append(d)
                ptrs[k], strides[k] = d.ptr, (N if arrs[k].ndim == 2 else 0)
    flux_dev = [D((nb, N)) for _ in range(3)]
    P3 = H._p * 3
    q.flux = P3(*[d.ptr for d in flux_dev])
    full = lambda a, b: np.zeros(N) if a is None else (a[b] if a.ndim == 2 else a)
    st = [orc.State3F(u0[0][b], u0[1][b], u0[2][b]) for b in range(nb)]
    status = np.zeros(nb, np.int32)
    for it in range(3):
        frozen_pass = it == 2
        if not frozen_pass:
            # the caller's part: EWMA of the profiles, then the flux "model" on it (here: the example's flux, perturbed)
            for k in range(3):
                H.check(H.lib.tg_ewma_profiles(E.u[k].ptr, E.uE[k].ptr, a_prof, E.status.ptr, nb, N, H.stream()))
            uE = [E.uE[k].to_host() for k in range(3)]
            fl = [f * rng.uniform(0.8, 1.2, (nb, N)) for f in orc.threefield_flux(uE[0], uE[1], uE[2], dx)]
            for k in range(3):
                flux_dev[k].set(fl[k])
        q.frozen_turb, q.outer_count = int(frozen_pass), (2 if frozen_pass else 0)
        H.check(H.lib.tg_picard_3f_step(C.byref(E.state), C.byref(q), H.stream()))
        err = E.err.to_host()
        for b in range(nb):
            s3 = st[b]
            if not frozen_pass:
                for k, s in enumerate(s3.f):
                    s.uE = orc.ewma_next(s.u, s.uE, a_prof)
                    assert relerr(uE[k][b], s.uE) < TOL
                    Dk, ck, *_ = orc.flux_split(fl[k][b], s.uE, dx, theta['Dmin'], theta['Dmax'], theta['dpdxThreshold'])
                    s.DE, s.cE = orc.ewma_next(Dk, s.DE, a_turb), orc.ewma_next(ck, s.cE, a_turb)
            h = [orc.dc_to_h(s.DE, s.cE) for s in s3.f]
            one = np.ones(N)
            fn, fi, fe = s3.f
            bc = ens['bc'][b]
            An, Bn, Cn, Dn = orc.h_to_matrix(dt, dx, bc[0], fn.um1, one, H2=h[0][0], H3=h[0][1], H6=full(H6[0], b), H7=full(H7[0], b))
            Ueq, Weq = orc.jk_to_matrix(dt, dx, (bc[1], bc[2]), (fi.um1, fe.um1),
                                        dict(H1=one, H2=h[1][0], H3=h[1][1], H6=full(H6[1], b), H7=full(H7[1], b), H8=full(H8[1], b)),
                                        dict(H1=one, H2=h[2][0], H3=h[2][1], H6=full(H6[2], b), H7=full(H7[2], b), H8=full(H8[2], b)))
            rms = orc.rms_of([orc.residual(An, Bn, Cn, Dn, fn.u), orc.residual(*Ueq[:4], fi.u, Ueq[4], fe.u),
                              orc.residual(*Weq[:4], fe.u, Weq[4], fi.u)])
            if not frozen_pass:
                assert abs(err[b, it] - rms) <= 1e-6 * rms + 5e-12
            fn.u = orc.tridiag_solve(An, Bn, Cn, Dn)
            fi.u, fe.u = orc.pair_solve(Ueq, Weq)
        for k in range(3):
            got_u, got_D, got_c = E.u[k].to_host(), E.DE[k].to_host(), E.cE[k].to_host()
            for b in range(nb):
                s = st[b].f[k]
                bar = TOL if N <= 512 else 3e-10
                assert relerr(got_u[b], s.u) < bar, (it, k, b)
                assert relerr(got_D[b], s.DE) < bar and relerr(got_c[b], s.cE) < bar, (it, k, b)
    assert np.all(E.iters.to_host() == 2) and not np.any(E.status.to_host() & 6)
    # the argument checks of the outside-input form
    q.frozen_turb = 0
    keepH8 = q.H8[0]
    q.H8[0] = keep[0].ptr
    with pytest.raises(ValueError, match='no H8'):
        H.check(H.lib.tg_picard_3f_step(C.byref(E.state), C.byref(q), H.stream()))
    q.H8[0] = keepH8
    q.pseudo_time = 1
    with pytest.raises(NotImplementedError):
        H.check(H.lib.tg_picard_3f_step(C.byref(E.state), C.byref(q), H.stream()))


def test_threefield_outside_inputs_on_the_device():
    """The same with the flux model and the collision terms on the GPU (tests/device_coupled_case.py).  In a process of its
    own: torch brings its own HIP runtime, which has to be loaded before libtango_hip.so for the two to share it."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, 'device_coupled_case.py')], capture_output=True, text=True, timeout=160)
    assert r.returncode == 0 and 'device coupled case ok' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
