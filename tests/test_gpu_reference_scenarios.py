"""The scenarios of the reference's own test files, posed to tango_amd through the reference's API and held to the
reference's tolerances: problems with analytic answers (manufactured solutions, steady states) and the known-answer
vectors written out in those files.  Nothing here comes from a golden file: a user's copy of the reference's tests, with
`import tango_amd as tango`, asks the same questions.  Each test names the reference test it restates."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def tango():
    import tango_amd
    assert tango_amd.device_count() > 0, 'GPU tests need a GPU'
    return tango_amd


# ---- analytic problems ---------------------------------------------------------------------------------------------
def cell_centred_grid(N, L=1.0):
    """Points dr/2, ..., L (the polar grids of the reference's tests)."""
    dr = L / (N - 0.5)
    return dr, np.linspace(dr / 2, L, N)


def slab_problem():
    """u'' D = -(1 - x^2), u(1) = uL, u'(0) = 0 (tests/test_HToMatrixFD.py:11-39, tests/test_fieldgroups.py:243-259)."""
    N, D, uL = 500, 0.8, 0.2
    dx = 1.0 / (N - 1)
    x = np.arange(N) * dx
    return dict(dt=1e5, dx=dx, UL=uL, U0=np.sin(np.pi * x) + uL * x, H1=np.ones(N), H2=D * np.ones(N), H7=1 - x**2,
                exact=uL + (1 - x**2) * (5 - x**2) / (12 * D))


def disc_problem():
    """(1/r)(r D u')' = -r, u(1) = uL (tests/test_HToMatrixFD.py:41-68, tests/test_fieldgroups.py:261-279)."""
    N, D, uL = 500, 1.3, 0.03
    dr, r = cell_centred_grid(N)
    return dict(dt=1e6, dx=dr, UL=uL, U0=0.3 * np.sin(np.pi * r) + uL * r, H1=r, H2=D * r, H7=r**2,
                exact=uL + (1 - r**3) / (9 * D))


def two_field_problem():
    """Two fields coupled through H6 and H8 with diffusion and convection, manufactured so that U = UL + a (1 - x^3) and
    W = WL + b cos(pi x / 2) (tests/test_JK_to_matrix_fd.py:10-76, tests/test_fieldgroups.py:281-343)."""
    N = 500
    dx, x = cell_centred_grid(N)
    DU, DW, cU, cW, nu, a, b, UL, WL = 1.6, 2.4, 0.3, 9.0, 1.3, 1.9, 0.09, 0.75, 3.33
    s, c, h = np.sin(np.pi * x / 2), np.cos(np.pi * x / 2), np.pi / 2
    J = dict(J1=np.ones(N), J2=DU * x, J3=-cU * x**2, J6=x, J8=nu * x,
             J7=9 * a * DU * x**2 + 2 * cU * x * UL + cU * a * (2 * x - 5 * x**4) - x * UL - a * (x - x**4) - nu * x * WL - nu * b * x * c)
    K = dict(K1=np.ones(N), K2=DW * x, K3=-cW * x**2, K6=x**2, K8=-nu * x,
             K7=h * DW * b * (s + h * x * c) + 2 * cW * x * WL + cW * b * (2 * x * c - h * x**2 * s) - x**2 * WL - b * x**2 * c
             + nu * x * UL + nu * a * (x - x**4))
    return dict(dt=1e9, dx=dx, rightBC=(UL, WL), psi0=(np.zeros(N), np.zeros(N)), J=J, K=K,
                exact=(UL + a * (1 - x**3), WL + b * c))


def mean_error(u, exact):
    return np.linalg.norm(u - exact) / len(u)


# ---- tests/test_HToMatrixFD.py -------------------------------------------------------------------------------------
def test_steady_diffusion_in_a_slab_and_on_a_disc(tango):
    for problem, tol in ((slab_problem(), 1e-4), (disc_problem(), 1e-8)):          # :11-39, :41-68
        p = problem
        u = tango.HToMatrixFD.H_to_matrix_and_solve(p['dt'], p['dx'], p['UL'], p['U0'], H1=p['H1'], H2=p['H2'], H7=p['H7'])
        assert mean_error(u, p['exact']) < tol


def test_time_dependent_convection_diffusion_on_a_disc(tango):
    """:70-113: ten implicit steps of a manufactured solution with H1, H2, H3 and a time-dependent H7."""
    N, D, c0, g, uL, dt = 300, 1.3, 0.7, 0.9, 0.5, 1e-2
    dr, r = cell_centred_grid(N)
    u = uL * np.ones(N)
    for m in range(1, 11):
        t = m * dt
        grow = 1 - np.exp(-g * t)
        H7 = r / D * g * np.exp(-g * t) * (1 - r**5) + 25 * r**4 * grow + c0 * (2 * r * uL + grow / D * (2 * r - 7 * r**6))
        u = tango.HToMatrixFD.H_to_matrix_and_solve(dt, dr, uL, u, H1=r, H2=D * r, H3=-c0 * r**2, H7=H7)
    assert mean_error(u, uL + grow / D * (1 - r**5)) < 2e-5


def test_convection_written_as_h4(tango):
    """:115-134."""
    N, uL = 500, 2.2
    dx = 1.0 / (N - 1)
    x = np.arange(N) * dx
    u = tango.HToMatrixFD.H_to_matrix_and_solve(1e9, dx, uL, np.sin(np.pi * x) + uL * x, H1=np.ones(N), H2=np.ones(N),
                                                H4=-np.sin(x), H7=1 - x**2)
    exact = uL - (np.cos(x) - np.cos(1.0)) - (x**2 - 1) / 2 + (x**4 - 1) / 12
    assert np.allclose(u, exact, rtol=1e-3, atol=1e-3)


# ---- tests/test_JK_to_matrix_fd.py, tests/test_fieldgroups.py --------------------------------------------------------
def test_two_coupled_fields_manufactured(tango):
    p = two_field_problem()                                                        # tests/test_JK_to_matrix_fd.py:10-76
    Ueq, Weq = tango.JK_to_matrix_fd.JK_to_matrix(p['dt'], p['dx'], p['rightBC'], p['psi0'], **p['J'], **p['K'])
    U, W = tango.JK_to_matrix_fd.solve(Ueq, Weq)
    assert np.allclose(U, p['exact'][0], rtol=1e-4, atol=1e-4) and np.allclose(W, p['exact'][1], rtol=1e-4, atol=1e-4)


def test_field_groups_assemble_and_solve(tango):
    """tests/test_fieldgroups.py:10-114: both group classes, through their own solve and through the module functions."""
    fg, mf = tango.fieldgroups, tango.multifield
    for p in (slab_problem(), disc_problem()):
        group = fg.UncoupledFieldGroup('default')
        eqn = group.Hcoeffs_to_matrix_eqn(p['dt'], p['dx'], p['UL'], p['U0'], mf.HCoefficients(H1=p['H1'], H2=p['H2'], H7=p['H7']))
        assert np.allclose(tango.HToMatrixFD.solve(eqn.A, eqn.B, eqn.C, eqn.f), p['exact'], rtol=1e-2)
        assert np.allclose(group.solve_matrix_eqn(eqn)['default'], p['exact'], rtol=1e-2)
    p = two_field_problem()
    pair = fg.PairCoupledFieldGroup('field0', 'field1')
    eqn = pair.Hcoeffs_to_matrix_eqn(p['dt'], p['dx'], p['rightBC'], p['psi0'], fg.JKCoefficients(**p['J'], **p['K']))
    U, W = tango.JK_to_matrix_fd.solve(eqn.UEqnCoeffs, eqn.WEqnCoeffs)
    assert np.allclose(U, p['exact'][0], rtol=1e-4) and np.allclose(W, p['exact'][1], rtol=1e-4)
    sol = pair.solve_matrix_eqn(eqn)
    assert np.allclose(sol['field0'], p['exact'][0], rtol=1e-4) and np.allclose(sol['field1'], p['exact'][1], rtol=1e-4)


def test_three_fields_through_the_group_chain(tango):
    """tests/test_fieldgroups.py:116-162, :192-235: one field on its own and a coupled pair, from Field objects and
    H coefficients to profiles."""
    fg, mf = tango.fieldgroups, tango.multifield
    one, two = disc_problem(), two_field_problem()
    fields = [mf.Field('field0', one['UL'], one['U0'], coupledTo=None),
              mf.Field('field1', two['rightBC'][0], two['psi0'][0], coupledTo='field2'),
              mf.Field('field2', two['rightBC'][1], two['psi0'][1], coupledTo='field1')]
    J, K = two['J'], two['K']
    H = {'field0': mf.HCoefficients(H1=one['H1'], H2=one['H2'], H7=one['H7']),
         'field1': mf.HCoefficients(**{'H' + k[1]: v for k, v in J.items()}),
         'field2': mf.HCoefficients(**{'H' + k[1]: v for k, v in K.items()})}
    groups = fg.fields_to_fieldgroups(fields, H)
    assert len(groups) == 2 and groups[0].label == 'field0' and (groups[1].label0, groups[1].label1) == ('field1', 'field2')
    assert groups[0].HCoeffs == H['field0'] and groups[0].rightBC == one['UL'] and np.allclose(groups[0].psi_mminus1, one['U0'])
    for k in '123678':
        assert np.allclose(getattr(groups[1].HCoeffs, 'J' + k), J['J' + k]) and np.allclose(getattr(groups[1].HCoeffs, 'K' + k), K['K' + k])
    assert np.allclose(groups[1].rightBC, two['rightBC'])
    dt, dx = two['dt'], two['dx']            # the reference's test steps both groups with the pair's dt and dx
    for g in groups:
        g.matrixEqn = g.Hcoeffs_to_matrix_eqn(dt, dx, g.rightBC, g.psi_mminus1, g.HCoeffs)
    for g in groups:
        g.profileSolution = g.solve_matrix_eqn(g.matrixEqn)
    profiles = fg.fieldgroups_to_profiles(groups)
    assert np.allclose(profiles['field0'], one['exact'], 1e-2)
    assert np.allclose(profiles['field1'], two['exact'][0], 1e-4) and np.allclose(profiles['field2'], two['exact'][1], 1e-4)


# ---- tests/solver_integration_test.py ---------------------------------------------------------------------------------
THETA = {'Dmin': 1e-5, 'Dmax': 1e13, 'dpdxThreshold': 10}


def shestakov_solver(tango, ewma=0.3, maxIterations=1000, tol=1e-11, **kw):
    N, nL = 500, 1e-2
    dx = 1.0 / (N - 1)
    x = np.arange(N) * dx
    sh = tango.extras.shestakov_nonlinear_diffusion

    def compute_all_H(t, x, profiles, HCoeffsTurb):
        return tango.multifield.HCoefficients(H1=np.ones_like(x), H7=sh.H7contrib_Source(x)) + HCoeffsTurb
    field = tango.multifield.Field(label='n', rightBC=nL, profile_mminus1=1 - 0.5 * x, compute_all_H=compute_all_H,
                                   lodestroMethod=tango.lodestro_method.lm(ewma, ewma, THETA))
    tango.multifield.check_fields_initialize([field])
    handler = tango.lodestro_method.TurbulenceHandler(dx, x, sh.AnalyticFluxModel(dx))
    allH = tango.multifield.ComputeAllHAllFields([field], handler)
    return tango.solver.Solver(1, x, np.array([0, 1e4]), maxIterations, tol, allH, [field], **kw), x, nL


def test_one_field_reaches_the_analytic_steady_state(tango):
    solver, x, nL = shestakov_solver(tango)                                        # :14-34
    while solver.ok:
        solver.take_timestep()
    nss = tango.extras.shestakov_nonlinear_diffusion.GetSteadyStateSolution(x, nL)
    r = (solver.profiles['n'] - nss) / np.max(np.abs(nss))
    assert np.sqrt(np.mean(r**2)) < 1e-3
    assert solver.reachedEnd and not solver.solutionError


def test_iteration_limit_stops_the_solver(tango):
    solver, _, _ = shestakov_solver(tango, maxIterations=60)                       # :202-232 without the HDF5 handler
    while solver.ok:
        solver.take_timestep()
    assert not solver.reachedEnd and len(solver.errHistoryFinal) == 60 and solver.errHistoryFinal[-1] > 1e-11


def test_small_ewma_parameter_still_converges(tango):
    solver, _, _ = shestakov_solver(tango, ewma=0.01, maxIterations=4000, tol=1e-9)   # :234-263
    while solver.ok:
        solver.take_timestep()
    assert solver.errHistoryFinal[-1] <= 1e-9 and solver.reachedEnd


def step_source(x, S0, delta, ramp=False):
    return np.where(x < delta, S0 * (x if ramp else 1.0), 0.0)


def collision_frequency(nu0, n, pi, pe):
    return nu0 / ((pi / n + pe / n) ** 1.5)


def running_integral(f, x):
    """Cumulative trapezoid from x[0] (the reference's test calls scipy's cumtrapz with initial=0)."""
    return np.concatenate(([0.0], np.cumsum(0.5 * (f[1:] + f[:-1]) * np.diff(x))))


@pytest.mark.parametrize('inner', [False, True])
def test_three_fields_carry_the_integrated_sources(tango, inner):
    """:36-102 and :104-170: at the steady state every flux equals the integral of its equation's right-hand side, with and
    without the inner iteration on the collision frequency.  A script's own flux model and callbacks."""
    N, nu0 = 500, 2.2
    dx = 1.0 / (N - 1)
    x = np.arange(N) * dx
    ddx = tango.derivatives.dx_centered_difference_edge_first_order

    class FluxModel(object):
        def get_flux(self, profiles):
            D = ddx(profiles['pi'], dx)**2 / profiles['pi']**2
            return {k: -D * ddx(profiles[k], dx) for k in ('n', 'pi', 'pe')}

    def callback(source, coupled):
        def compute_all_H(t, x, profiles, HCoeffsTurb):
            extra = {}
            if coupled:
                nu = collision_frequency(nu0, profiles['n'], profiles['pi'], profiles['pe'])
                extra = dict(H6=-nu, H8=nu)
            return tango.multifield.HCoefficients(H1=np.ones_like(x), H7=source, **extra) + HCoeffsTurb
        return compute_all_H
    sources = {'n': step_source(x, 8, 0.3), 'pi': step_source(x, 1, 0.1), 'pe': step_source(x, 3, 0.4, ramp=True)}
    start = 1 - 0.5 * x
    spec = (('n', 2, None), ('pi', 0.1, 'pe'), ('pe', 0.3, 'pi'))
    fields = [tango.multifield.Field(label=lab, rightBC=bc, profile_mminus1=1.0 * start, compute_all_H=callback(sources[lab], to is not None),
                                     lodestroMethod=tango.lodestro_method.lm(0.3, 0.3, THETA), coupledTo=to) for lab, bc, to in spec]
    tango.multifield.check_fields_initialize(fields)
    model = FluxModel()
    allH = tango.multifield.ComputeAllHAllFields(fields, tango.lodestro_method.TurbulenceHandler(dx, x, model))
    kw = dict(useInnerIteration=True, innerIterationMaxCount=2) if inner else {}
    solver = tango.solver.Solver(1, x, np.array([0, 1e6]), 1000, 1e-11, allH, fields, **kw)
    while solver.ok:
        solver.take_timestep()
    n, pi, pe = (solver.profiles[k] for k in ('n', 'pi', 'pe'))
    flux = model.get_flux(solver.profiles)
    nu = collision_frequency(nu0, n, pi, pe)
    rhs = {'n': sources['n'], 'pi': sources['pi'] - nu * (pi - pe), 'pe': sources['pe'] - nu * (pe - pi)}
    for k in ('n', 'pi', 'pe'):
        assert np.allclose(flux[k], running_integral(rhs[k], x), rtol=1e-2, atol=0.02), k


# ---- tests/test_interfacegrids_gene.py --------------------------------------------------------------------------------
def test_zero_extension_of_a_coarse_function(tango):
    """:8-71: zero outside the source grid; inside, the extended function resampled back is the original."""
    ig = tango.interfacegrids_gene
    xs = np.linspace(0, 2, 50)
    fs = 1 - (xs - 1)**2
    for extend, xl in ((ig.extend_with_zeros_both_sides, np.linspace(-0.3, 2.2, 221)), (ig.extend_with_zeros_left_side, np.linspace(-0.3, 1.7, 221))):
        fl = extend(xs, fs, xl, enforcePositive=True)
        assert np.all(fl[xl < xs[0]] == 0) and np.all(fl[xl > xs[-1]] == 0)
        inside = (xs > xl[0]) & (xs < xl[-1])
        inside[[0, -1]] = False                 # the end points may be disturbed by the interpolation
        back = tango._ops.spline_resample(xl, fl, xs[inside])
        assert np.allclose(back, fs[inside], atol=1e-4)


def test_extrapolation_known_answers(tango):
    """:74-119: the numbers written out in the reference's tests."""
    ig = tango.interfacegrids_gene
    assert np.isclose(ig.least_squares_slope(np.array([-1, 0, 1, 2]), np.array([7, 5, 3, 1]), 1, 3), -2, rtol=0, atol=1e-13)
    y = np.array([11.1, 9.7, 10.1, 8.1, 11, 8.5, 7.5])
    assert np.isclose(ig.least_squares_slope(np.arange(7), y, 6, 7.5), -0.58021978021978016, rtol=0, atol=1e-13)
    got = ig.extrap1d_constrained_linear_regression(np.array([3.0, 3.1, 3.5, 3.9, 4.1, 5, 7]), np.array([0, 0, 0.1, 0.5, 1, 1.3, 2]),
                                                    np.array([4.3, 5, 6.3, 7, 7.3, 8]), side='right', numPts=4)
    assert np.allclose(got, [0.90259, 1.18710, 1.71549, 2.0, 2.12193, 2.40644], rtol=0, atol=1e-5)
    y = np.array([5, 4.9, 4.75, 4.5, 4.2, 3.87, 3.6, 3.4, 3.25, 3.11, 2.97, 2.87])
    got = ig.truncate_on_left_extrapolate_on_right(np.arange(12.0), y, np.linspace(3.5, 14, 10), numPts=4)
    assert np.allclose(got, [4.35546444, 3.9775479, 3.63970781, 3.4, 3.22681231, 3.06202911, 2.91182052, 2.78809524, 2.6447619, 2.50142857],
                       rtol=0, atol=1e-6)
    got = ig.truncate_on_left_fixed_slope_on_right(np.array([1.0, 2.0, 3.0, 4.0]), np.array([1.0, 3.0, 4.5, 6.0]),
                                                   np.array([1.5, 2.5, 3.5, 4.0, 4.5, 5.5]), -0.2)
    assert np.allclose(got, [2.09375, 3.78125, 5.21875, 6., 5.9, 5.7], rtol=0, atol=1e-4)


# ---- tests/test_tridiag.py, tests/test_lodestro.py, tests/test_smoother.py, tests/test_derivatives.py -----------------
def test_small_known_answers_of_the_building_blocks(tango):
    N = 12                                                                         # tests/test_tridiag.py:7-25
    A, B, C = 0.1 * np.ones(N), -np.ones(N), 0.1 * np.ones(N)
    A[-1] = 0
    C[0] = 0
    D = np.linspace(0, 0.22, N)
    M = np.diag(B) + np.diag(A[:-1], 1) + np.diag(C[1:], -1)
    assert np.linalg.norm(tango.tridiag.solve(A, B, C, D) - np.linalg.solve(M, D)) < 1e-14
    assert np.isclose(tango.lodestro_method.EWMA._compute_next_ewma(37, 7.1, 0.12), 10.688, rtol=0, atol=1e-13)   # tests/test_lodestro.py:7-26
    ewma = tango.lodestro_method.EWMA(0.12)
    for y in (1.2, 2.2, -0.3, 3.4):
        last = ewma.next_ewma_iterate(y)
    assert np.isclose(last, 1.398528, rtol=0, atol=1e-13)
    x9 = np.array([1., 2., 9., 4., 10., 3., 1., 7., 5.])                           # tests/test_smoother.py:5-10
    assert np.allclose(tango.smoother.moving_average(x9, 5), [1., 4., 5.2, 5.6, 5.4, 5., 5.2, 13. / 3, 5], rtol=1e-14)
    u = np.array([2, 3, 7, 5, -3, 2, 8])                                           # tests/test_derivatives.py:5-20
    assert np.allclose(tango.derivatives.dx_centered_difference_edge_first_order(u, 0.4),
                       [5 / 2, 25 / 4, 5 / 2, -25 / 2, -15 / 4, 55 / 4, 15], rtol=0, atol=1e-13)
    assert np.allclose(tango.derivatives.dx_centered_difference(u, 0.4),
                       [-5 / 4, 25 / 4, 5 / 2, -25 / 2, -15 / 4, 55 / 4, 65 / 4], rtol=0, atol=1e-13)
