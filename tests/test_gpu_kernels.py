"""HIP kernels (called through tango_amd's reference-named API, i.e. through the C ABI) against the golden
vectors produced by the reference and against the CPU oracle.  Parity bar: 1e-10 relative for fp64 results
(stated per test; most are at the 1e-14 level)."""
import numpy as np
import pytest

from conftest import relerr
from oracle import tango_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-10


@pytest.fixture(scope='module')
def tango():
    import tango_amd
    assert tango_amd.device_count() > 0, 'GPU tests need a GPU'
    return tango_amd


def test_tridiag_golden(tango, golden):
    g = golden('g1_tridiag')
    for tag in ('t12', 'r8', 'r500', 'r8192'):
        A, B, C, D = (g[f'{tag}_{k}'] for k in 'ABCD')
        assert relerr(tango.tridiag.solve(A, B, C, D), g[f'{tag}_u']) < 1e-13
        # dgtsv's own sweeps (pivoting included; r500 and r8192 need 16 and 385 row interchanges): bit for bit
        assert np.array_equal(tango.tridiag.solve_with_scipy(A, B, C, D), g[f'{tag}_u'])
        assert relerr(tango.tridiag.solve_python(A, B, C, D), g[f'{tag}_u_py']) < 1e-13
    # the reference's own test (tests/test_tridiag.py:7-25)
    A, B, C, D = (g[f't12_{k}'] for k in 'ABCD')
    M = np.diag(B) + np.diag(A[:-1], 1) + np.diag(C[1:], -1)
    assert np.linalg.norm(tango.tridiag.solve(A, B, C, D) - np.linalg.solve(M, D)) < 1e-14


@pytest.mark.parametrize('n', [1, 2, 3, 7, 8, 9, 63, 64, 65, 255, 256, 500, 511, 512, 513, 1024, 1500, 4096, 8192, 10000, 16384])
def test_tridiag_sizes_batched(tango, n):
    rng = np.random.default_rng(n)
    nb = 5 if n < 4096 else 2
    A = rng.uniform(-1, 1, (nb, n)); C = rng.uniform(-1, 1, (nb, n))
    B = (np.abs(A) + np.abs(C) + rng.uniform(0.05, 1, (nb, n))) * rng.choice([-1.0, 1.0], (nb, 1))
    D = rng.normal(size=(nb, n))
    A[:, -1] = 7.0; C[:, 0] = -3.0          # must be ignored, as LAPACK's band storage ignores them
    ref = orc.tridiag_solve(A, B, C, D)
    assert relerr(tango.tridiag.solve(A, B, C, D), ref) < 1e-12
    # one profile given as a 1-D array returns a 1-D array
    one = tango.tridiag.solve(A[0], B[0], C[0], D[0])
    assert one.shape == (n,) and relerr(one, ref[0]) < 1e-12
    # batch-fastest (interleaved) layout
    dev = [tango.DeviceArray.from_host(np.ascontiguousarray(v.T)) for v in (A, B, C, D)]
    from tango_amd import _ops, _hip
    u = _ops.tridiag_solve(*dev, layout=_hip.BATCH_FASTEST).to_host().T
    assert np.array_equal(u, ref)       # the reference's LAPACK sweeps operation by operation


@pytest.mark.parametrize('n,nb', [(1, 3), (2, 8), (7, 9), (64, 5), (255, 16), (257, 7), (500, 13), (512, 8), (512, 16), (512, 48), (512, 37)])
def test_tridiag_interleaved_tiled(tango, golden, n, nb):
    """Interleaved (N, profiles) arrays solved by the point-fastest algorithm on tiles of 16 profiles: the oracle's
    result to 1e-12, and the point-fastest entry's bit for bit where that one gives a profile a whole wavefront too."""
    from tango_amd import _ops, _hip
    rng = np.random.default_rng(1000 * n + nb)
    A = rng.uniform(-1, 1, (nb, n)); C = rng.uniform(-1, 1, (nb, n))
    B = (np.abs(A) + np.abs(C) + rng.uniform(0.05, 1, (nb, n))) * rng.choice([-1.0, 1.0], (nb, 1))
    D = rng.normal(size=(nb, n))
    A[:, -1] = 7.0; C[:, 0] = -3.0
    if n == 500:                             # the reference's own 500-point system among them
        g = golden('g1_tridiag')
        A[0], B[0], C[0], D[0] = (g[f'r500_{k}'] for k in 'ABCD')
    ref = orc.tridiag_solve(A, B, C, D)
    dev = [tango.DeviceArray.from_host(np.ascontiguousarray(v.T)) for v in (A, B, C, D)]
    keep = [d.to_host() for d in dev]
    u = tango.tridiag.solve_interleaved(*dev).to_host()
    assert u.shape == (n, nb)
    bar = 1e-12 if n != 500 else 1e-11       # r500 needs row interchanges in LAPACK: not diagonally dominant
    assert relerr(u.T, ref) < bar
    assert all(np.array_equal(k, d.to_host()) for k, d in zip(keep, dev))
    if n > 256:
        assert np.array_equal(u.T, tango.tridiag.solve(A, B, C, D))
    # the exact variant on the same arrays, and the solution written over the right-hand side
    assert np.array_equal(tango.tridiag.solve_interleaved(*dev, exact=True).to_host().T, ref)
    status = tango.DeviceArray((nb,), np.int32, zero=True)
    _hip.check(_hip.lib.tg_tridiag_solve(dev[0].ptr, dev[1].ptr, dev[2].ptr, dev[3].ptr, dev[3].ptr, nb, n, _hip.BATCH_FASTEST_TILED,
                                         status.ptr, None, 0, _hip.stream()))
    assert np.array_equal(dev[3].to_host(), u) and not status.to_host().any()


def test_tridiag_interleaved_tiled_errors(tango):
    from tango_amd import _hip
    z = [tango.DeviceArray.from_host(np.zeros((3, 9))) for _ in range(3)] + [tango.DeviceArray.from_host(np.ones((3, 9)))]
    with pytest.raises(np.linalg.LinAlgError):
        tango.tridiag.solve_interleaved(*z)
    big = [tango.DeviceArray.from_host(np.ones((513, 2))) for _ in range(4)]
    with pytest.raises(NotImplementedError):
        tango.tridiag.solve_interleaved(*big)


def test_tridiag_inputs_untouched_and_errors(tango):
    A = np.array([0.1, 0.1, 0.0]); B = np.array([1.0, 1.0, 1.0]); C = np.array([0.0, 0.1, 0.1]); D = np.array([1.0, 2.0, 3.0])
    keep = [v.copy() for v in (A, B, C, D)]
    tango.tridiag.solve(A, B, C, D)
    assert all(np.array_equal(a, b) for a, b in zip(keep, (A, B, C, D)))
    with pytest.raises(AssertionError):
        tango.tridiag.solve(A, B[:2], C, D)
    with pytest.raises(AssertionError):
        tango.tridiag.solve_python(np.array([0.1, 0.1, 0.5]), B, C, D)
    with pytest.raises(np.linalg.LinAlgError):
        tango.tridiag.solve(np.zeros(3), np.zeros(3), np.zeros(3), np.ones(3))


def test_tridiag_picard_matrices_vs_extended_precision(tango):
    """On the ill-conditioned Picard matrices (cond ~1e8 at N=512, ~1e11 at N=8192) the GPU solve must be at
    least as close to an extended-precision solution as LAPACK's pivoted solve is."""
    def thomas_ld(A, B, C, D):
        A, B, C, D = (np.asarray(v, dtype=np.longdouble) for v in (A, B, C, D))
        n = len(B); cp = np.empty(n, np.longdouble); dp = np.empty(n, np.longdouble)
        cp[0] = A[0] / B[0]; dp[0] = D[0] / B[0]
        for j in range(1, n):
            den = B[j] - C[j] * cp[j - 1]; cp[j] = A[j] / den; dp[j] = (D[j] - C[j] * dp[j - 1]) / den
        u = np.empty(n, np.longdouble); u[-1] = dp[-1]
        for j in range(n - 2, -1, -1):
            u[j] = dp[j] - cp[j] * u[j + 1]
        return u.astype(float)
    # measured at N = 8192: 1.4e-10 from LAPACK's solution, which itself lies 1e-10 from the extended-precision one
    for N, bar in ((512, 1e-11), (8192, 3e-10)):
        dx = 1.0 / (N - 1); x = np.arange(N) * dx
        st = orc.State1F(1 - 0.5 * x)
        for _ in range(6):
            H2, H3, _ = orc.turb_coeffs_1f(st, dx, 0.3, 0.3)
            A, B, C, f = orc.h_to_matrix(1e4, dx, 1e-2, st.um1, np.ones(N), H2=H2, H3=H3, H7=orc.shestakov_source(x))
            ref = orc.tridiag_solve(A, B, C, f)
            st.u = ref
        exact = thomas_ld(A, B, C, f)
        got = tango.tridiag.solve(A, B, C, f)
        assert relerr(got, ref) < bar
        assert np.array_equal(tango.tridiag.solve_with_scipy(A, B, C, f), ref)
        assert relerr(got, exact) <= max(2 * relerr(ref, exact), 1e-13)


@pytest.mark.parametrize('tag,keys', [('cart', ('H2', 'H7')), ('polar', ('H2', 'H7')), ('conv', ('H2', 'H3', 'H7')),
                                      ('h4', ('H2', 'H4', 'H7')), ('rnd', ('H2', 'H3', 'H4', 'H6', 'H7'))])
def test_h_to_matrix_golden(tango, golden, tag, keys):
    g = golden('g2_assembly')
    kw = {k: g[f'{tag}_{k}'] for k in keys}
    out = tango.HToMatrixFD.H_to_matrix(float(g[f'{tag}_dt']), float(g[f'{tag}_dx']), float(g[f'{tag}_UL']),
                                        g[f'{tag}_U'], g[f'{tag}_H1'], **kw)
    for got, k in zip(out, 'ABCD'):
        assert got.shape == g[f'{tag}_{k}'].shape
        assert relerr(got, g[f'{tag}_{k}']) < 1e-14, k
    if f'{tag}_u' in g:
        u = tango.HToMatrixFD.H_to_matrix_and_solve(float(g[f'{tag}_dt']), float(g[f'{tag}_dx']), float(g[f'{tag}_UL']),
                                                    g[f'{tag}_U'], g[f'{tag}_H1'], **kw)
        assert relerr(u, g[f'{tag}_u']) < TOL


def test_h_to_matrix_h3_only_and_batched(tango, golden):
    g = golden('g2_assembly')
    dt, dx, UL = float(g['rnd_dt']), float(g['rnd_dx']), float(g['rnd_UL'])
    out = tango.HToMatrixFD.H_to_matrix(dt, dx, UL, g['rnd_U'], g['rnd_H1'], H3=g['rnd_H3'], H7=g['rnd_H7'])
    for got, k in zip(out, 'ABCD'):
        assert relerr(got, g[f'h3only_{k}']) < 1e-14
    # ensemble: per-profile U, H2, UL, dt; shared H1, H3, H7
    U = np.stack([g['rnd_U'], g['rnd_U'][::-1], g['rnd_U'] * 2])
    H2 = np.stack([g['rnd_H2'], g['rnd_H2'] * 2, g['rnd_H2'] * 0.5])
    ULs = np.array([1.7, 0.3, 9.0]); dts = np.array([0.37, 5.0, 1e4])
    ref = orc.h_to_matrix(dts, dx, ULs, U, g['rnd_H1'], H2=H2, H3=g['rnd_H3'], H7=g['rnd_H7'])
    got = tango.HToMatrixFD.H_to_matrix(dts, dx, ULs, U, g['rnd_H1'], H2=H2, H3=g['rnd_H3'], H7=g['rnd_H7'])
    for a, b in zip(got, ref):
        assert a.shape == (3, 257) and relerr(a, b) < 1e-14
    with pytest.raises(AssertionError):
        tango.HToMatrixFD.H_to_matrix(dt, dx, UL, g['rnd_U'], g['rnd_H1'][:-1])


def test_flux_split_golden(tango, golden):
    g = golden('g4_fluxsplit')
    th = {'Dmin': float(g['shest_Dmin']), 'Dmax': float(g['shest_Dmax']), 'dpdxThreshold': float(g['shest_thr'])}
    fs = tango.lodestro_method.FluxSplit(th)
    for tag in ('it0', 'it20'):
        flux = tango.extras.shestakov_nonlinear_diffusion.get_flux(g[f'{tag}_p'], float(g['shest_dx']))
        assert relerr(flux, g[f'{tag}_flux']) < 1e-14
        D, c, data = fs.flux_to_transport_coeffs(g[f'{tag}_flux'], g[f'{tag}_p'], float(g['shest_dx']))
        assert relerr(D, g[f'{tag}_D']) < 1e-14 and relerr(data['DHat'], g[f'{tag}_DHat']) < 1e-14
        assert relerr(data['cHat'], g[f'{tag}_cHat']) < 1e-14
        assert np.array_equal(data['theta'], g[f'{tag}_theta'])       # theta must round exactly as the reference's
        assert relerr(c, g[f'{tag}_c']) < 1e-13
    th2 = {'Dmin': float(g['syn_Dmin']), 'Dmax': float(g['syn_Dmax']), 'dpdxThreshold': float(g['syn_thr'])}
    for tag, kw in (('syn', {}), ('geo', dict(gxxAvg=g['geo_gxx'], gradxAvg=g['geo_gradx']))):
        D, c, data = tango.lodestro_method.FluxSplit(th2, **kw).flux_to_transport_coeffs(g['syn_flux'], g['syn_p'], float(g['syn_dx']))
        for got, k in ((D, 'D'), (c, 'c'), (data['DHat'], 'DHat'), (data['cHat'], 'cHat'), (data['theta'], 'theta')):
            exp = g[f'{tag}_{k}']
            assert np.array_equal(np.isnan(got), np.isnan(exp)), k
            m = ~np.isnan(exp)
            assert relerr(got[m], exp[m]) < 1e-13, (tag, k)
    # custom theta callable (tango/lodestro_method.py:557-561)
    th3 = dict(th2, custom_ftheta=lambda DHat, dpdx, p: np.full_like(DHat, 0.25))
    D, c, data = tango.lodestro_method.FluxSplit(th3).flux_to_transport_coeffs(g['syn_flux'], g['syn_p'], float(g['syn_dx']))
    m = ~np.isnan(g['syn_DHat'])
    assert relerr(D[m], 0.25 * g['syn_DHat'][m]) < 1e-14 and relerr(c[m], 0.75 * g['syn_cHat'][m]) < 1e-14


def test_small_known_answers(tango, golden):
    g = golden('g567_small')
    e = tango.lodestro_method.EWMA(float(g['ewma_param']))
    for y, exp in zip(g['ewma_in'], g['ewma_out']):
        assert abs(e.next_ewma_iterate(float(y)) - exp) < 1e-15
    assert abs(tango.lodestro_method.EWMA._compute_next_ewma(37, 7.1, 0.12) - 10.688) < 1e-13
    # array EWMA through the kernel, including the seeding call
    rng = np.random.default_rng(5)
    ys = rng.normal(size=(4, 3, 33))
    ea = tango.lodestro_method.EWMA(0.12); prev = None
    for y in ys:
        prev = orc.ewma_next(y, prev, 0.12)
        assert relerr(ea.next_ewma_iterate(y), prev) < 1e-15
    for w in (1, 3, 4, 5, 11):
        assert relerr(tango.smoother.Smoother(w).smooth(g['sm_x']), g[f'sm_w{w}']) < 1e-14
    assert np.allclose(tango.smoother.moving_average(g['sm9_x'], 5), [1., 4., 5.2, 5.6, 5.4, 5., 5.2, 13. / 3, 5], rtol=1e-14)
    with pytest.raises(AssertionError):
        tango.smoother.moving_average(g['sm_x'], 4)
    assert np.allclose(tango.derivatives.dx_centered_difference_edge_first_order(g['d_u'], 0.4), g['d_edge1'], rtol=0, atol=1e-13)
    assert np.allclose(tango.derivatives.dx_centered_difference(g['d_u'], 0.4), g['d_edge2'], rtol=0, atol=1e-13)


def test_residual_golden(tango, golden):
    g = golden('g11_residual')
    ufg = tango.fieldgroups.UncoupledFieldGroup('n')
    r = ufg.calculate_residual(tango.fieldgroups.UncoupledMatrixEqn(g['A'], g['B'], g['C'], g['f']), {'n': g['u']})['n']
    assert relerr(r, g['r']) < 1e-14
    from tango_amd import _ops
    _, ss = _ops.residual(g['A'], g['B'], g['C'], g['f'], g['u'])
    assert abs(np.sqrt(ss / 300) - float(g['rms'])) < 1e-14 * float(g['rms']) * 10


def test_uncoupled_fieldgroup_chain(tango):
    """tests/test_fieldgroups.py:28-66 of the reference: assemble + solve reproduce analytic steady states."""
    N = 500; dx = 1 / (N - 0.5); x = np.linspace(dx / 2, 1, N)
    ufg = tango.fieldgroups.UncoupledFieldGroup('default')
    H = tango.multifield.HCoefficients(H1=x, H2=1.3 * x, H7=x ** 2)
    meq = ufg.Hcoeffs_to_matrix_eqn(1e9, dx, 0.03, 0.3 * np.sin(np.pi * x) + 0.03 * x, H)
    sol = ufg.solve_matrix_eqn(meq)['default']
    assert np.allclose(sol, 0.03 + 1 / (9 * 1.3) * (1 - x ** 3), rtol=1e-4)


def test_grid_mapping_golden(tango, golden):
    """tango/interfacegrids_gene.py:35-87: profile onto the turbulence grid, coefficients back with zeros outside;
    a non-uniform grid with extrapolated end points; batched rows."""
    g = golden('g16_gridmap')
    ig = tango.interfacegrids_gene
    gm = ig.GridInterfaceTangoOutside(g['xT'], g['xG'])
    assert relerr(gm.map_profile_onto_turb_grid(g['prof']), g['prof_on_G']) < 1e-12
    D, c = gm.map_transport_coeffs_onto_transport_grid(g['Dg'], g['cg'])
    assert relerr(D, g['D_on_T']) < 1e-12 and relerr(c, g['c_on_T']) < 1e-12
    assert np.all(D >= 0) and D[0] == 0 and D[-1] == 0
    assert np.array_equal(gm.get_x_transport_grid(), g['xT']) and np.array_equal(gm.get_x_turbulence_grid(), g['xG'])
    from tango_amd import _ops
    assert relerr(_ops.spline_resample(g['xn'], g['yn'], g['xq']), g['yq']) < 1e-11
    rows = np.stack([g['prof'], 2 * g['prof'] + 1, g['prof'] ** 2])
    out = gm.map_profile_onto_turb_grid(rows)
    assert out.shape == (3, len(g['xG'])) and relerr(out[0], g['prof_on_G']) < 1e-12
    assert relerr(out[1], 2 * g['prof_on_G'] + 1) < 1e-12
    with pytest.raises(AssertionError):
        ig.GridInterfaceTangoOutside(g['xG'], g['xT'])
    # coefficients extrapolated by a fitted polynomial beyond the trusted part of the turbulence grid (:90-178, :248-298)
    ge = ig.TangoOutsideExtrapCoeffs(g['xT'], g['xG'], 0.55, 0.7, 2)
    assert relerr(ge.map_profile_onto_turb_grid(g['prof']), g['ext_prof_on_G']) < 1e-12
    D, c = ge.map_transport_coeffs_onto_transport_grid(g['Dg'], g['cg'])
    assert relerr(D, g['ext_D_on_T']) < 1e-11 and relerr(c, g['ext_c_on_T']) < 1e-11
    D2, c2 = ge.map_transport_coeffs_onto_transport_grid(np.stack([g['Dg'], 2 * g['Dg']]), np.stack([g['cg'], -g['cg']]))
    assert relerr(D2[0], g['ext_D_on_T']) < 1e-11 and relerr(c2[1], -g['ext_c_on_T']) < 1e-11


@pytest.mark.parametrize('path', ['pipeline', 'pipeline, python flux model', 'generic'])
def test_solver_with_grid_mapper(tango, golden, path):
    """A one-field run whose flux model lives on a finer, shorter turbulence grid, against the reference's trajectory:
    the device-resident pipeline (resampling, EWMA, flux, split and the fused step with no host array in between; with
    the packaged flux model and with a Python flux model called on the host) and the generic path (every map a kernel
    call, host arrays in between)."""
    g = golden('g16_gridmap')
    N, NG = int(g['run_N']), int(g['run_NG'])
    x = np.linspace(0, 1, N)
    xg = np.linspace(0.1, 0.9, NG); dxg = xg[1] - xg[0]
    sh = tango.extras.shestakov_nonlinear_diffusion
    lm = tango.lodestro_method.lm(0.3, 1.0, {'Dmin': 1e-5, 'Dmax': 1e13, 'dpdxThreshold': 10})
    cb = sh.ComputeAllH(1.0, 0.3)
    if path == 'generic':
        cb = (lambda inner: (lambda t, xx, p, HT: inner(t, xx, p, HT)))(cb)        # a plain callable: no static_H attribute

    class PythonFlux(object):
        def get_flux(self, profiles):
            return {'n': np.asarray(sh.get_flux(profiles['n'], dxg))}
    model = PythonFlux() if 'python' in path else sh.AnalyticFluxModel(dxg)
    f = tango.multifield.Field(label='n', rightBC=0.05, profile_mminus1=1 - 0.5 * x, compute_all_H=cb,
                               gridMapper=tango.interfacegrids_gene.GridInterfaceTangoOutside(x, xg), lodestroMethod=lm)
    tango.multifield.check_fields_initialize([f])
    call = tango.multifield.ComputeAllHAllFields([f], tango.lodestro_method.TurbulenceHandler(dxg, x, model))
    s = tango.solver.Solver(1, x, np.array([0, 0.5]), 25, 1e-14, call, [f], detectStaticH=False, iterationsPerCheck=4)
    assert (type(s._fused).__name__ == '_FusedMapped1F') == (path != 'generic') and (s._fused is None) == (path == 'generic')
    s.take_timestep()
    assert len(s.errHistoryFinal) == 25
    assert np.allclose(s.errHistoryFinal, g['run_err'], rtol=1e-7, atol=1e-13)
    assert relerr(s.profilesAllIterations['n'][2], g['run_it2']) < 1e-10
    assert relerr(s.profiles['n'], g['run_final']) < 1e-10
    assert relerr(lm._EWMATurbD._yEWMA_lminus1, g['run_DE']) < 1e-10


def test_extrapolated_coefficients_on_the_device(tango, golden):
    """The back-map of the extrapolating mappers as the device pipeline does it (spline over the trusted range, the
    least-squares polynomial beyond it as a fixed matrix applied by tg_zone_apply) against the reference's own maps of
    sample coefficients (golden g16: outer boundary only, g17: both boundaries)."""
    from tango_amd import _hip as H
    from tango_amd.solver import _FusedMapped1F
    g16, g17 = golden('g16_gridmap'), golden('g17_gridmap_more')
    xT, xG = g17['xT'], g17['xG']
    ig = tango.interfacegrids_gene
    D = H.DeviceArray
    xT_d, xG_d = D.from_host(xT), D.from_host(xG)
    work = D((int(H.lib.tg_spline_work_bytes(2, xG.size)) // 8,))
    cases = ((ig.TangoOutsideExtrapCoeffs(xT, xG, 0.55, 0.7, 2), g16['ext_D_on_T'], g16['ext_c_on_T']),
             (ig.TangoOutsideExtrapCoeffsBothSides(xT, xG, 0.2, 0.32, 0.55, 0.7, 2), g17['both_D_on_T'], g17['both_c_on_T']))
    for mapper, D_ref, c_ref in cases:
        lo, hi, zones = _FusedMapped1F._back_map(mapper, xT, xG)
        assert len(zones) == (2 if 'BothSides' in type(mapper).__name__ else 1)
        for src, nonneg, ref in ((g17['Dg'], True, D_ref), (g17['cg'], False, c_ref)):
            src_d = D.from_host(np.stack([src, 0.5 * src]))             # two profiles: the second checks the row strides
            out = D((2, xT.size), zero=True)
            H.check(H.lib.tg_spline_resample(xG_d.ptr, src_d.ptr, 2, xG.size, xT_d.ptr, xT.size, lo, hi, int(nonneg), out.ptr, work.ptr,
                                             work.nbytes, H.stream()))
            for z_lo, z_n, M, o_lo, rows in zones:
                H.check(H.lib.tg_zone_apply(src_d.ptr, 2, xG.size, z_lo, z_n, M.ptr, None, out.ptr, xT.size, o_lo, rows, int(nonneg), H.stream()))
            got = out.to_host()
            assert relerr(got[0], ref) < 1e-12 and relerr(got[1], 0.5 * ref) < 1e-12
            if nonneg:
                assert got.min() >= 0.0 and np.any(ref == 0.0)


@pytest.mark.parametrize('tag', ['outer', 'both'])
@pytest.mark.parametrize('path', ['pipeline', 'generic'])
def test_solver_with_extrapolating_grid_mapper(tango, golden, tag, path):
    """The mappers that extrapolate the coefficients beyond the trusted part of the turbulence grid (the one the reference's
    GENE drivers use, and its two-sided variant), against the reference's trajectories: device-resident pipeline (the fit
    as a fixed matrix, tg_zone_apply) and generic path (NumPy fit on the host)."""
    g = golden('g19_gridmap_extrap_runs')
    N, NG = int(g['run_N']), int(g['run_NG'])
    x = np.linspace(0, 1, N)
    xg = np.linspace(0.1, 0.9, NG); dxg = xg[1] - xg[0]
    sh = tango.extras.shestakov_nonlinear_diffusion
    ig = tango.interfacegrids_gene
    mapper = ig.TangoOutsideExtrapCoeffs(x, xg, 0.6, 0.8, 1) if tag == 'outer' else ig.TangoOutsideExtrapCoeffsBothSides(x, xg, 0.2, 0.35, 0.6, 0.8, 2)
    lm = tango.lodestro_method.lm(0.3, 1.0, {'Dmin': 1e-5, 'Dmax': 1e13, 'dpdxThreshold': 10})
    cb = sh.ComputeAllH(1.0, 0.3)
    if path == 'generic':
        cb = (lambda inner: (lambda t, xx, p, HT: inner(t, xx, p, HT)))(cb)
    f = tango.multifield.Field(label='n', rightBC=0.05, profile_mminus1=1 - 0.5 * x, compute_all_H=cb, gridMapper=mapper, lodestroMethod=lm)
    tango.multifield.check_fields_initialize([f])
    call = tango.multifield.ComputeAllHAllFields([f], tango.lodestro_method.TurbulenceHandler(dxg, x, sh.AnalyticFluxModel(dxg)))
    s = tango.solver.Solver(1, x, np.array([0, 0.5]), 25, 1e-14, call, [f], detectStaticH=False, iterationsPerCheck=5)
    assert (type(s._fused).__name__ == '_FusedMapped1F') == (path == 'pipeline')
    s.take_timestep()
    assert len(s.errHistoryFinal) == len(g[tag + '_err'])
    assert relerr(s.profilesAllIterations['n'][2], g[tag + '_it2']) < 1e-10
    assert relerr(s.profiles['n'], g[tag + '_final']) < 1e-10
    assert relerr(lm._EWMATurbD._yEWMA_lminus1, g[tag + '_DE']) < 1e-10
    # measured: profiles 1.5e-11, D 1.2e-11; the late residuals (5e-5 and below, differences of O(1) terms) then agree to 1e-6
    assert np.allclose(s.errHistoryFinal, g[tag + '_err'], rtol=2e-5, atol=1e-12)


@pytest.mark.parametrize('slope', ['fitted', 'imposed'])
@pytest.mark.parametrize('path', ['pipeline', 'generic'])
def test_solver_with_turbulence_grid_beyond_the_transport_grid(tango, golden, path, slope):
    """GridInterfaceTangoInside / ...FixedOutside: the turbulence grid ends beyond the outer end of the transport grid and the
    profile is continued there by the constrained regression line (on the pipeline: a fixed matrix over its last ten points)
    or by a line of imposed slope (the end point plus a fixed bias).  The
    reference's trajectory, and its map of a profile whose continuation turns negative (cut off at zero)."""
    from tango_amd import _hip as H
    g = golden('g20_gridmap_inside_run')
    N, NG = int(g['run_N']), int(g['run_NG'])
    x = np.linspace(0, 0.8, N)
    xg = np.linspace(0.1, 1.0, NG); dxg = xg[1] - xg[0]
    sh = tango.extras.shestakov_nonlinear_diffusion
    ig = tango.interfacegrids_gene
    mapper = ig.GridInterfaceTangoInside(x, xg) if slope == 'fitted' else ig.GridInterfaceTangoInsideFixedOutside(x, xg, -1.2)
    k = 'run_' if slope == 'fitted' else 'fix_'
    lm = tango.lodestro_method.lm(0.3, 1.0, {'Dmin': 1e-5, 'Dmax': 1e13, 'dpdxThreshold': 10})
    cb = sh.ComputeAllH(1.0, 0.3)
    if path == 'generic':
        cb = (lambda inner: (lambda t, xx, p, HT: inner(t, xx, p, HT)))(cb)
    f = tango.multifield.Field(label='n', rightBC=0.25, profile_mminus1=1 - 0.9 * x, compute_all_H=cb, gridMapper=mapper, lodestroMethod=lm)
    tango.multifield.check_fields_initialize([f])
    call = tango.multifield.ComputeAllHAllFields([f], tango.lodestro_method.TurbulenceHandler(dxg, x, sh.AnalyticFluxModel(dxg)))
    s = tango.solver.Solver(0.8, x, np.array([0, 0.5]), 25, 1e-14, call, [f], detectStaticH=False, iterationsPerCheck=5)
    assert (type(s._fused).__name__ == '_FusedMapped1F') == (path == 'pipeline')
    s.take_timestep()
    assert len(s.errHistoryFinal) == len(g[k + 'err'])
    assert relerr(s.profilesAllIterations['n'][2], g[k + 'it2']) < 1e-10
    assert relerr(s.profiles['n'], g[k + 'final']) < 1e-10
    assert relerr(lm._EWMATurbD._yEWMA_lminus1, g[k + 'DE']) < 1e-10 and relerr(lm._EWMAProfile._yEWMA_lminus1, g[k + 'uE']) < 1e-10
    assert np.allclose(s.errHistoryFinal, g[k + 'err'], rtol=2e-5, atol=1e-12)
    if path == 'pipeline':
        fu = s._fused
        src = H.DeviceArray.from_host(g['steep'][None, :])
        out = H.DeviceArray((1, NG), zero=True)
        fu._resample(fu.xT, src, N, fu.xG, NG, fu.fwd_lo, fu.fwd_hi, fu.fwd_nonneg, out)
        for z_lo, z_n, M, o_lo, rows, bias in fu.fwd_zones:
            H.check(H.lib.tg_zone_apply(src.ptr, 1, N, z_lo, z_n, M.ptr, H.ptr_of(bias), out.ptr, NG, o_lo, rows, 1, H.stream()))
        got = out.to_host()[0]
        assert relerr(got, g['steep_on_G' if slope == 'fitted' else 'fix_steep_on_G']) < 1e-12 and got.min() == 0.0 and np.count_nonzero(got == 0.0) == np.count_nonzero(g['steep_on_G' if slope == 'fitted' else 'fix_steep_on_G'] == 0.0)
    else:
        assert relerr(mapper.map_profile_onto_turb_grid(g['steep']), g['steep_on_G' if slope == 'fitted' else 'fix_steep_on_G']) < 1e-12


def test_grid_mapper_pipeline_ensemble(tango):
    """The device-resident mapped pipeline on an ensemble whose members stop on their own iterations, member by member against
    the same run on the generic path (same kernels for every map, host arrays in between)."""
    N, NG, nb = 96, 200, 5
    x = np.linspace(0, 1, N)
    xg = np.linspace(0.12, 0.88, NG); dxg = xg[1] - xg[0]
    sh = tango.extras.shestakov_nonlinear_diffusion
    S0 = np.linspace(0.8, 1.2, nb); nL = np.linspace(0.04, 0.06, nb)
    u0 = np.linspace(0.9, 1.1, nb)[:, None] * (1 - 0.5 * x)[None, :]
    # with the flux smoothed over 5 points this problem stops contracting after a few iterations (in the reference too: its
    # residuals 15.7, 1.72, ... are the ones found here), so that case is compared over its first four iterations
    for window, iterations in ((None, 25), (5, 4)):
        res = []
        for generic in (False, True):
            lm = tango.lodestro_method.lm(0.3, 1.0, {'Dmin': 1e-5, 'Dmax': 1e13, 'dpdxThreshold': 10})
            cb = sh.ComputeAllH(S0, 0.3)
            if generic:
                cb = (lambda inner: (lambda t, xx, p, HT: inner(t, xx, p, HT)))(cb)
            f = tango.multifield.Field(label='n', rightBC=nL, profile_mminus1=u0, compute_all_H=cb,
                                       gridMapper=tango.interfacegrids_gene.GridInterfaceTangoOutside(x, xg), lodestroMethod=lm)
            tango.multifield.check_fields_initialize([f])
            handler = tango.lodestro_method.TurbulenceHandler(dxg, x, sh.AnalyticFluxModel(dxg),
                                                              fluxSmoother=tango.smoother.Smoother(window) if window else None)
            call = tango.multifield.ComputeAllHAllFields([f], handler)
            s = tango.solver.Solver(1, x, np.array([0, 0.5]), iterations, 1e-30, call, [f], detectStaticH=False, iterationsPerCheck=7)
            assert (s._fused is None) == generic
            s.take_timestep()
            err = np.array(s.errHistory)
            res.append((np.array(s.profiles['n']), err if generic else err.max(axis=0), np.array(lm._EWMATurbc._yEWMA_lminus1),
                        np.array(lm._EWMATurbD._yEWMA_lminus1)))
        if window:
            assert relerr(res[1][1][1:3], [15.7179152, 1.71811159]) < 1e-8      # the reference's own residuals
        for a, b in zip(res[0], res[1]):
            assert a.shape == b.shape and relerr(a, b) < 2e-9


def test_grid_mapping_more_golden(tango, golden):
    """tango/interfacegrids_gene.py:181-245, :300-443, :528-692: coefficients extrapolated at both boundaries; a
    turbulence grid that reaches beyond the transport grid (fitted and imposed outward slope); batched rows."""
    g = golden('g17_gridmap_more')
    ig = tango.interfacegrids_gene
    gb = ig.TangoOutsideExtrapCoeffsBothSides(g['xT'], g['xG'], 0.2, 0.32, 0.55, 0.7, 2)
    assert relerr(gb.map_profile_onto_turb_grid(g['prof']), g['both_prof_on_G']) < 1e-12
    D, c = gb.map_transport_coeffs_onto_transport_grid(g['Dg'], g['cg'])
    assert relerr(D, g['both_D_on_T']) < 1e-11 and relerr(c, g['both_c_on_T']) < 1e-11
    D2, c2 = gb.map_transport_coeffs_onto_transport_grid(np.stack([g['Dg'], 2 * g['Dg']]), np.stack([g['cg'], -g['cg']]))
    assert relerr(D2[0], g['both_D_on_T']) < 1e-11 and relerr(c2[1], -g['both_c_on_T']) < 1e-11
    gi = ig.GridInterfaceTangoInside(g['xTi'], g['xGi'])
    assert relerr(gi.map_profile_onto_turb_grid(g['profi']), g['in_prof_on_G']) < 1e-12
    assert relerr(gi.map_profile_onto_turb_grid(g['steep']), g['in_steep_on_G']) < 1e-12
    rows = gi.map_profile_onto_turb_grid(np.stack([g['profi'], g['steep']]))
    assert relerr(rows[0], g['in_prof_on_G']) < 1e-12 and relerr(rows[1], g['in_steep_on_G']) < 1e-12
    D, c = gi.map_transport_coeffs_onto_transport_grid(g['Dgi'], g['cgi'])
    assert relerr(D, g['in_D_on_T']) < 1e-12 and relerr(c, g['in_c_on_T']) < 1e-12
    assert np.array_equal(gi.get_x_transport_grid(), g['xTi']) and np.array_equal(gi.get_x_turbulence_grid(), g['xGi'])
    gf = ig.GridInterfaceTangoInsideFixedOutside(g['xTi'], g['xGi'], -1.7)
    assert relerr(gf.map_profile_onto_turb_grid(g['profi']), g['fix_prof_on_G']) < 1e-12
    assert relerr(gf.map_profile_onto_turb_grid(g['steep']), g['fix_steep_on_G']) < 1e-12
    rows = gf.map_profile_onto_turb_grid(np.stack([g['steep'], g['profi']]))
    assert relerr(rows[0], g['fix_steep_on_G']) < 1e-12 and relerr(rows[1], g['fix_prof_on_G']) < 1e-12
    D, c = gf.map_transport_coeffs_onto_transport_grid(g['Dgi'], g['cgi'])
    assert relerr(D, g['fix_D_on_T']) < 1e-12 and relerr(c, g['fix_c_on_T']) < 1e-12
    with pytest.raises(AssertionError):
        ig.GridInterfaceTangoInsideFixedOutside(g['xTi'], g['xGi'], 0.5)
    with pytest.raises(AssertionError):
        ig.GridInterfaceTangoInside(g['xGi'], g['xTi'])
    assert abs(ig.least_squares_slope(g['xTi'][-10:], g['profi'][-10:], g['xTi'][-1], g['profi'][-1]) - float(g['slope'])) < 1e-13
    assert relerr(ig.extrap1d_constrained_linear_regression(g['xTi'], g['profi'], np.linspace(-0.2, 0.1, 7), side='left', numPts=6),
                  g['reg_left']) < 1e-13
    assert relerr(ig.make_extrapolator(g['xTi'], g['profi'], 'left', 6)(np.linspace(-0.2, 0.5, 23)), g['ext_left']) < 1e-12
