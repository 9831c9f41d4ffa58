"""The CPU oracle (oracle/tango_oracle.py) against the golden vectors produced by importing the
reference (tests/golden/make_golden.py).  The oracle repeats the reference's arithmetic in the
same order, so agreement is at the last-ulp level; tolerances below are 1e-13 relative unless
the comparison crosses a differently-rounded path (then stated)."""
import numpy as np
import pytest

from conftest import relerr
from oracle import tango_oracle as orc

TIGHT = 1e-13


def test_g1_tridiag(golden):
    g = golden('g1_tridiag')
    for tag in ('t12', 'r8', 'r500', 'r8192'):
        A, B, C, D = (g[f'{tag}_{k}'] for k in 'ABCD')
        assert relerr(orc.tridiag_solve(A, B, C, D), g[f'{tag}_u']) == 0.0
        # non-pivoting elimination vs LAPACK dgtsv and vs the reference's own Thomas
        assert relerr(orc.thomas(A, B, C, D), g[f'{tag}_u']) < 1e-12
        assert relerr(orc.thomas(A, B, C, D), g[f'{tag}_u_py']) < 1e-12
    # the reference's own known-answer test: dense solve of the N=12 system
    A, B, C, D = (g[f't12_{k}'] for k in 'ABCD')
    M = np.diag(B) + np.diag(A[:-1], 1) + np.diag(C[1:], -1)
    assert np.linalg.norm(orc.tridiag_solve(A, B, C, D) - np.linalg.solve(M, D)) < 1e-14


@pytest.mark.parametrize('tag,keys', [('cart', ('H2', 'H7')), ('polar', ('H2', 'H7')),
                                      ('conv', ('H2', 'H3', 'H7')), ('h4', ('H2', 'H4', 'H7')),
                                      ('rnd', ('H2', 'H3', 'H4', 'H6', 'H7'))])
def test_g2_assembly(golden, tag, keys):
    g = golden('g2_assembly')
    kw = {k: g[f'{tag}_{k}'] for k in keys}
    A, B, C, D = orc.h_to_matrix(float(g[f'{tag}_dt']), float(g[f'{tag}_dx']), float(g[f'{tag}_UL']),
                                 g[f'{tag}_U'], g[f'{tag}_H1'], **kw)
    for got, k in zip((A, B, C, D), 'ABCD'):
        assert np.array_equal(got, g[f'{tag}_{k}']), k
    if f'{tag}_u' in g:
        assert relerr(orc.tridiag_solve(A, B, C, D), g[f'{tag}_u']) == 0.0


def test_g2_beta_and_h3only(golden):
    g = golden('g2_assembly')
    assert np.array_equal(orc.half_grid(g['rnd_H2']), g['rnd_H2h'])
    beta = orc.fbeta(g['rnd_H2h'], g['rnd_H3h'], float(g['rnd_dx']))
    assert np.array_equal(beta, g['rnd_beta'])
    # every branch of the upwinding rule occurs in this vector
    c = -g['rnd_H3h']
    assert (c > 0).any() and (c < 0).any() and (c == 0).any()
    assert (beta == 0.5).any() and ((beta > 0.5) & (c > 0)).any() and ((beta < 0.5) & (c < 0)).any()
    A, B, C, D = orc.h_to_matrix(float(g['rnd_dt']), float(g['rnd_dx']), float(g['rnd_UL']), g['rnd_U'],
                                 g['rnd_H1'], H3=g['rnd_H3'], H7=g['rnd_H7'])
    for got, k in zip((A, B, C, D), 'ABCD'):
        assert np.array_equal(got, g[f'h3only_{k}']), k


def test_g2_batched_equals_single(golden):
    g = golden('g2_assembly')
    U = np.stack([g['rnd_U'], g['rnd_U'][::-1]])
    H2 = np.stack([g['rnd_H2'], g['rnd_H2'] * 2])
    UL = np.array([1.7, 0.3])
    dt = np.array([0.37, 5.0])
    out = orc.h_to_matrix(dt, float(g['rnd_dx']), UL, U, g['rnd_H1'], H2=H2, H3=g['rnd_H3'], H7=g['rnd_H7'])
    for b in range(2):
        one = orc.h_to_matrix(dt[b], float(g['rnd_dx']), UL[b], U[b], g['rnd_H1'], H2=H2[b], H3=g['rnd_H3'],
                              H7=g['rnd_H7'])
        for x, y in zip(out, one):
            assert np.array_equal(x[b], y)


def test_g3_pair(golden):
    g = golden('g3_pair')
    z = np.zeros_like(g['x'])
    J = {f'H{i}': g[f'J{i}'] for i in (1, 2, 3, 6, 7, 8)}
    K = {f'H{i}': g[f'K{i}'] for i in (1, 2, 3, 6, 7, 8)}
    Ue, We = orc.jk_to_matrix(float(g['dt']), float(g['dx']), (float(g['UL']), float(g['WL'])), (z, z), J, K)
    for got, k in zip(Ue, ('AU', 'BU', 'CU', 'DU', 'GW')):
        assert np.array_equal(got, g[k]), k
    for got, k in zip(We, ('AW', 'BW', 'CW', 'DW', 'GU')):
        assert np.array_equal(got, g[k]), k
    U, W = orc.pair_solve(Ue, We)
    assert relerr(U, g['U']) == 0.0 and relerr(W, g['W']) == 0.0
    Ub, Wb = orc.pair_block_thomas(Ue, We)
    assert relerr(Ub, g['U']) < 1e-11 and relerr(Wb, g['W']) < 1e-11
    # the reference's manufactured-solution check (tests/test_JK_to_matrix_fd.py:69-76)
    assert np.allclose(U, g['Uan'], rtol=1e-4, atol=1e-4) and np.allclose(W, g['Wan'], rtol=1e-4, atol=1e-4)
    rU = orc.residual(Ue[0], Ue[1], Ue[2], Ue[3], g['res_U_in'], Ue[4], g['res_W_in'])
    rW = orc.residual(We[0], We[1], We[2], We[3], g['res_W_in'], We[4], g['res_U_in'])
    assert np.array_equal(rU, g['res_U']) and np.array_equal(rW, g['res_W'])


def test_g4_fluxsplit(golden):
    g = golden('g4_fluxsplit')
    for tag in ('it0', 'it20'):
        flux = orc.shestakov_flux(g[f'{tag}_p'], float(g['shest_dx']))
        assert np.array_equal(flux, g[f'{tag}_flux'])
        out = orc.flux_split(flux, g[f'{tag}_p'], float(g['shest_dx']), float(g['shest_Dmin']),
                             float(g['shest_Dmax']), float(g['shest_thr']))
        for got, k in zip(out, ('D', 'c', 'DHat', 'cHat', 'theta')):
            assert np.array_equal(got, g[f'{tag}_{k}']), (tag, k)
    for tag, kw in (('syn', {}), ('geo', dict(gxx=g['geo_gxx'], gradx=g['geo_gradx']))):
        out = orc.flux_split(g['syn_flux'], g['syn_p'], float(g['syn_dx']), float(g['syn_Dmin']),
                             float(g['syn_Dmax']), float(g['syn_thr']), **kw)
        for got, k in zip(out, ('D', 'c', 'DHat', 'cHat', 'theta')):
            assert np.array_equal(got, g[f'{tag}_{k}'], equal_nan=True), (tag, k)
    th = g['syn_theta']
    assert (th == 0).any() and (th == 0.5).any() and (th == 1).any() and ((th > 0.5) & (th < 1)).any()


def test_g567_small(golden):
    g = golden('g567_small')
    prev = None
    for y, exp in zip(g['ewma_in'], g['ewma_out']):
        prev = orc.ewma_next(y, prev, float(g['ewma_param']))
        assert prev == exp
    assert abs(prev - 1.398528) < 1e-13                     # tests/test_lodestro.py:17-26
    assert abs(orc.ewma_next(37, 7.1, 0.12) - 10.688) < 1e-13  # tests/test_lodestro.py:7-15
    for w in (1, 3, 4, 5, 11):
        assert np.array_equal(orc.moving_average(g['sm_x'], orc.odd_window(w)), g[f'sm_w{w}'])
    exp9 = np.array([1., 4., 5.2, 5.6, 5.4, 5., 5.2, 13. / 3, 5])   # tests/test_smoother.py:5-10
    assert np.allclose(orc.moving_average(g['sm9_x'], 5), exp9, rtol=1e-14)
    assert np.array_equal(orc.ddx_edge1(g['d_u'], float(g['d_dx'])), g['d_edge1'])
    assert np.array_equal(orc.ddx_edge2(g['d_u'], float(g['d_dx'])), g['d_edge2'])
    assert np.allclose(g['d_edge1'], [5 / 2, 25 / 4, 5 / 2, -25 / 2, -15 / 4, 55 / 4, 15], rtol=0, atol=1e-13)


@pytest.mark.parametrize('N', [256, 500, 512])
@pytest.mark.parametrize('tag,ap', [('p1', 1.0), ('p03', 0.3)])
def test_g8_onefield_trajectory(golden, N, tag, ap):
    g = golden('g8_onefield')
    key = f'N{N}_{tag}'
    dx = 1.0 / (N - 1)
    x = np.arange(N) * dx
    H1 = np.ones(N)
    H7 = orc.shestakov_source(x, 1.0, 0.1)
    st = orc.State1F(1 - 0.5 * x)
    errs = []
    for it in range(int(g[f'{key}_iters'])):
        rms, bad = orc.picard_step_1f(st, 1e4, dx, 1e-2, H1, H7, 0.3, ap)
        errs.append(float(rms))
        assert not bad
        if it in (0, 1, 2, 10):
            assert relerr(st.u, g[f'{key}_it{it}_u']) < TIGHT
            assert relerr(st.DE, g[f'{key}_it{it}_DE']) < TIGHT
            assert relerr(st.cE, g[f'{key}_it{it}_cE']) < 1e-12
            assert relerr(st.uE, g[f'{key}_it{it}_uE']) < TIGHT
    assert relerr(st.u, g[f'{key}_final']) < 1e-12
    assert np.allclose(errs, g[f'{key}_err'], rtol=1e-6, atol=1e-16)
    assert errs[-1] < 1e-11 and errs[-2] >= 1e-11      # stops on the same iteration


def test_g8_fine_grid(golden):
    g = golden('g8_onefield')
    N = 8192
    dx = 1.0 / (N - 1)
    x = np.arange(N) * dx
    st = orc.State1F(1 - 0.5 * x)
    H7 = orc.shestakov_source(x, 1.0, 0.1)
    for it in range(11):
        orc.picard_step_1f(st, 1e4, dx, 1e-2, np.ones(N), H7, 0.3, 0.3)
        if it in (0, 10):
            assert relerr(st.u, g[f'N8192_it{it}_u']) < 1e-12


def test_g10_ensemble(golden):
    g = golden('g10_ensemble32')
    ens = orc.shestakov_ensemble(32, 512)
    for k in ('nL', 'S0', 'delta'):
        assert np.array_equal(ens[k], g[k])
    sel = [0, 5, 17, 31]
    H7 = orc.shestakov_source(ens['x'], ens['S0'][sel], ens['delta'][sel])
    out, iters, err = orc.run_1f(ens['u0'][sel], 1e4, ens['dx'], ens['nL'][sel], np.ones(512), H7,
                                 0.3, 1.0, 1e-11, 1000 // 10)
    assert np.array_equal(iters, g['iters'][sel])
    assert relerr(out, g['finals'][sel]) < 1e-12
    assert np.allclose(err, g['errs'][sel], rtol=1e-6, atol=1e-16)


def test_g9_threefield(golden):
    g = golden('g9_threefield')
    N = 500
    dx = 1.0 / (N - 1)
    x = np.arange(N) * dx
    ic = 1 - 0.5 * x
    (n, pi, pe), it, err = orc.run_3f(ic, ic, ic, 1e6, dx, (2, 0.1, 0.3), 2.2, x, 0.3, 0.3, 1e-9, 1000)
    assert it == int(g['N500_iters']) == 65
    assert max(relerr(n, g['N500_n']), relerr(pi, g['N500_pi']), relerr(pe, g['N500_pe'])) < 1e-11
    assert np.allclose(err[:it], g['N500_err'], rtol=1e-6, atol=1e-16)
    # values quoted for this problem: n[0]=3.943314, pi[0]=0.325379, pe[0]=0.316446
    assert abs(n[0] - 3.943314) < 1e-6 and abs(pi[0] - 0.325379) < 1e-6 and abs(pe[0] - 0.316446) < 1e-6
    # block-Thomas instead of LAPACK for both solves: same trajectory within the parity bar
    (n2, pi2, pe2), it2, _ = orc.run_3f(ic, ic, ic, 1e6, dx, (2, 0.1, 0.3), 2.2, x, 0.3, 0.3, 1e-9, 1000,
                                        tri=orc.thomas, pair=orc.pair_block_thomas)
    assert it2 == it
    assert max(relerr(n2, n), relerr(pi2, pi), relerr(pe2, pe)) < 1e-10


def test_g9_threefield_inner(golden):
    g = golden('g9_threefield')
    N = 500
    dx = 1.0 / (N - 1)
    x = np.arange(N) * dx
    ic = 1 - 0.5 * x
    (n, pi, pe), it, err = orc.run_3f(ic, ic, ic, 1e6, dx, (2, 0.1, 0.3), 2.2, x, 0.3, 0.3, 1e-9, 1000, inner=2)
    assert it == int(g['inner_iters']) == 64
    assert max(relerr(n, g['inner_n']), relerr(pi, g['inner_pi']), relerr(pe, g['inner_pe'])) < 1e-11


def test_g11_residual(golden):
    g = golden('g11_residual')
    r = orc.residual(g['A'], g['B'], g['C'], g['f'], g['u'])
    assert np.array_equal(r, g['r'])
    assert abs(orc.rms_of([r]) - float(g['rms'])) < 1e-15 * float(g['rms']) * 4


def test_steady_state_integration():
    """The reference's own integration check (tests/solver_integration_test.py:14-34): the converged
    one-field profile is within rms 1e-3 of the analytic steady state."""
    N = 500
    dx = 1.0 / (N - 1)
    x = np.arange(N) * dx
    out, iters, _ = orc.run_1f(1 - 0.5 * x, 1e4, dx, 1e-2, np.ones(N), orc.shestakov_source(x), 0.3, 1.0,
                               1e-11, 1000)
    nss = orc.shestakov_steady_state(x, 1e-2)
    r = (out[0] - nss) / np.max(np.abs(nss))
    assert np.sqrt(np.mean(r ** 2)) < 1e-3


def test_g12_pseudo_timestep_control(golden):
    """tread_lightly / tread_initially (tango/solver.py:198-235,311-372) on the examples' problem."""
    g = golden('g12_tread')
    N = int(g['N']); dx = 1 / (N - 1); x = np.arange(N) * dx
    H7 = orc.shestakov_source(x, 1, 0.1)
    cases = {'lightly': (dict(maxFractionalChange=0.2), None),
             'lightly_tight': (dict(maxFractionalChange=0.05, innerFraction=0.8), None),
             'initially': (None, dict(numInitialSteps=6, dtau=0.3)),
             'both': (dict(maxFractionalChange=0.1), dict(numInitialSteps=4, dtau=0.5))}
    for tag, (tl, ti) in cases.items():
        u, it, err, hist, halv = orc.run_1f_tread(g['n0'], 1e4, dx, 1.0, np.ones(N), H7, 0.1, 1.0, 1e-10, 1000, tl, ti)
        assert it == int(g[f'{tag}_iters'])
        assert relerr(u, g[f'{tag}_final']) < 1e-13
        assert relerr(hist[:40], g[f'{tag}_first40']) < 1e-13
        assert np.allclose(err, g[f'{tag}_err'], rtol=1e-9, atol=1e-15)
        assert (halv.sum() > 0) == (tl is not None)


def test_g13_timesteps(golden):
    """Several timesteps of unequal length (tango/solver.py:117-186, :533-548)."""
    g = golden('g13_timesteps')
    N = int(g['N']); dx = 1 / (N - 1); x = np.arange(N) * dx
    out = orc.run_1f_timesteps(g['n0'], g['tArray'], dx, 1e-2, np.ones(N), orc.shestakov_source(x, 1.0, 0.1), 0.3, 1.0, 1e-10, 400)
    assert [o[1] for o in out] == g['iters'].tolist()
    for m, (u, it, err) in enumerate(out):
        assert relerr(u, g['profiles'][m]) < 1e-13
        assert np.allclose(err, g['errs'][m, :it], rtol=1e-9, atol=1e-15)


def test_g16_grid_mapping(golden):
    """Spline resampling and zero extension (tango/interfacegrids_gene.py:35-87, :446-484)."""
    g = golden('g16_gridmap')
    assert relerr(orc.spline_resample(g['xT'], g['prof'], g['xG']), g['prof_on_G']) < 1e-13
    assert relerr(orc.spline_resample(g['xn'], g['yn'], g['xq']), g['yq']) < 1e-12
    assert relerr(orc.extend_with_zeros_both_sides(g['xG'], g['Dg'], g['xT'], True), g['D_on_T']) < 1e-13
    assert relerr(orc.extend_with_zeros_both_sides(g['xG'], g['cg'], g['xT']), g['c_on_T']) < 1e-13
    assert np.all(g['D_on_T'] >= 0) and g['D_on_T'][0] == 0 and g['D_on_T'][-1] == 0


def test_g17_grid_mapping_more(golden):
    """Two-sided extrapolation of coefficients and the maps for a turbulence grid that reaches beyond the transport
    grid (tango/interfacegrids_gene.py:181-245, :300-443, :528-692)."""
    g = golden('g17_gridmap_more')
    assert relerr(orc.spline_resample(g['xT'], g['prof'], g['xG']), g['both_prof_on_G']) < 1e-13
    assert relerr(orc.extrap_on_left_and_right(g['xG'], g['Dg'], g['xT'], 0.2, 0.32, 0.55, 0.7, 2, True), g['both_D_on_T']) < 1e-12
    assert relerr(orc.extrap_on_left_and_right(g['xG'], g['cg'], g['xT'], 0.2, 0.32, 0.55, 0.7, 2), g['both_c_on_T']) < 1e-12
    assert relerr(orc.truncate_on_left_extrapolate_on_right(g['xTi'], g['profi'], g['xGi'], 10, True), g['in_prof_on_G']) < 1e-13
    assert relerr(orc.truncate_on_left_extrapolate_on_right(g['xTi'], g['steep'], g['xGi'], 10, True), g['in_steep_on_G']) < 1e-13
    assert np.any(g['in_steep_on_G'] == 0)          # the clamp is exercised
    assert relerr(orc.extend_with_zeros_left_side(g['xGi'], g['Dgi'], g['xTi'], True), g['in_D_on_T']) < 1e-13
    assert relerr(orc.extend_with_zeros_left_side(g['xGi'], g['cgi'], g['xTi']), g['in_c_on_T']) < 1e-13
    assert relerr(orc.truncate_on_left_fixed_slope_on_right(g['xTi'], g['profi'], g['xGi'], -1.7, True), g['fix_prof_on_G']) < 1e-13
    assert relerr(orc.truncate_on_left_fixed_slope_on_right(g['xTi'], g['steep'], g['xGi'], -1.7, True), g['fix_steep_on_G']) < 1e-13
    assert np.array_equal(g['fix_D_on_T'], g['in_D_on_T']) and np.array_equal(g['fix_c_on_T'], g['in_c_on_T'])
    assert relerr(orc.constrained_regression_line(g['xTi'], g['profi'], np.linspace(-0.2, 0.1, 7), 'left', 6), g['reg_left']) < 1e-13
    assert relerr(orc.truncate_on_left_extrapolate_on_right(g['xTi'], g['profi'], np.linspace(-0.2, 0.5, 23), 6, False, 'left'),
                  g['ext_left']) < 1e-13


TREAD3_CASES = {'lightly': (dict(maxFractionalChange=0.2), None),
                'lightly_tight': (dict(maxFractionalChange=0.05, innerFraction=0.8), None),
                'initially': (None, dict(numInitialSteps=6, dtau=0.05)),
                'both': (dict(maxFractionalChange=0.1), dict(numInitialSteps=4, dtau=0.1))}


@pytest.mark.parametrize('tag', sorted(TREAD3_CASES))
def test_threefield_pseudo_timestep_control_g18(golden, tag):
    """The three-field restatement with tread-lightly / tread-initially against the reference's own runs."""
    g = golden('g18_threefield_tread')
    N = int(g['N']); dx = 1 / (N - 1); x = np.arange(N) * dx
    tl, ti = TREAD3_CASES[tag]
    (n, pi, pe), it, err, hist, halv = orc.run_3f_tread(g['ic'], g['ic'], g['ic'], 1e6, dx, tuple(g['bc']), float(g['nu0']), x, 0.3, 0.3,
                                                        1e-9, 400, tread_lightly=tl, tread_initially=ti)
    assert it == int(g[f'{tag}_iters'])
    if tl is not None:
        assert halv.max() >= 1           # the control acted
    assert np.allclose(err, g[f'{tag}_err'], rtol=1e-7, atol=1e-13)
    assert np.allclose(hist[:30], g[f'{tag}_first30'], rtol=1e-11, atol=0)
    for got, key in ((n, 'n'), (pi, 'pi'), (pe, 'pe')):
        assert np.allclose(got, g[f'{tag}_{key}'], rtol=1e-11, atol=0)
