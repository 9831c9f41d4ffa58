"""Failure paths of the reference on the GPU path: invalid profiles (tango/solver.py:420-439), singular systems
(scipy raises numpy.linalg.LinAlgError from tango/tridiag.py:96), argument errors."""
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


def one_field_solver(tango, x, nL, n0, maxit=200, tol=1e-9, ewma_iterates=None, **kw):
    dx = x[1] - x[0]
    sh = tango.extras.shestakov_nonlinear_diffusion
    lm = tango.lodestro_method.lm(0.3, 1.0, {'Dmin': 1e-5, 'Dmax': 1e13, 'dpdxThreshold': 10})
    if ewma_iterates is not None:
        lm.set_ewma_iterates(*ewma_iterates)
    f = tango.multifield.Field(label='n', rightBC=nL, profile_mminus1=n0, compute_all_H=sh.ComputeAllH(1.0, 0.1), lodestroMethod=lm)
    tango.multifield.check_fields_initialize([f])
    call = tango.multifield.ComputeAllHAllFields([f], tango.lodestro_method.TurbulenceHandler(dx, x, sh.AnalyticFluxModel(dx)))
    return tango.solver.Solver(1, x, np.array([0, 1e4]), maxit, tol, call, [f], **kw)


def test_negative_profile_stops_the_solver(tango):
    """A negative right boundary value makes the first new iterate invalid: solutionError, ok False, one iteration."""
    N = 128
    x = np.arange(N) * (1 / (N - 1))
    s = one_field_solver(tango, x, -0.05, 1 - 0.5 * x)
    assert s._fused is not None
    s.take_timestep()
    assert s.solutionError and not s.ok and not s.converged
    assert len(s.errHistoryFinal) == 1
    u, it, err = orc.run_1f(1 - 0.5 * x, 1e4, x[1] - x[0], -0.05, np.ones(N), orc.shestakov_source(x, 1.0, 0.1), 0.3, 1.0, 1e-9, 200)
    assert int(it[0]) == 1
    assert relerr(s.profiles['n'], u[0]) < TOL          # the offending iterate is what is left in profiles


def test_invalid_member_does_not_disturb_the_others(tango):
    N, nb = 256, 5
    x = np.arange(N) * (1 / (N - 1))
    nL = np.array([1e-2, 2e-2, -0.05, 1.5e-2, np.nan])
    ics = np.tile(1 - 0.5 * x, (nb, 1))
    s = one_field_solver(tango, x, nL, ics, iterationsPerCheck=4)
    assert s._fused is not None and s.batched
    s.take_timestep()
    assert s.solutionError and not s.converged
    st = s.profileStatus
    assert [bool(v & 2) for v in st] == [False, False, True, False, True]
    assert [bool(v & 1) for v in st] == [True, True, False, True, False]
    assert s.iterationsPerProfile[2] == 1 and s.iterationsPerProfile[4] == 1
    H7 = orc.shestakov_source(x, 1.0, 0.1)
    for b in (0, 1, 3):
        u, it, err = orc.run_1f(ics[b], 1e4, x[1] - x[0], nL[b], np.ones(N), H7, 0.3, 1.0, 1e-9, 200)
        assert abs(int(s.iterationsPerProfile[b]) - int(it[0])) <= 1
        if int(s.iterationsPerProfile[b]) == int(it[0]):
            assert relerr(s.profiles['n'][b], u[0]) < TOL


def test_singular_system_raises_linalgerror(tango):
    n = 16
    A = np.zeros(n); B = np.ones(n); C = np.zeros(n); D = np.ones(n)
    B[5] = 0.0                                  # an all-zero row
    with pytest.raises(np.linalg.LinAlgError):
        tango.tridiag.solve(A, B, C, D)
    ok = tango.tridiag.solve(A, np.ones(n), C, D)
    assert np.array_equal(ok, D)


def test_argument_errors(tango):
    z = np.zeros(8)
    with pytest.raises(AssertionError):
        tango.HToMatrixFD.H_to_matrix(1.0, 0.1, 1.0, z, np.ones(8), H2=np.ones(7))
    with pytest.raises(ValueError):
        tango.multifield.Field(rightBC=1.0, profile_mminus1=z, compute_all_H=lambda *a: None)      # no label
    from tango_amd import _hip as H
    import ctypes as C
    S, Q = H.State1F(), H.Params1F()
    S.nb, S.n = 1, 8
    assert H.lib.tg_picard_1f_step(C.byref(S), C.byref(Q), None) != 0           # NULL arrays are refused, not dereferenced
    assert b'' != H.lib.tg_last_error()
    Q2 = H.Params1F(); Q2.tread_lightly = 1
    assert H.lib.tg_picard_1f_step(C.byref(S), C.byref(Q2), None) != 0


def test_restart_from_ewma_iterates(tango):
    """Continue a solve in a new Solver from the profile and the EWMA iterates of an interrupted one
    (tango/lodestro_method.py:499-529 set_ewma_iterates): same trajectory as the uninterrupted run."""
    N = 256
    x = np.arange(N) * (1 / (N - 1))
    a = one_field_solver(tango, x, 1e-2, 1 - 0.5 * x, maxit=30, tol=1e-30)
    a.take_timestep()
    part = one_field_solver(tango, x, 1e-2, 1 - 0.5 * x, maxit=12, tol=1e-30)
    part.take_timestep()
    lm_old = part.fields[0].lodestroMethod
    its = (lm_old._EWMAProfile._yEWMA_lminus1, lm_old._EWMATurbD._yEWMA_lminus1, lm_old._EWMATurbc._yEWMA_lminus1)
    cont = one_field_solver(tango, x, 1e-2, 1 - 0.5 * x, maxit=18, tol=1e-30, ewma_iterates=its, profiles={'n': part.profiles['n']})
    assert cont._fused is not None
    cont.take_timestep()
    assert relerr(cont.profiles['n'], a.profiles['n']) < 1e-12
    assert np.allclose(cont.errHistoryFinal, a.errHistoryFinal[12:30], rtol=1e-9, atol=1e-15)


def test_restart_file_round_trip(tango, tmp_path, monkeypatch):
    """tango/restart.py:59-98: interrupt a timestep after 12 iterations, write the state, continue in a new Solver
    from the file with startIterationNumber; the result equals 30 iterations in one go."""
    monkeypatch.chdir(tmp_path)
    N = 256
    x = np.arange(N) * (1 / (N - 1))
    n0 = 1 - 0.5 * x
    a = one_field_solver(tango, x, 1e-2, n0, maxit=30, tol=1e-30)
    a.take_timestep()
    part = one_field_solver(tango, x, 1e-2, n0, maxit=12, tol=1e-30)
    part.take_timestep()
    assert tango.restart.check_if_should_restart() is None
    fn = tango.restart.write_state(part, basename='run', setNumber=0)
    assert tango.restart.check_if_should_restart('run') == fn == tango.restart.check_if_should_restart()
    setNumber, start, t, timestep, profs, pE, DE, cE = tango.restart.read_metadata_from_previousfile(fn)
    assert (setNumber, start, timestep) == (1, 12, 1) and t == 1e4
    um1 = tango.restart.read_profiles_mminus1(fn)['n']
    assert np.array_equal(um1, n0)
    cont = one_field_solver(tango, x, 1e-2, um1, maxit=30, tol=1e-30, ewma_iterates=(pE['n'], DE['n'], cE['n']),
                            profiles=profs, startIterationNumber=start)
    assert cont._fused is not None
    cont.take_timestep()
    assert cont.countLastTimestep == 30 and len(cont.errHistoryFinal) == 18
    assert relerr(cont.profiles['n'], a.profiles['n']) < 1e-12
    assert np.allclose(cont.errHistoryFinal, a.errHistoryFinal[12:30], rtol=1e-9, atol=1e-15)
    fn2 = tango.restart.write_state(cont, basename='run', setNumber=setNumber)
    assert tango.restart.find_latest_savefile('run') == fn2 == 'run_s1.npz'
    assert tango.restart.read_metadata_from_previousfile(fn2)[1] == 30


def test_static_callback_found_by_trial_is_checked_again(tango):
    """A callback that looks independent of profiles and time in the two trial calls but is not (here: it counts its
    calls) is caught at a convergence check instead of silently running with frozen coefficients."""
    N = 128
    x = np.arange(N) * (1 / (N - 1))
    sh = tango.extras.shestakov_nonlinear_diffusion
    calls = [0]

    class PowerFlux(object):
        def __init__(self, dx):
            self.dx = dx

        def get_flux(self, profiles):
            return {'n': np.asarray(sh.get_flux(profiles['n'], self.dx))}

    def cb(t, xx, profiles, HT):
        calls[0] += 1
        amp = 1.0 if calls[0] <= 4 else 2.0           # the source doubles after the fourth call
        return tango.multifield.HCoefficients(H1=np.ones_like(xx), H7=amp * sh.GetSource(xx)) + HT
    lm = tango.lodestro_method.lm(0.3, 1.0, {'Dmin': 1e-5, 'Dmax': 1e13, 'dpdxThreshold': 10})
    f = tango.multifield.Field(label='n', rightBC=1e-2, profile_mminus1=1 - 0.5 * x, compute_all_H=cb, lodestroMethod=lm)
    tango.multifield.check_fields_initialize([f])
    call = tango.multifield.ComputeAllHAllFields([f], tango.lodestro_method.TurbulenceHandler(x[1] - x[0], x, PowerFlux(x[1] - x[0])))
    s = tango.solver.Solver(1, x, np.array([0, 1e4]), 50, 1e-30, call, [f], iterationsPerCheck=5)
    assert type(s._fused).__name__ == '_FusedExternal1F' and s._static_by_trial
    with pytest.raises(RuntimeError, match='detectStaticH'):
        s.take_timestep()
