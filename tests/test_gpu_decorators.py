"""Flux-model decorators and the generalised analytic flux on the device: the Tukey taper of tango/extras/bufferzone.py,
the AR(1) multiplicative noise of tango/extras/noisyflux.py and the |n'/n|^p flux of
examples/stripped_down_tango_example.py, as parts of the fused one-field step."""

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


def test_tukey_window_on_the_device(tango, golden):
    from tango_amd.extras import bufferzone
    g = golden('g15_decorators')
    for n, a, key in ((200, 0.3, 'tukey_200_0p3'), (51, 0.5, 'tukey_51_0p5'), (40, 1.0, 'tukey_40_1')):
        assert np.allclose(bufferzone.tukey_window_device(n, a).to_host(), g[key], rtol=0, atol=2e-15), key
    assert np.array_equal(bufferzone.tukey_window_device(17, 0.0).to_host(), np.ones(17))


def _noise(tango, nb, n, seed=3, seq=0, amp=0.1, tac=7.5, damp=None, taper=None):
    from tango_amd import _hip as H
    out = H.DeviceArray((nb, n))
    damp = int(0.03 * n) if damp is None else damp
    H.check(H.lib.tg_ar1_noise(seed, seq, amp, tac, damp, H.ptr_of(taper), out.ptr, nb, n, H.stream()))
    return out.to_host()


def test_ar1_noise_statistics(tango):
    """Device-generated noise: what tango/extras/noisyflux.py:106-162 specifies, checked as statistics (the reference's
    numbers come from NumPy's Mersenne Twister and cannot be reproduced on a GPU: parity of the numbers is unpinned)."""
    nb, n, amp, tac = 4096, 1300, 0.1, 7.5
    lam = (tac - 1) / (tac + 1)
    m = _noise(tango, nb, n, amp=amp, tac=tac)
    d = int(0.03 * n)
    ramp = np.ones(n); ramp[:d] = np.linspace(0, 1, d); ramp[-d:] = np.linspace(1, 0, d)
    assert np.all(m[:, 0] == 1.0) and np.all(m[:, -1] == 1.0)                 # dampen_sides: no noise at the two ends
    y = (m - 1.0)[:, d:-d] / amp                                              # the AR(1) series itself, unit variance
    cnt = y.size
    assert abs(y.mean()) < 5 / np.sqrt(cnt / (2 * tac))                       # mean 0 (samples correlated over ~tac points)
    assert abs(y.var() - 1.0) < 0.02
    assert abs(np.mean(y[:, 1:] * y[:, :-1]) - lam) < 0.01                    # lag-one correlation
    assert abs(np.mean(y[:, 5:] * y[:, :-5]) - lam ** 5) < 0.01
    # the 512-point passes join seamlessly (first pass boundary at point 512, second at 1024)
    full = (m - 1.0) / amp
    for j in (512, 1024):
        assert abs(np.mean(full[:, j] * full[:, j - 1]) - lam) < 0.05
        assert abs(np.var(full[:, j]) - 1.0) < 0.08
    assert abs(np.var(full[:, d]) - 1.0) < 0.08                               # stationary from the first undamped point on
    # the damped end zones scale the same series linearly
    assert abs(np.var((m - 1.0)[:, d // 2]) / amp ** 2 - ramp[d // 2] ** 2) < 0.05
    # profiles are independent, sequence numbers give fresh draws, equal arguments give equal numbers on any launch shape
    assert abs(np.mean(y[0::2] * y[1::2])) < 0.01
    assert np.array_equal(_noise(tango, nb, n, amp=amp, tac=tac), m)
    assert np.array_equal(_noise(tango, 5, n, amp=amp, tac=tac), m[:5])
    m2 = _noise(tango, 64, n, seq=1, amp=amp, tac=tac)
    assert abs(np.mean((m2 - 1) * (m[:64] - 1)) / amp ** 2) < 0.02
    # a taper multiplies in
    from tango_amd.extras import bufferzone
    w = bufferzone.tukey_window_device(n, 0.3)
    assert np.allclose(_noise(tango, 8, n, amp=amp, tac=tac, taper=w), m[:8] * w.to_host(), rtol=1e-15)
    with pytest.raises(ValueError):
        _noise(tango, 4, 64, tac=1.0)
    with pytest.raises(ValueError):
        _noise(tango, 4, 64, damp=32)


def _oracle_run(ens, b, its, flux_fns, a_prof=0.3, dt=1e4):
    N = len(ens['x'])
    H7 = orc.shestakov_source(ens['x'], ens['S0'], ens['delta'])
    st = orc.State1F(ens['u0'][b])
    for it in range(its):
        orc.picard_step_1f(st, dt, ens['dx'], ens['nL'][b], np.ones(N), H7[b], 0.3, a_prof, flux_fn=flux_fns(it))
    return st


@pytest.mark.parametrize('N,nb,p,k', [(512, 4, 2.0, 1), (512, 3, 1.5, 4), (256, 5, 3.0, 1), (500, 2, 2.5, 1), (1024, 2, 2.0, 1)])
def test_power_law_flux_built_in(tango, N, nb, p, k):
    from tango_amd import _hip as H
    from tango_amd.ensemble import OneFieldEnsemble
    its = 8
    ens = orc.shestakov_ensemble(nb, N)
    E = OneFieldEnsemble(ens, a_prof=0.3, max_iter=its, fixed_count=True)
    E.params.flux_model, E.params.flux_power = H.FLUX_POWER, p
    for _ in range(its // k):
        E.step(iters=k)
    dx = ens['dx']
    flux = lambda n: -np.abs((orc.ddx_edge1(n, dx) / n)) ** p * orc.ddx_edge1(n, dx)
    for b in range(nb):
        st = _oracle_run(ens, b, its, lambda it: flux)
        assert relerr(E.u.to_host()[b], st.u) < TOL and relerr(E.DE.to_host()[b], st.DE) < TOL and relerr(E.cE.to_host()[b], st.cE) < TOL


def _solver(tango, ens, model, maxit, smoother=None, **kw):
    sh = tango.extras.shestakov_nonlinear_diffusion
    th = {'Dmin': 1e-5, 'Dmax': 1e13, 'dpdxThreshold': 10}
    x = ens['x']
    turb = tango.lodestro_method.TurbulenceHandler(ens['dx'], x, model, fluxSmoother=smoother)
    lm = tango.lodestro_method.lm(0.3, 0.3, th)
    f = tango.multifield.Field(label='n', rightBC=ens['nL'], profile_mminus1=ens['u0'], compute_all_H=sh.ComputeAllH(ens['S0'], ens['delta']),
                               lodestroMethod=lm)
    tango.multifield.check_fields_initialize([f])
    return tango.solver.Solver(1, x, np.array([0, 1e4]), maxit, 1e-30, tango.multifield.ComputeAllHAllFields([f], turb), [f], **kw)


@pytest.mark.parametrize('N', [512, 300])
def test_buffer_zone_on_the_fused_path(tango, N):
    """BufferZone around the built-in flux model: the taper is the kernel's flux multiplier, several iterations per launch,
    with and without the fused smoother behind it; against the oracle fed the tapered flux."""
    from tango_amd.extras import bufferzone
    sh = tango.extras.shestakov_nonlinear_diffusion
    nb, its = 3, 9
    ens = orc.shestakov_ensemble(nb, N)
    w = bufferzone.tukey_window(N, 2 * 0.1)
    for window in (1, 5):
        sm = tango.smoother.Smoother(window) if window > 1 else None
        s = _solver(tango, ens, bufferzone.BufferZone(sh.AnalyticFluxModel(ens['dx']), 0.1), its, smoother=sm, iterationsPerCheck=3)
        assert type(s._fused).__name__ == '_Fused1F' and s._fused.taper is not None
        s.take_timestep()
        H7 = orc.shestakov_source(ens['x'], ens['S0'], ens['delta'])
        for b in range(nb):
            st = orc.State1F(ens['u0'][b])
            for _ in range(its):
                orc.picard_step_1f(st, 1e4, ens['dx'], ens['nL'][b], np.ones(N), H7[b], 0.3, 0.3, smooth_window=window,
                                   flux_fn=lambda n: w * orc.shestakov_flux(n, ens['dx']))
            assert relerr(s.profiles['n'][b], st.u) < TOL, (window, b)


def test_noisy_flux_on_the_fused_path(tango):
    """NoisyFlux with a device seed around BufferZone around the power-law model: everything stays on the device, a fresh
    multiplier per iteration; the oracle is fed the same multipliers (regenerated from seed and sequence number)."""
    from tango_amd.extras import bufferzone, noisyflux
    sh = tango.extras.shestakov_nonlinear_diffusion
    N, nb, its, amp, tac_x = 512, 4, 6, 0.05, 0.02
    ens = orc.shestakov_ensemble(nb, N)
    dx = ens['dx']
    model = noisyflux.NoisyFlux(bufferzone.BufferZone(sh.PowerLawFluxModel(dx, p=2), 0.05), amp, tac_x, dx, device_seed=11)
    s = _solver(tango, ens, model, its, iterationsPerCheck=4)
    assert type(s._fused).__name__ == '_Fused1F' and s._fused.noise is model
    s.take_timestep()
    assert model.calls == its
    w = bufferzone.tukey_window_device(N, 0.1)
    mult = [_noise(tango, nb, N, seed=11, seq=it, amp=amp, tac=tac_x / dx, taper=w) for it in range(its)]
    assert all(np.std(m[:, N // 2]) > 0.2 * amp for m in mult)
    flux = lambda n: -np.abs(orc.ddx_edge1(n, dx) / n) ** 2 * orc.ddx_edge1(n, dx)
    for b in range(nb):
        st = _oracle_run(ens, b, its, lambda it: (lambda n: mult[it][b] * flux(n)))
        assert relerr(s.profiles['n'][b], st.u) < TOL, b
    # without a device seed the noise is the reference's host-side draw: not this path
    host = noisyflux.NoisyFlux(sh.AnalyticFluxModel(dx), amp, tac_x, dx)
    s2 = _solver(tango, ens, host, 3)
    assert type(s2._fused).__name__ != '_Fused1F'
