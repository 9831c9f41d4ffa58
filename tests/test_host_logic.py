"""Host-side logic that needs no GPU: data-model validation, grouping, ensemble generators, sharding.
The expectations follow the reference's own tests (tests/test_multifield.py, tests/test_fieldgroups.py)."""
import numpy as np
import pytest

import tango_amd as tango
from tango_amd import multifield, fieldgroups
from tango_amd.ensemble import shard_range, shestakov_ensemble
from oracle import tango_oracle as orc


def _fields():
    p = np.ones(6)
    f0 = multifield.Field(label='a', rightBC=1.0, profile_mminus1=p, compute_all_H=lambda *a: None)
    f1 = multifield.Field(label='b', rightBC=2.0, profile_mminus1=p, compute_all_H=lambda *a: None, coupledTo='c')
    f2 = multifield.Field(label='c', rightBC=3.0, profile_mminus1=p, compute_all_H=lambda *a: None, coupledTo='b')
    return [f0, f1, f2]


def test_field_requires_label():
    with pytest.raises(ValueError):
        multifield.Field()


def test_check_fields_initialize():
    multifield.check_fields_initialize(_fields())
    fs = _fields(); fs[1].label = 'a'
    with pytest.raises(ValueError):
        multifield.check_fields_initialize(fs)
    fs = _fields(); fs[0].rightBC = np.ones(3)
    with pytest.raises(ValueError):
        multifield.check_fields_initialize(fs)
    fs = _fields(); fs[2].profile_mminus1 = np.ones(7)
    with pytest.raises(ValueError):
        multifield.check_fields_initialize(fs)
    fs = _fields(); fs[2].coupledTo = 'a'
    with pytest.raises(ValueError):
        multifield.check_fields_initialize(fs)
    fs = _fields(); fs[1].coupledTo = 'b'
    with pytest.raises(ValueError):
        multifield.check_fields_initialize(fs)
    fs = _fields(); fs[0].compute_all_H = 3
    with pytest.raises(ValueError):
        multifield.check_fields_initialize(fs)
    # ensembles: one boundary value per profile is accepted
    f = multifield.Field(label='n', rightBC=np.ones(4), profile_mminus1=np.ones((4, 6)), compute_all_H=lambda *a: None)
    multifield.check_fields_initialize([f])


def test_get_field_by_label():
    fs = _fields()
    assert multifield.get_field_by_label(fs, 'c') is fs[2]
    with pytest.raises(ValueError):
        multifield.get_field_by_label(fs, 'zz')


def test_hcoefficients_add():
    a = multifield.HCoefficients(H1=np.ones(3), H2=np.full(3, 2.0))
    b = multifield.HCoefficients(H2=np.full(3, 5.0), H7=np.full(3, 7.0))
    c = a + b
    assert np.array_equal(c.H1, np.ones(3)) and np.array_equal(c.H2, np.full(3, 7.0)) and np.array_equal(c.H7, np.full(3, 7.0))
    assert c.H3 is None and c.H8 is None
    assert 'H2=' in repr(c) and 'H3' not in repr(c)


def test_fields_to_fieldgroups_order_and_contents():
    fs = _fields()
    Hs = {f.label: multifield.HCoefficients(H1=np.ones(6), H8=np.full(6, float(i))) for i, f in enumerate(fs)}
    groups = fieldgroups.fields_to_fieldgroups(fs, Hs)
    assert len(groups) == 2
    assert isinstance(groups[0], fieldgroups.UncoupledFieldGroup) and groups[0].label == 'a'
    assert groups[0].HCoeffs is Hs['a'] and groups[0].rightBC == 1.0
    g = groups[1]
    assert (g.label0, g.label1) == ('b', 'c') and g.rightBC == (2.0, 3.0)
    assert np.array_equal(g.HCoeffs.J8, Hs['b'].H8) and np.array_equal(g.HCoeffs.K8, Hs['c'].H8)
    # pair listed partner-first
    groups = fieldgroups.fields_to_fieldgroups([fs[2], fs[0], fs[1]], Hs)
    assert (groups[0].label0, groups[0].label1) == ('c', 'b') and groups[1].label == 'a'


def test_fieldgroups_to_profiles():
    g0 = fieldgroups.UncoupledFieldGroup('f0'); g0.profileSolution = {'f0': np.ones(4)}
    g1 = fieldgroups.PairCoupledFieldGroup('f1', 'f2'); g1.profileSolution = {'f1': 2 * np.ones(4), 'f2': 3 * np.ones(4)}
    p = fieldgroups.fieldgroups_to_profiles([g0, g1])
    assert sorted(p) == ['f0', 'f1', 'f2'] and p['f2'][0] == 3


def test_is_unacceptable_one_profile():
    from tango_amd.solver import is_unacceptable_one_profile as bad
    old = np.linspace(1, 2, 10)
    assert not bad(old, old * 1.1)
    assert bad(old, old * 1.2)                      # >= 20 % counts
    new = old.copy(); new[7] *= 3                   # outside the inner half: ignored
    assert not bad(old, new)
    new = old.copy(); new[2] *= 0.5
    assert bad(old, new)


def test_smoother_window_rule():
    assert tango.smoother.add_one_if_even(4) == 5 and tango.smoother.add_one_if_even(5) == 5
    with pytest.raises(ValueError):
        tango.smoother.add_one_if_even(2.5)
    x = np.arange(5.0)
    assert tango.smoother.moving_average(x, 1) is x


def test_shard_range_covers_everything():
    for nb in (1, 7, 64, 65536, 1048576):
        for world in (1, 2, 3, 8):
            ranges = [shard_range(nb, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == nb
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in ranges]
            assert max(sizes) - min(sizes) <= 1


def test_ensemble_generator_matches_oracle_and_shards():
    full = shestakov_ensemble(64, 128)
    ref = orc.shestakov_ensemble(64, 128)
    for k in ('u0', 'nL', 'S0', 'delta', 'x'):
        assert np.array_equal(full[k], ref[k])
    lo, hi = shard_range(64, 1, 3)
    part = shestakov_ensemble(64, 128, lo=lo, hi=hi)
    assert np.array_equal(part['u0'], full['u0'][lo:hi]) and np.array_equal(part['nL'], full['nL'][lo:hi])
    assert full['nL'][0] == 1e-2 and np.array_equal(full['u0'][0], 1 - 0.5 * full['x'])


def test_solver_constructor_and_ok_without_gpu():
    """Constructing a Solver touches no kernel on the generic path; `ok` follows tango/solver.py:404-418."""
    x = np.linspace(0, 1, 16)
    f = multifield.Field(label='n', rightBC=0.1, profile_mminus1=np.ones(16), compute_all_H=lambda *a: None)
    s = tango.solver.Solver(1, x, np.array([0, 1.0]), 10, 1e-9, lambda *a, **k: None, [f])
    assert s.ok and s.N == 16 and s._fused is None
    s.solutionError = True
    assert not s.ok
    s.solutionError = False; s.iterationNumber = 10
    assert not s.ok
    assert tango.solver.Solver._get_tdt(np.array([0, 2.5, 10]), 2) == (10, 7.5)


def test_restart_file_layout(tmp_path, monkeypatch):
    """tango_amd.restart without a GPU: file naming, latest-file search and the reference's return tuple."""
    import types
    import tango_amd
    monkeypatch.chdir(tmp_path)
    ew = lambda v: types.SimpleNamespace(_yEWMA_lminus1=v)
    lm = types.SimpleNamespace(_EWMAProfile=ew(np.full(4, 2.0)), _EWMATurbD=ew(np.full(4, 3.0)), _EWMATurbc=ew(np.full(4, 4.0)))
    f = types.SimpleNamespace(label='n', lodestroMethod=lm)
    s = types.SimpleNamespace(t=0.5, m=2, iterationNumber=0, countLastTimestep=7, fields=[f], profiles={'n': np.arange(4.0)},
                              restart_profile_mminus1={'n': np.ones(4)})
    assert tango_amd.restart.check_if_should_restart() is None
    fn = tango_amd.restart.write_state(s, 'abc', 3)
    assert fn == 'abc_s3.npz'
    tango_amd.restart.write_state(s, 'abc', 11)
    assert tango_amd.restart.find_latest_savefile('abc') == 'abc_s11.npz' == tango_amd.restart.check_if_should_restart()
    out = tango_amd.restart.read_metadata_from_previousfile(fn)
    assert out[:4] == (4, 7, 0.5, 2)
    assert np.array_equal(out[4]['n'], np.arange(4.0)) and np.array_equal(out[5]['n'], np.full(4, 2.0))
    assert np.array_equal(out[6]['n'], np.full(4, 3.0)) and np.array_equal(out[7]['n'], np.full(4, 4.0))
    assert np.array_equal(tango_amd.restart.read_profiles_mminus1(fn)['n'], np.ones(4))
    tango_amd.restart.write_state(s, 'other', 0)
    with pytest.raises(ValueError):
        tango_amd.restart.check_if_should_restart()


def test_flux_decorators_match_reference():
    """extras.noisyflux / extras.bufferzone against the reference's output with the same seed (golden g15)."""
    from conftest import load_golden
    from tango_amd.extras import bufferzone, noisyflux
    g = load_golden('g15_decorators')
    np.random.seed(11)
    assert np.allclose(noisyflux.ar1noise(300, 7.5), g['ar1_300_7p5'], rtol=1e-12, atol=1e-14)
    assert np.allclose(noisyflux.ar1noise(64, 3.0), g['ar1_64_3'], rtol=1e-12, atol=1e-14)
    v = g['v']
    assert np.array_equal(noisyflux.dampen_sides(v), g['dampen_default'])
    assert np.array_equal(noisyflux.dampen_sides(v, 10), g['dampen_10'])
    np.random.seed(12)
    assert np.allclose(noisyflux.NoisyFlux._add_noise(v, 0.1, 7.5), g['noisy'], rtol=1e-12)
    assert np.allclose(noisyflux.UniformNoisyFlux._add_noise(v, 0.2), g['uniform'], rtol=1e-12)
    for n, a, key in ((200, 0.3, 'tukey_200_0p3'), (51, 0.5, 'tukey_51_0p5'), (40, 1.0, 'tukey_40_1')):
        assert np.allclose(bufferzone.tukey_window(n, a), g[key], rtol=0, atol=1e-15), key
    assert np.allclose(bufferzone._damp_flux(v, 0.15), v * g['tukey_200_0p3'], rtol=0, atol=1e-15)

    class M(object):
        def get_flux(self, profiles):
            return {'n': np.tile(v, (3, 1))}
    out = bufferzone.BufferZone(M(), 0.15).get_flux({})['n']
    assert out.shape == (3, 200) and np.allclose(out[2], v * g['tukey_200_0p3'], atol=1e-15)
    np.random.seed(5)
    nz = noisyflux.NoisyFlux(M(), 0.1, 0.05, 1 / 199).get_flux({})['n']
    assert nz.shape == (3, 200) and not np.allclose(nz[0], nz[1]) and np.array_equal(nz[:, 0], v[[0, 0, 0]])


def test_static_callback_detection():
    """Solver's trial calls: a callback whose own coefficients depend on x only is recognised; dependence on the
    profile, on time, or an exception is not; a static_H attribute overrides the trial."""
    from types import SimpleNamespace
    from tango_amd.solver import _callback_is_static
    x = np.linspace(0, 1, 12)
    H = multifield.HCoefficients

    def mk(cb, detect=True):
        f = SimpleNamespace(label='n', compute_all_H=cb)
        return SimpleNamespace(fields=[f], profiles={'n': 1 - 0.5 * x}, x=x, tArray=np.array([0.0, 2.0]), detectStaticH=detect)

    static = lambda t, xx, p, HT: H(H1=np.ones_like(xx), H7=np.where(xx < 0.3, 2.0, 0.0)) + HT
    assert _callback_is_static(mk(static))
    assert not _callback_is_static(mk(static, detect=False))
    assert not _callback_is_static(mk(lambda t, xx, p, HT: H(H1=np.ones_like(xx), H7=p['n']) + HT))
    assert not _callback_is_static(mk(lambda t, xx, p, HT: H(H1=np.ones_like(xx) * (1 + t)) + HT))
    assert not _callback_is_static(mk(lambda t, xx, p, HT: 1 / 0))
    assert not _callback_is_static(mk(lambda t, xx, p, HT: None))

    class Marked(object):
        static_H = False

        def __call__(self, t, xx, p, HT):
            return H(H1=np.ones_like(xx)) + HT
    assert not _callback_is_static(mk(Marked()))
    Marked.static_H = True
    assert _callback_is_static(mk(Marked(), detect=False))


def test_ewma_memory_handed_over_as_a_callable_is_fetched_once():
    """The fused solver paths leave an ensemble's EWMA memories on the device and hand lodestro_method.EWMA a
    callable; the first read calls it, later reads use the value; plain arrays and None behave as in the reference."""
    from tango_amd.lodestro_method import EWMA
    calls = []

    def fetch():
        calls.append(1)
        return np.array([1.0, 2.0])
    e = EWMA(0.3)
    assert e._yEWMA_lminus1 is None
    e.set_ewma_iterate(fetch)
    assert not calls
    assert np.array_equal(e._yEWMA_lminus1, [1.0, 2.0]) and np.array_equal(e._yEWMA_lminus1, [1.0, 2.0]) and len(calls) == 1
    e.set_ewma_iterate(np.array([5.0]))
    assert np.array_equal(e._yEWMA_lminus1, [5.0]) and len(calls) == 1
    e.reset_ewma_iterate()
    assert e._yEWMA_lminus1 is None


def test_short_quotient_rounds_like_a_division():
    """The kernels divide by launch constants (dt, dx, dx^2, theta's span Dmax - Dmin) as q = n * inv, q + (n - d q) * inv
    with inv the correctly rounded 1/d and both corrections fused (quot_by in csrc/tg_comm.h).  theta must round exactly
    as the reference's division does (1 - theta cancels down to 1e-14), so the sequence is run here in exact rational
    arithmetic (Fraction -> float rounds correctly, which makes every step the IEEE operation it stands for) on random
    operand pairs and on mantissas that are hard to round, and compared with the division bit for bit."""
    import math
    import random
    from fractions import Fraction as F
    rnd = random.Random(7)

    def mantissa(kind):
        if kind == 1:
            m = (1 << 52) - 1 - rnd.getrandbits(rnd.randint(0, 20))      # all ones, a few low bits cleared
        elif kind == 2:
            m = rnd.getrandbits(rnd.randint(0, 20))                       # next to a power of two
        else:
            m = rnd.getrandbits(52) | (kind == 3)
        return (1 << 52) | m

    beyond_one_ulp = 0
    for _ in range(20000):
        n = math.ldexp(mantissa(rnd.randint(0, 3)), rnd.randint(-80, 80) - 52)
        d = math.ldexp(mantissa(rnd.randint(0, 3)), rnd.randint(-80, 80) - 52)
        inv = float(F(1) / F(d))
        q = float(F(n) * F(inv))
        r = float(F(n) - F(d) * F(q))
        got = float(F(r) * F(inv) + F(q))
        want = n / d
        assert got == want, (n.hex(), d.hex())
        beyond_one_ulp += abs(F(q) - F(n) / F(d)) > F(math.ulp(want))
    assert beyond_one_ulp > 0       # the sample does contain first products more than an ulp off: the correction step is what rounds


def test_combine_savefiles(tmp_path, monkeypatch):
    """restart.combine_savefiles stacks the per-set save files in set order (tango/restart.py:164-212)."""
    from tango_amd import restart
    monkeypatch.chdir(tmp_path)
    with pytest.raises(ValueError, match='no files found'):
        restart.combine_savefiles('run')
    for k in (2, 0, 1):
        data = {'setNumber': k, 't': 1e4, 'timestepNumber': 1, 'iterationNumber': 10 * k + 9, 'labels': np.array(['n'])}
        for key in restart.FIELD_KEYS:
            data[f'n/{key}'] = np.full(5, float(k))
        np.savez(f'run_s{k}.npz', **data)
    name = restart.combine_savefiles('run')
    assert name == 'run_combined.npz'
    with np.load(name) as z:
        assert z['iterationNumber'].tolist() == [9, 19, 29] and z['setNumber'].tolist() == [0, 1, 2]
        assert z['n/profile'].shape == (3, 5) and z['n/profile'][:, 0].tolist() == [0.0, 1.0, 2.0]
    assert restart.find_latest_savefile('run') == 'run_s2.npz'       # the combined file is not a set
