"""tg_picard_1f_step_static (the entry point that allocates nothing: its work space is an array inside the library)
against tg_picard_1f_step with a caller-supplied work space, through the C ABI, bit for bit, on two calls in a row,
the second on a smaller ensemble (the built-in array still holds the first call's contents)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(nb, n, its, static, tread=False, k=1):
    from tango_amd import _hip as H
    from tango_amd.ensemble import OneFieldEnsemble, shestakov_ensemble
    E = OneFieldEnsemble(shestakov_ensemble(nb, n), a_turb=0.3, a_prof=0.3, tol=1e-9, max_iter=its * k, segmented=not static, tread=tread)
    if tread:       # pseudo-timestep control: short first steps, halving of steps that move the profile by 5 % or more
        q = E.params
        q.pseudo_time, q.tread_initial_steps, q.tread_initial_dtau = 1, 1, 0.5
        q.tread_lightly, q.tread_max_change, q.tread_inner_count = 1, 0.05, n // 2
    if static:
        assert not E.params.work                 # no caller work space at all
    else:
        sized = H.lib.tg_picard_1f_work_bytes_tread if tread else H.lib.tg_picard_1f_work_bytes
        assert E.params.work and E.params.work_bytes == sized(nb, n)
    fn = H.lib.tg_picard_1f_step_static if static else H.lib.tg_picard_1f_step
    E.params.iters_per_launch = k
    for _ in range(its):
        H.check(fn(C.byref(E.state), C.byref(E.params), H.stream()))
    H.synchronize()
    return {k: getattr(E, k).to_host() for k in ('u', 'uE', 'DE', 'cE', 'status', 'iters', 'err')}


def test_static_work_space_matches_caller_work_space():
    import tango_amd
    assert tango_amd.device_count() > 0
    for nb, n, its in ((24, 8192, 3), (5, 8192, 2), (3, 16384, 2), (7, 4096, 2)):
        got = _run(nb, n, its, static=True)
        ref = _run(nb, n, its, static=False)
        for k in ref:
            assert np.array_equal(got[k], ref[k]), (nb, n, k)
        assert np.all(got['iters'] == its) and np.all(np.isfinite(got['u'])) and not np.any(got['status'] & 6)
    # several iterations per call: a chain of segment kernels at 8192 points; at 4096 the step prefers the workgroup kernel,
    # which keeps the profile in registers and needs no work space: the entry point passes such a call on to it
    for nb, n, its, k in ((9, 8192, 2, 3), (7, 4096, 2, 3)):
        got, ref = _run(nb, n, its, static=True, k=k), _run(nb, n, its, static=False, k=k)
        for key in ref:
            assert np.array_equal(got[key], ref[key]), (nb, n, k, key)
        assert np.all(got['iters'] == its * k) and not np.any(got['status'] & 6)
    # under pseudo-timestep control the built-in array also holds the previous iterates
    got, ref = _run(6, 4096, 3, static=True, tread=True), _run(6, 4096, 3, static=False, tread=True)
    for k in ref:
        assert np.array_equal(got[k], ref[k]), k
    assert np.all(got['iters'] == 3) and not np.any(got['status'] & 6)


def test_static_work_space_rejects_what_it_cannot_hold():
    from tango_amd import _hip as H
    from tango_amd.ensemble import OneFieldEnsemble, shestakov_ensemble
    # too many grid points for the built-in array: 640 x 8192 needs more than TG_STATIC_WORK_BYTES
    assert H.lib.tg_picard_1f_work_bytes(640, 8192) > H.STATIC_WORK_BYTES >= H.lib.tg_picard_1f_work_bytes(512, 8192)
    E = OneFieldEnsemble(shestakov_ensemble(640, 8192), max_iter=1, keep_err=False, segmented=False)
    with pytest.raises(ValueError, match='TG_STATIC_WORK_BYTES'):
        H.check(H.lib.tg_picard_1f_step_static(C.byref(E.state), C.byref(E.params), H.stream()))
    # a profile length whose step needs no work space is not this entry point's business
    E = OneFieldEnsemble(shestakov_ensemble(8, 512), max_iter=1, keep_err=False)
    with pytest.raises(NotImplementedError):
        H.check(H.lib.tg_picard_1f_step_static(C.byref(E.state), C.byref(E.params), H.stream()))
