"""Tridiagonal solve C_j u_{j-1} + B_j u_j + A_j u_{j+1} = D_j on the GPU.

Mirrors tango/tridiag.py: `solve` (reference :12-31) is the entry the rest of the package uses.  The
reference routes it to LAPACK dgtsv through SciPy (:88-97) and also ships a pure-Python Thomas sweep
(:33-63); here every variant runs a HIP kernel:
  `solve`             the wave-per-profile chunked elimination with parallel cyclic reduction (arrays of shape
                      (N,) or (profiles, N), host or device): the fast path, within rounding of the reference;
  `solve_with_scipy`  the lane-per-profile kernel on the interleaved batch-fastest layout, which performs dgtsv's
                      sweeps operation by operation (partial pivoting included): the reference's result bit for bit;
  `solve_python`      the same kernel behind the reference's asserts on A[-1] and C[0];
  `solve_interleaved` device arrays of shape (N, profiles), i.e. the interleaved batch-fastest layout: `exact=True` the
                      dgtsv kernel, `exact=False` (N <= 512) the fast path's algorithm on tiles of 16 profiles
                      transposed through LDS, its results bit for bit at half the memory traffic."""
import numpy as np

from . import _hip as H
from . import _ops


def solve(A, B, C, D):
    return _ops.tridiag_solve(A, B, C, D)


def solve_interleaved(A, B, C, D, exact=False):
    """Device arrays (N, profiles) in, device array (N, profiles) out."""
    return _ops.tridiag_solve(A, B, C, D, layout=H.BATCH_FASTEST if exact else H.BATCH_FASTEST_TILED)


def _solve_batch_fastest(A, B, C, D):
    one = A.ndim == 1
    dev = [H.DeviceArray.from_host(np.ascontiguousarray(np.atleast_2d(v).T)) for v in (A, B, C, D)]
    u = _ops.tridiag_solve(*dev, layout=H.BATCH_FASTEST).to_host().T
    return np.ascontiguousarray(u[0] if one else u)


def solve_with_scipy(A, B, C, D):
    """tango/tridiag.py:88-97: LAPACK dgtsv's sweeps, one lane per profile; A[-1] and C[0] are ignored as LAPACK's
    band storage ignores them.  Host arrays of shape (N,) or (profiles, N)."""
    A, B, C, D = (np.asarray(v, dtype=np.float64) for v in (A, B, C, D))
    assert B.shape == A.shape and C.shape == A.shape and D.shape == A.shape, 'lengths must be equal'
    return _solve_batch_fastest(A, B, C, D)


def solve_python(A, B, C, D):
    """Thomas algorithm (tango/tridiag.py:33-63), run as the batch-fastest lane-per-profile kernel.
    Asserts, like the reference, that lengths agree and that A[-1] == 0 and C[0] == 0."""
    A, B, C, D = (np.asarray(v, dtype=np.float64) for v in (A, B, C, D))
    assert B.shape[-1] == A.shape[-1] and C.shape[-1] == A.shape[-1] and D.shape[-1] == A.shape[-1], \
        'lengths must be equal'
    assert np.all(A[..., -1] == 0), 'last element of A must be zero'
    assert np.all(C[..., 0] == 0), 'first element of C must be zero'
    return _solve_batch_fastest(A, B, C, D)
