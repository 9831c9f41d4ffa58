"""Lane-level NumPy model of the GPU tridiagonal solve (TEST INFRASTRUCTURE ONLY).

The HIP kernels solve one profile with T cooperating lanes, each owning P contiguous grid
points: a local elimination inside every lane's chunk (a Thomas sweep that carries one
fill-in column for the unknown just left of the chunk), a reduced tridiagonal system with one
unknown per lane (the last point of each chunk) solved by parallel cyclic reduction, and a
local back substitution.  This file performs exactly those steps with array operations over
the lane axis, so the algorithm and its rounding behaviour can be checked on the CPU against
LAPACK (oracle.tango_oracle.tridiag_solve) before any kernel runs.
"""
import numpy as np


def partition_pcr_solve(A, B, C, D, P=8):
    """Solve C_j u_{j-1} + B_j u_j + A_j u_{j+1} = D_j (1-D arrays, length n) with chunks of P
    points per lane.  A[n-1] and C[0] are ignored, as LAPACK's band storage ignores them."""
    A, B, C, D = (np.array(v, dtype=float) for v in (A, B, C, D))
    n = len(B)
    A[-1] = 0.0
    C[0] = 0.0
    T = 1
    while T * P < n:
        T *= 2
    pad = T * P - n
    a = np.concatenate((A, np.zeros(pad))).reshape(T, P)
    b = np.concatenate((B, np.ones(pad))).reshape(T, P)
    c = np.concatenate((C, np.zeros(pad))).reshape(T, P)
    d = np.concatenate((D, np.zeros(pad))).reshape(T, P)

    # downward sweep: x_i + e_i x_{i+1} = f_i - l_i xL   (xL = last unknown of the lane to the left)
    e = np.empty((T, P)); f = np.empty((T, P)); l = np.empty((T, P))
    inv = 1.0 / b[:, 0]
    e[:, 0] = a[:, 0] * inv; f[:, 0] = d[:, 0] * inv; l[:, 0] = c[:, 0] * inv
    for i in range(1, P):
        inv = 1.0 / (b[:, i] - c[:, i] * e[:, i - 1])
        e[:, i] = a[:, i] * inv
        f[:, i] = (d[:, i] - c[:, i] * f[:, i - 1]) * inv
        l[:, i] = -(c[:, i] * l[:, i - 1]) * inv
    # upward sweep: x_i = F_i - L_i xL - E_i x_{P-1}  for i <= P-2
    F = f.copy(); L = l.copy(); E = e.copy()
    for i in range(P - 3, -1, -1):
        F[:, i] = f[:, i] - e[:, i] * F[:, i + 1]
        L[:, i] = l[:, i] - e[:, i] * L[:, i + 1]
        E[:, i] = -e[:, i] * E[:, i + 1]

    def from_right(v, fill):   # value held by lane t+1
        return np.concatenate((v[1:], [fill]))

    if P > 1:
        # reduced system in y_t = x_{P-1} of lane t
        F0 = from_right(F[:, 0], 0.0); L0 = from_right(L[:, 0], 0.0); E0 = from_right(E[:, 0], 0.0)
        el, ll, fl = e[:, P - 1], l[:, P - 1], f[:, P - 1]
        g = 1.0 - el * L0
        s = ll / g
        p = -(el * E0) / g
        r = (fl - el * F0) / g
    else:
        s, p, r = l[:, 0].copy(), e[:, 0].copy(), f[:, 0].copy()

    # parallel cyclic reduction, unit diagonal kept by normalising every step
    h = 1
    while h < T:
        def sh(v, k, fill):
            out = np.full(T, fill)
            if k > 0:
                out[k:] = v[:-k]
            else:
                out[:k] = v[-k:]
            return out
        sm, pm, rm = sh(s, h, 0.0), sh(p, h, 0.0), sh(r, h, 0.0)      # lane t-h
        sp, pp, rp = sh(s, -h, 0.0), sh(p, -h, 0.0), sh(r, -h, 0.0)   # lane t+h
        inv = 1.0 / (1.0 - pm * s - sp * p)
        r = (r - rm * s - rp * p) * inv
        s, p = -(sm * s) * inv, -(pp * p) * inv
        h *= 2
    y = r
    x = np.empty((T, P))
    x[:, P - 1] = y
    if P > 1:
        yl = np.concatenate(([0.0], y[:-1]))
        for i in range(P - 1):
            x[:, i] = F[:, i] - L[:, i] * yl - E[:, i] * y
    return x.reshape(-1)[:n]
