"""CPU oracle for Tango's implicit transport inner loop (TEST INFRASTRUCTURE ONLY).

This module is a NumPy restatement of the algorithms on the hot path of the reference
(sundials-codes/tango).  It exists only to check the HIP kernels: it may be imported by
tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg, never by the
product package (tango_amd/), which must fail loudly when the HIP library is missing.

Every function works on arrays whose LAST axis is the radial grid (length N) and whose
leading axes (if any) index independent profiles, so one call restates the reference for a
whole ensemble; with a 1-D input it is the reference's own per-profile semantics.  Arithmetic
is written in the same order as the reference expressions so that results agree with the
imported reference to the last few ulps (pinned by tests/golden/*.npz, which were produced
by importing the reference itself; see tests/golden/make_golden.py).

The two linear solves of the reference live in a third-party dependency: SciPy ->
LAPACK dgtsv / dgbsv (reference call sites tango/tridiag.py:94-96 and
tango/JK_to_matrix_fd.py:158-164; SciPy version unpinned by the reference, 1.15.3 in this
image).  `tridiag_solve` / `pair_solve` call the same SciPy routine when SciPy is present and
otherwise fall back to restated, non-pivoting Thomas / 2x2 block-Thomas eliminations
(`thomas`, `pair_block_thomas`).

Reference citations are relative to the reference checkout (tango/...).
"""
from __future__ import annotations

import numpy as np

try:  # SciPy is what the reference itself calls; optional here.
    import scipy.linalg as _sla
except Exception:  # pragma: no cover - image always has scipy
    _sla = None


# --------------------------------------------------------------------------------------
# derivatives  (tango/derivatives.py:9-41)
# --------------------------------------------------------------------------------------
def ddx_edge1(u, dx):
    """Centered difference, one-sided two-point stencils at both ends
    (tango/derivatives.py:9-24)."""
    u = np.asarray(u, dtype=float)
    out = np.empty_like(u)
    out[..., 0] = (u[..., 1] - u[..., 0]) / dx
    out[..., 1:-1] = (u[..., 2:] - u[..., :-2]) / (2 * dx)
    out[..., -1] = (u[..., -1] - u[..., -2]) / dx
    return out


def ddx_edge2(u, dx):
    """Centered difference, second-order three-point stencils at both ends
    (tango/derivatives.py:26-41)."""
    u = np.asarray(u, dtype=float)
    out = np.empty_like(u)
    out[..., 0] = (-3 / 2 * u[..., 0] + 2 * u[..., 1] - 1 / 2 * u[..., 2]) / dx
    out[..., 1:-1] = (u[..., 2:] - u[..., :-2]) / (2 * dx)
    out[..., -1] = (3 / 2 * u[..., -1] - 2 * u[..., -2] + 1 / 2 * u[..., -3]) / dx
    return out


# --------------------------------------------------------------------------------------
# smoother  (tango/smoother.py:6-57)
# --------------------------------------------------------------------------------------
def odd_window(w):
    """Even windows are widened by one (tango/smoother.py:50-57)."""
    if w % 2 == 0:
        return w + 1
    if w % 2 == 1:
        return w
    raise ValueError('n does not appear to be an integer.')


def moving_average(x, window):
    """Centered moving average formed from differences of a running sum, with shrinking
    symmetric windows (1, 3, 5, ...) on the (window-1)/2 points at either end
    (tango/smoother.py:13-48)."""
    x = np.asarray(x, dtype=float)
    if window == 1:
        return x
    assert window % 2 == 1, 'windowSize must be an odd integer.'
    h = int(window - 1) // 2
    n = x.shape[-1]
    out = np.zeros_like(x)
    pad = np.zeros(x.shape[:-1] + (1,))
    run = np.cumsum(np.concatenate((pad, x), axis=-1), axis=-1)
    out[..., h:n - h] = (run[..., window:] - run[..., :-window]) / window
    for j in range(h):
        w = 2 * j + 1
        out[..., j] = np.mean(x[..., :w], axis=-1)
        out[..., n - 1 - j] = np.mean(x[..., n - w:], axis=-1)
    return out


# --------------------------------------------------------------------------------------
# EWMA relaxation  (tango/lodestro_method.py:682-728)
# --------------------------------------------------------------------------------------
def ewma_next(y, y_prev, a):
    """y_bar_l = a*y_l + (1-a)*y_bar_{l-1}; the first call (y_prev None) seeds
    y_bar_{l-1} = y_l (tango/lodestro_method.py:706-722)."""
    if y_prev is None:
        y_prev = y
    return a * y + (1 - a) * y_prev


# --------------------------------------------------------------------------------------
# flux -> (D, c) split  (tango/lodestro_method.py:563-679)
# --------------------------------------------------------------------------------------
def ftheta(DHat, dpdx, Dmin, Dmax, thr):
    """Default theta rule (tango/lodestro_method.py:640-679)."""
    theta = np.ones_like(DHat)
    theta[DHat < Dmin] = 0
    with np.errstate(invalid='ignore'):
        small = np.abs(dpdx) < thr
        mid = small & (DHat >= Dmin) & (DHat <= Dmax)
        theta[mid] = (1 + (Dmax - DHat[mid]) / (Dmax - Dmin)) / 2
        theta[small & (DHat > Dmax)] = 0.5
    assert np.count_nonzero((theta >= 0) & (theta <= 1)) == np.size(theta), \
        'some theta is not between 0 and 1'
    return theta


def flux_split(flux, p, dx, Dmin, Dmax, thr, gxx=1, gradx=1):
    """Split a flux into diffusive D and convective c so that flux = -D dp/dx + c p
    (tango/lodestro_method.py:624-638).  Returns (D, c, DHat, cHat, theta)."""
    flux = np.asarray(flux, dtype=float)
    p = np.asarray(p, dtype=float)
    dpdx = ddx_edge1(p, dx)
    with np.errstate(divide='ignore', invalid='ignore'):
        DHat = -flux / (gxx * dpdx)
        DHat[dpdx == 0] = 0
        cHat = flux / (gradx * p)
    theta = ftheta(DHat, dpdx, Dmin, Dmax, thr)
    D = theta * DHat
    c = (1 - theta) * cHat
    return D, c, DHat, cHat, theta


def dc_to_h(D, c, Vprime=None, gxx=1, gradx=1):
    """H2 = V' D <g^xx>, H3 = -V' c <|grad x|>; Cartesian when V' is absent
    (tango/lodestro_method.py:390-408)."""
    if Vprime is not None:
        return Vprime * D * gxx, -Vprime * c * gradx
    return D, -c


# --------------------------------------------------------------------------------------
# finite-difference assembly  (tango/HToMatrixFD.py:53-299)
# --------------------------------------------------------------------------------------
def half_grid(u):
    """u_{j+1/2} = (u_j + u_{j+1})/2 in slot j, last slot zero (tango/HToMatrixFD.py:207-219)."""
    h = np.zeros_like(u)
    h[..., :-1] = (u[..., :-1] + u[..., 1:]) / 2
    return h


def fbeta(H2h, H3h, dx):
    """Adaptive upwinding weight on the half grid, gamma = 0.9 (tango/HToMatrixFD.py:268-299)."""
    gamma = 0.9
    c = -H3h
    beta = np.zeros_like(H3h)
    star = np.zeros_like(H3h)
    pos = c > 0
    star[pos] = 1 - H2h[pos] / (c[pos] * dx)
    beta[pos & (star <= 1 / 2)] = 1 / 2
    m = pos & (star > 1 / 2)
    beta[m] = gamma * star[m] + (1 - gamma)
    beta[c == 0] = 0
    neg = c < 0
    star[neg] = -H2h[neg] / (c[neg] * dx)
    beta[neg & (star >= 1 / 2)] = 1 / 2
    m = neg & (star < 1 / 2)
    beta[m] = gamma * star[m]
    return beta


def _bc(shape_like, v):
    """Broadcast a per-profile scalar against the leading (batch) axes."""
    v = np.asarray(v, dtype=float)
    if v.ndim == 0:
        return v
    return v.reshape(v.shape + (1,) * (shape_like.ndim - v.ndim))


def h_to_matrix(dt, dx, UL, U_mminus1, H1, H2=None, H3=None, H4=None, H6=None, H7=None):
    """Tridiagonal rows (A super, B diag, C sub, D rhs) of one Picard update, no-flux left
    boundary and Dirichlet right boundary.  Contributions are accumulated in the reference's
    order H1, H2, H3, H4, H6, H7, boundary (tango/HToMatrixFD.py:87-117); dt and UL may be
    per-profile arrays over the leading axes."""
    U = np.asarray(U_mminus1, dtype=float)
    n = U.shape[-1]
    H1 = np.asarray(H1, dtype=float)
    assert H1.shape[-1] == n
    shape = np.broadcast_shapes(U.shape, H1.shape)
    A = np.zeros(shape)
    Bd = np.zeros(shape)
    C = np.zeros(shape)
    g = np.zeros(shape)
    dtb = _bc(A, dt)

    # H1: B_j += H1/dt ; g_j += -H1*U/dt ; nothing on the last row   (:126-132)
    t = np.broadcast_to(H1 / dtb, shape).copy()
    t[..., n - 1] = 0
    Bd += t
    t = np.broadcast_to(-H1 * U / dtb, shape).copy()
    t[..., n - 1] = 0
    g += t

    if H2 is not None:
        H2 = np.asarray(H2, dtype=float)
        assert H2.shape[-1] == n
        h = np.broadcast_to(half_grid(H2), shape)
        # interior rows (:139-142)
        A[..., 1:-1] += -h[..., 1:-1] / dx ** 2
        Bd[..., 1:-1] += (h[..., :-2] + h[..., 1:-1]) / dx ** 2
        C[..., 1:-1] += -h[..., :-2] / dx ** 2
        # left boundary row (:145-146)
        A[..., 0] += -h[..., 0] / dx ** 2
        Bd[..., 0] += h[..., 0] / dx ** 2
    if H3 is not None:
        H3 = np.asarray(H3, dtype=float)
        assert H3.shape[-1] == n
        if H2 is None:
            H2 = np.zeros(n)
        h2 = half_grid(np.asarray(H2, dtype=float))
        h3 = half_grid(H3)
        h2, h3 = np.broadcast_arrays(h2, h3)
        beta = np.broadcast_to(fbeta(h2, h3, dx), shape)
        h3 = np.broadcast_to(h3, shape)
        # interior rows (:159-162)
        A[..., 1:-1] += -h3[..., 1:-1] * (1 - beta[..., 1:-1]) / dx
        Bd[..., 1:-1] += (-h3[..., 1:-1] * beta[..., 1:-1]
                          + h3[..., :-2] * (1 - beta[..., :-2])) / dx
        C[..., 1:-1] += h3[..., :-2] * beta[..., :-2] / dx
        # left boundary row (:165-166)
        A[..., 0] += -h3[..., 0] * (1 - beta[..., 0]) / dx
        Bd[..., 0] += -h3[..., 0] * beta[..., 0] / dx
    if H4 is not None:
        H4 = np.broadcast_to(np.asarray(H4, dtype=float), shape)
        assert H4.shape[-1] == n
        g[..., 1:-1] += -(H4[..., 2:] - H4[..., :-2]) / (2 * dx)   # (:173)
        g[..., 0] += -(H4[..., 0] + H4[..., 1]) / (2 * dx)         # (:176)
    if H6 is not None:
        H6 = np.asarray(H6, dtype=float)
        assert H6.shape[-1] == n
        t = np.broadcast_to(-H6, shape).copy()                    # (:186-188)
        t[..., n - 1] = 0
        Bd += t
    if H7 is not None:
        H7 = np.asarray(H7, dtype=float)
        assert H7.shape[-1] == n
        t = np.broadcast_to(-H7, shape).copy()                    # (:194-197)
        t[..., n - 1] = 0
        g += t
    # Dirichlet row (:200-204)
    Bd[..., n - 1] += 1
    g[..., n - 1] += -np.asarray(UL, dtype=float)
    return A, Bd, C, -g


def h8_to_g(H8):
    """Coupling column G_j = -H8_j with a zero on the Dirichlet row
    (tango/JK_to_matrix_fd.py:119-130)."""
    H8 = np.asarray(H8, dtype=float)
    G = np.zeros_like(H8)
    G[..., :-1] = -H8[..., :-1]
    return G


def jk_to_matrix(dt, dx, rightBC, psi_mminus1, J, K):
    """Two pair-coupled fields: J, K are dicts with keys among H1,H2,H3,H4,H6,H7,H8 (H1
    required).  Returns ((AU,BU,CU,DU,GW), (AW,BW,CW,DW,GU)) as in
    tango/JK_to_matrix_fd.py:89-116 (the U equation carries the column that multiplies W)."""
    assert J.get('H1') is not None, 'Must supply J1'
    assert K.get('H1') is not None, 'Must supply K1'
    keys = ('H2', 'H3', 'H4', 'H6', 'H7')
    AU, BU, CU, DU = h_to_matrix(dt, dx, rightBC[0], psi_mminus1[0], J['H1'],
                                 **{k: J.get(k) for k in keys})
    AW, BW, CW, DW = h_to_matrix(dt, dx, rightBC[1], psi_mminus1[1], K['H1'],
                                 **{k: K.get(k) for k in keys})
    GW = h8_to_g(J['H8']) if J.get('H8') is not None else np.zeros_like(AU)
    GU = h8_to_g(K['H8']) if K.get('H8') is not None else np.zeros_like(AW)
    return (AU, BU, CU, DU, GW), (AW, BW, CW, DW, GU)


# --------------------------------------------------------------------------------------
# linear solves  (tango/tridiag.py:12-97, tango/JK_to_matrix_fd.py:132-192)
# --------------------------------------------------------------------------------------
def thomas(A, B, C, D):
    """Non-pivoting Thomas elimination of C_j u_{j-1} + B_j u_j + A_j u_{j+1} = D_j
    (what tango/tridiag.py:33-63 computes, written in the usual sign convention)."""
    A, B, C, D = (np.asarray(v, dtype=float) for v in (A, B, C, D))
    n = B.shape[-1]
    cp = np.empty_like(B)
    dp = np.empty_like(B)
    cp[..., 0] = A[..., 0] / B[..., 0]
    dp[..., 0] = D[..., 0] / B[..., 0]
    for j in range(1, n):
        den = B[..., j] - C[..., j] * cp[..., j - 1]
        cp[..., j] = A[..., j] / den
        dp[..., j] = (D[..., j] - C[..., j] * dp[..., j - 1]) / den
    u = np.empty_like(B)
    u[..., n - 1] = dp[..., n - 1]
    for j in range(n - 2, -1, -1):
        u[..., j] = dp[..., j] - cp[..., j] * u[..., j + 1]
    return u


def tridiag_solve(A, B, C, D):
    """The reference's tridiagonal solve: SciPy solve_banded((1,1)) = LAPACK dgtsv with
    partial pivoting (tango/tridiag.py:88-97).  One LAPACK call per profile."""
    A, B, C, D = (np.asarray(v, dtype=float) for v in (A, B, C, D))
    if _sla is None:  # pragma: no cover
        return thomas(A, B, C, D)
    if B.ndim == 1:
        ab = np.array((np.roll(A, 1), B, np.roll(C, -1)))
        return _sla.solve_banded((1, 1), ab, D)
    lead = B.shape[:-1]
    n = B.shape[-1]
    A2, B2, C2, D2 = (np.broadcast_to(v, lead + (n,)).reshape(-1, n) for v in (A, B, C, D))
    out = np.empty_like(B2)
    for i in range(B2.shape[0]):
        ab = np.array((np.roll(A2[i], 1), B2[i], np.roll(C2[i], -1)))
        out[i] = _sla.solve_banded((1, 1), ab, D2[i])
    return out.reshape(lead + (n,))


def _pair_band(U, W):
    AU, BU, CU, DU, GW = U
    AW, BW, CW, DW, GU = W
    n = len(BU)

    def il(a, b):
        c = np.zeros(2 * n)
        c[0::2] = a
        c[1::2] = b
        return c
    z = np.zeros(n)
    ab = np.array((np.roll(il(AU, AW), 2), np.roll(il(GW, z), 1), il(BU, BW),
                   il(GU, z), np.roll(il(CU, CW), -2)))
    return ab, il(DU, DW)


def pair_solve(U, W):
    """The reference's pair-coupled solve: unknowns interleaved [U0,W0,U1,W1,...], five
    bands, SciPy solve_banded((2,2)) = LAPACK dgbsv (tango/JK_to_matrix_fd.py:144-166)."""
    U = tuple(np.asarray(v, dtype=float) for v in U)
    W = tuple(np.asarray(v, dtype=float) for v in W)
    if _sla is None:  # pragma: no cover
        return pair_block_thomas(U, W)
    if U[1].ndim == 1:
        ab, rhs = _pair_band(U, W)
        psi = _sla.solve_banded((2, 2), ab, rhs)
        return psi[0::2], psi[1::2]
    lead = U[1].shape[:-1]
    n = U[1].shape[-1]
    U2 = [np.broadcast_to(v, lead + (n,)).reshape(-1, n) for v in U]
    W2 = [np.broadcast_to(v, lead + (n,)).reshape(-1, n) for v in W]
    ou = np.empty_like(U2[1])
    ow = np.empty_like(U2[1])
    for i in range(ou.shape[0]):
        ab, rhs = _pair_band([v[i] for v in U2], [v[i] for v in W2])
        psi = _sla.solve_banded((2, 2), ab, rhs)
        ou[i], ow[i] = psi[0::2], psi[1::2]
    return ou.reshape(lead + (n,)), ow.reshape(lead + (n,))


def pair_block_thomas(U, W):
    """Non-pivoting block-Thomas elimination of the same pair-coupled system seen as a
    block-tridiagonal matrix with 2x2 diagonal blocks [[BU,GW],[GU,BW]] and diagonal
    off-diagonal blocks diag(AU,AW), diag(CU,CW)."""
    AU, BU, CU, DU, GW = (np.asarray(v, dtype=float) for v in U)
    AW, BW, CW, DW, GU = (np.asarray(v, dtype=float) for v in W)
    n = BU.shape[-1]
    sh = BU.shape
    # E_j (2x2) = inv(Bhat_j) A_j ; F_j (2) = inv(Bhat_j) rhs_j
    E = np.zeros(sh + (2, 2))
    F = np.zeros(sh + (2,))
    for j in range(n):
        b00, b01, b10, b11 = BU[..., j], GW[..., j], GU[..., j], BW[..., j]
        r0, r1 = DU[..., j], DW[..., j]
        if j > 0:
            cu, cw = CU[..., j], CW[..., j]
            b00 = b00 - cu * E[..., j - 1, 0, 0]
            b01 = b01 - cu * E[..., j - 1, 0, 1]
            b10 = b10 - cw * E[..., j - 1, 1, 0]
            b11 = b11 - cw * E[..., j - 1, 1, 1]
            r0 = r0 - cu * F[..., j - 1, 0]
            r1 = r1 - cw * F[..., j - 1, 1]
        det = b00 * b11 - b01 * b10
        i00, i01, i10, i11 = b11 / det, -b01 / det, -b10 / det, b00 / det
        au, aw = AU[..., j], AW[..., j]
        E[..., j, 0, 0] = i00 * au
        E[..., j, 0, 1] = i01 * aw
        E[..., j, 1, 0] = i10 * au
        E[..., j, 1, 1] = i11 * aw
        F[..., j, 0] = i00 * r0 + i01 * r1
        F[..., j, 1] = i10 * r0 + i11 * r1
    u = np.empty(sh)
    w = np.empty(sh)
    u[..., n - 1] = F[..., n - 1, 0]
    w[..., n - 1] = F[..., n - 1, 1]
    for j in range(n - 2, -1, -1):
        u[..., j] = F[..., j, 0] - E[..., j, 0, 0] * u[..., j + 1] - E[..., j, 0, 1] * w[..., j + 1]
        w[..., j] = F[..., j, 1] - E[..., j, 1, 0] * u[..., j + 1] - E[..., j, 1, 1] * w[..., j + 1]
    return u, w


def block3_dense_solve(Cb, Bb, Ab, rhs):
    """Dense reference solution of a block-tridiagonal system with 3x3 blocks: row j reads
    Cb[j] x_{j-1} + Bb[j] x_j + Ab[j] x_{j+1} = rhs[j] with Cb[0] = Ab[N-1] = 0.
    Shapes (N,3,3) and (N,3).  There is no such routine in the reference (its three-field
    system is one scalar and one pair-coupled solve); this is the checker for the 3x3 block
    solver, which is a strict superset of that structure."""
    n = Bb.shape[0]
    M = np.zeros((3 * n, 3 * n))
    for j in range(n):
        M[3 * j:3 * j + 3, 3 * j:3 * j + 3] = Bb[j]
        if j > 0:
            M[3 * j:3 * j + 3, 3 * j - 3:3 * j] = Cb[j]
        if j < n - 1:
            M[3 * j:3 * j + 3, 3 * j + 3:3 * j + 6] = Ab[j]
    return np.linalg.solve(M, rhs.reshape(-1)).reshape(n, 3)


# --------------------------------------------------------------------------------------
# residual and convergence  (tango/fieldgroups.py:122-144,226-256; tango/solver.py:441-475)
# --------------------------------------------------------------------------------------
def _shift_up(u):   # u_{j+1}, zero past the end
    return np.concatenate((u[..., 1:], np.zeros(u.shape[:-1] + (1,))), axis=-1)


def _shift_dn(u):   # u_{j-1}, zero before the start
    return np.concatenate((np.zeros(u.shape[:-1] + (1,)), u[..., :-1]), axis=-1)


def residual(A, B, C, f, u, G=None, w=None):
    """Normalised residual (A u_{j+1} + B u_j + C u_{j-1} [+ G w_j] - f) / max|f|
    (tango/fieldgroups.py:139-140, 247-251)."""
    u = np.asarray(u, dtype=float)
    r = A * _shift_up(u) + B * u + C * _shift_dn(u)
    if G is not None:
        r = r + G * w
    r = r - f
    return r / np.max(np.abs(f), axis=-1, keepdims=True)


def rms_of(resids):
    """Global residual over the fields of one profile: sqrt(sum r^2 / total length)
    (tango/solver.py:464-470).  `resids` is a list of (..., N) arrays; returns (...)."""
    total = 0
    ssq = 0
    for r in resids:
        total += r.shape[-1]
        ssq = ssq + np.sum(r ** 2, axis=-1)
    return np.sqrt(1 / total * ssq)


def profiles_invalid(profiles):
    """True where any profile has a non-finite or negative entry (tango/solver.py:420-439)."""
    bad = False
    for p in profiles:
        bad = bad | ~np.all(np.isfinite(p), axis=-1) | np.any(p < 0, axis=-1)
    return bad


# --------------------------------------------------------------------------------------
# analytic flux models used by the benchmark configurations
# --------------------------------------------------------------------------------------
def shestakov_flux(n, dx, p=None):
    """Gamma = -(dn/dx)^3 / n^2 (tango/extras/shestakov_nonlinear_diffusion.py:51-61); with p
    set, the generalised |dn/dx / n|^p form (examples/stripped_down_tango_example.py:59-65)."""
    n = np.asarray(n, dtype=float)
    dndx = ddx_edge1(n, dx)
    if p is None:
        return -dndx ** 3 / n ** 2
    return -np.abs((dndx / n) ** p) * dndx


def shestakov_source(x, S0=1, delta=0.1):
    """S0 where x < delta, else 0 (tango/extras/shestakov_nonlinear_diffusion.py:63-68);
    S0 and delta may be per-profile arrays."""
    x = np.asarray(x, dtype=float)
    S0 = np.asarray(S0, dtype=float)
    delta = np.asarray(delta, dtype=float)
    if S0.ndim or delta.ndim:
        S0 = S0.reshape(S0.shape + (1,) * (1 if S0.ndim else 0))
        delta = delta.reshape(delta.shape + (1,) * (1 if delta.ndim else 0))
    return np.where(x < delta, S0, 0.0) * np.ones_like(x)


def shestakov_steady_state(x, nL, S0=1, delta=0.1):
    """Exact steady state (tango/extras/shestakov_nonlinear_diffusion.py:87-102)."""
    x = np.asarray(x, dtype=float)
    nright = (nL ** (1 / 3) + 1 / 3 * (S0 * delta) ** (1 / 3) * (1 - x)) ** 3
    nleft = (nL ** (1 / 3) + 1 / 3 * (S0 * delta) ** (1 / 3)
             * (1 - delta + (3 / 4) * (delta - x ** (4 / 3) / delta ** (1 / 3)))) ** 3
    return np.where(x < delta, nleft, nright)


def threefield_flux(n, pi, pe, dx):
    """D = (dpi/dx)^2/pi^2 shared by the three fluxes
    (examples/shestakov_threefield_nonlinearnu.py:55-73)."""
    dn, dpi, dpe = ddx_edge1(n, dx), ddx_edge1(pi, dx), ddx_edge1(pe, dx)
    D = dpi ** 2 / pi ** 2
    return -D * dn, -D * dpi, -D * dpe


def threefield_sources(x):
    """(examples/shestakov_threefield_nonlinearnu.py:75-92)."""
    x = np.asarray(x, dtype=float)
    Sn = np.where(x < 0.3, 8.0, 0.0)
    Si = np.where(x < 0.1, 1.0, 0.0)
    Se = np.where(x < 0.4, 3 * x, 0.0)
    return Sn, Si, Se


def calc_nu(nu0, n, pi, pe):
    """nu = nu0 / (pi/n + pe/n)^(3/2) (examples/shestakov_threefield_nonlinearnu.py:108-110)."""
    nu0 = np.asarray(nu0, dtype=float)
    if nu0.ndim:
        nu0 = nu0[..., None]
    return nu0 / ((pi / n + pe / n) ** (3 / 2))


# --------------------------------------------------------------------------------------
# one Picard iteration, one field  (call stack: tango/solver.py:188-272 ->
#   tango/multifield.py:169-206 -> tango/lodestro_method.py:327-388 ->
#   tango/HToMatrixFD.py:53-117 -> tango/fieldgroups.py:122-144 -> tango/tridiag.py:88-97)
# --------------------------------------------------------------------------------------
THETA_DEFAULT = dict(Dmin=1e-5, Dmax=1e13, dpdxThreshold=10)


class State1F:
    """Persistent per-profile state of a one-field solve: the profile, the profile of the
    previous timestep, and the three EWMA memories of the lodestro method."""

    def __init__(self, u, u_mminus1=None):
        self.u = np.array(u, dtype=float)
        self.um1 = self.u.copy() if u_mminus1 is None else np.array(u_mminus1, dtype=float)
        self.uE = None
        self.DE = None
        self.cE = None


def turb_coeffs_1f(st, dx, a_turb, a_prof, theta=THETA_DEFAULT, flux_fn=None, smooth_window=1,
                   Vprime=None, gxx=1, gradx=1, gxx_turb=1, gradx_turb=1):
    """LoDestro stage of an iteration for one field: EWMA(profile) -> flux -> smoothing ->
    split -> EWMA(D), EWMA(c) -> (H2, H3).  Mutates the EWMA memories in `st`
    (tango/lodestro_method.py:342-375)."""
    st.uE = ewma_next(st.u, st.uE, a_prof)
    flux = (flux_fn or (lambda n: shestakov_flux(n, dx)))(st.uE)
    sm = moving_average(flux, odd_window(smooth_window))
    D, c, DHat, cHat, th = flux_split(sm, st.uE, dx, theta['Dmin'], theta['Dmax'],
                                      theta['dpdxThreshold'], gxx_turb, gradx_turb)
    st.DE = ewma_next(D, st.DE, a_turb)
    st.cE = ewma_next(c, st.cE, a_turb)
    H2, H3 = dc_to_h(st.DE, st.cE, Vprime, gxx, gradx)
    return H2, H3, dict(flux=flux, smoothed=sm, D=D, c=c, DHat=DHat, cHat=cHat, theta=th)


def picard_step_1f(st, dt, dx, UL, H1, H7, a_turb, a_prof, theta=THETA_DEFAULT, flux_fn=None,
                   smooth_window=1, solve=tridiag_solve, **geom):
    """One full iteration for one field.  Returns (rms residual evaluated before the solve
    with the new matrix on the current profile, invalid flag of the new profile); the new
    profile is left in st.u."""
    H2, H3, _ = turb_coeffs_1f(st, dx, a_turb, a_prof, theta, flux_fn, smooth_window, **geom)
    A, B, C, f = h_to_matrix(dt, dx, UL, st.um1, H1, H2=H2, H3=H3, H7=H7)
    rms = rms_of([residual(A, B, C, f, st.u)])
    st.u = solve(A, B, C, f)
    return rms, profiles_invalid([st.u])


def run_1f_timesteps(u0, tArray, dx, UL, H1, H7, a_turb, a_prof, tol, max_iter, theta=THETA_DEFAULT):
    """The timestep loop of tango/solver.py:117-186 for one profile: each timestep iterates to convergence with
    u_{m-1} = the previous converged profile; the EWMA memories persist across timesteps.  Returns the list of
    (profile, iterations, errHistory) per timestep; stops after a timestep that did not converge."""
    st = State1F(np.asarray(u0, dtype=float))
    out = []
    for m in range(1, len(tArray)):
        dt = tArray[m] - tArray[m - 1]
        st.um1 = st.u.copy()
        err = []
        done = bad = False
        while len(err) < max_iter and not done and not bad:
            rms, bad = picard_step_1f(st, dt, dx, UL, H1, H7, a_turb, a_prof, theta)
            err.append(float(rms))
            done = bool(rms < tol)
            bad = bool(bad)
        out.append((st.u.copy(), len(err), np.array(err)))
        if not done:
            break
    return out


def is_unacceptable(old, new, max_fractional_change=0.2, inner_fraction=0.5):
    """tango/solver.py:564-595: any point of the inner part of the profile changed by the given fraction or more."""
    k = int(inner_fraction * len(old))
    with np.errstate(divide='ignore', invalid='ignore'):
        frac = (new[:k] - old[:k]) / old[:k]
    return bool(np.any(np.abs(frac) >= max_fractional_change))


def picard_step_1f_tread(st, it, dt, dx, UL, H1, H7, a_turb, a_prof, theta=THETA_DEFAULT, tread_lightly=None,
                         tread_initially=None, solve=tridiag_solve):
    """One iteration with the pseudo-timestep controls of tango/solver.py:198-235,311-372 for one profile (1-D
    arrays).  tread_lightly: dict with optional maxFractionalChange / innerFraction, or None;
    tread_initially: dict with numInitialSteps and dtau, or None.  The previous iterate takes the u_{m-1} slot;
    an unacceptable step is redone from the previous iterate with the same transport coefficients and
    dtau = dt/2, dt/4, ... until acceptable.  Returns (rms residual, invalid flag, number of halvings)."""
    st.um1 = st.u
    dtau = dt
    if tread_initially is not None and it < tread_initially['numInitialSteps']:
        dtau = tread_initially['dtau']
    H2, H3, _ = turb_coeffs_1f(st, dx, a_turb, a_prof, theta)
    A, B, C, f = h_to_matrix(dtau, dx, UL, st.um1, H1, H2=H2, H3=H3, H7=H7)
    rms = rms_of([residual(A, B, C, f, st.u)])
    old = st.u
    new = solve(A, B, C, f)
    halvings = 0
    if tread_lightly is not None:
        kw = dict(max_fractional_change=tread_lightly.get('maxFractionalChange', 0.2),
                  inner_fraction=tread_lightly.get('innerFraction', 0.5))
        dtau = dt
        while is_unacceptable(old, new, **kw):
            dtau /= 2
            halvings += 1
            A, B, C, f = h_to_matrix(dtau, dx, UL, old, H1, H2=H2, H3=H3, H7=H7)
            new = solve(A, B, C, f)
    st.u = new
    return rms, profiles_invalid([st.u]), halvings


def run_1f_tread(u0, dt, dx, UL, H1, H7, a_turb, a_prof, tol, max_iter, tread_lightly=None, tread_initially=None,
                 theta=THETA_DEFAULT, solve=tridiag_solve):
    """One timestep of one profile with pseudo-timestep control; returns (profile, iterations, errHistory,
    all iterates (iterations, N), halvings per iteration)."""
    st = State1F(np.asarray(u0, dtype=float))
    err, hist, halv = [], [], []
    it = 0
    done = bad = False
    while it < max_iter and not done and not bad:
        rms, bad, h = picard_step_1f_tread(st, it, dt, dx, UL, H1, H7, a_turb, a_prof, theta, tread_lightly,
                                           tread_initially, solve)
        err.append(rms); hist.append(st.u.copy()); halv.append(h)
        done = bool(rms < tol)
        bad = bool(bad)
        it += 1
    return st.u, it, np.array(err), np.array(hist), np.array(halv)


def run_1f(u0, dt, dx, UL, H1, H7, a_turb, a_prof, tol, max_iter, theta=THETA_DEFAULT,
           flux_fn=None, smooth_window=1, solve=tridiag_solve, fixed_count=False):
    """One timestep of the reference Solver loop for an ensemble of independent one-field
    profiles, each stopped exactly as tango/solver.py:150-154 would stop it when run alone:
    a profile whose residual is below tol still receives that iteration's solve and is then
    frozen.  Returns (profiles, iterations used per profile, errHistory (B, max_iter))."""
    u0 = np.atleast_2d(np.asarray(u0, dtype=float))
    nb = u0.shape[0]
    ULb = np.broadcast_to(np.asarray(UL, dtype=float), (nb,))
    H7b = np.broadcast_to(np.asarray(H7, dtype=float), u0.shape)
    out = np.empty_like(u0)
    iters = np.zeros(nb, dtype=np.int64)
    err = np.zeros((nb, max_iter))
    for b in range(nb):
        st = State1F(u0[b])
        it = 0
        done = False
        bad = False
        while it < max_iter and not done and not bad:
            rms, bad = picard_step_1f(st, dt, dx, ULb[b], H1, H7b[b], a_turb, a_prof, theta,
                                      flux_fn, smooth_window, solve)
            err[b, it] = rms
            done = (not fixed_count) and bool(rms < tol)
            bad = bool(bad)
            it += 1
        out[b] = st.u
        iters[b] = it
    return out, iters, err


# --------------------------------------------------------------------------------------
# one Picard iteration, three fields n, pi<->pe
# (examples/shestakov_threefield_nonlinearnu.py:95-206 on top of the same stack, with
#  tango/JK_to_matrix_fd.py for the pair and tango/fieldgroups.py:275-331 for the grouping)
# --------------------------------------------------------------------------------------
class State3F:
    def __init__(self, n, pi, pe):
        self.f = [State1F(n), State1F(pi), State1F(pe)]


def picard_step_3f(st, dt, dx, BCs, nu0, x, a_turb, a_prof, theta=THETA_DEFAULT,
                   tri=tridiag_solve, pair=pair_solve, turbulence=True, frozen=None):
    """One iteration of the three-field nonlinear-nu problem.  With turbulence=False the
    (H2,H3) of `frozen` are reused and no EWMA memory advances: that is one pass of the
    reference's inner iteration (tango/solver.py:274-309)."""
    fn, fi, fe = st.f
    if turbulence:
        for s in st.f:
            s.uE = ewma_next(s.u, s.uE, a_prof)
        fluxes = threefield_flux(fn.uE, fi.uE, fe.uE, dx)
        frozen = []
        for s, fl in zip(st.f, fluxes):
            D, c, *_ = flux_split(fl, s.uE, dx, theta['Dmin'], theta['Dmax'],
                                  theta['dpdxThreshold'])
            s.DE = ewma_next(D, s.DE, a_turb)
            s.cE = ewma_next(c, s.cE, a_turb)
            frozen.append(dc_to_h(s.DE, s.cE))
    Sn, Si, Se = threefield_sources(x)
    one = np.ones_like(x)
    nu = calc_nu(nu0, fn.u, fi.u, fe.u)
    An, Bn, Cn, Dn = h_to_matrix(dt, dx, BCs[0], fn.um1, one, H2=frozen[0][0], H3=frozen[0][1],
                                 H7=Sn)
    Ueq, Weq = jk_to_matrix(dt, dx, (BCs[1], BCs[2]), (fi.um1, fe.um1),
                            dict(H1=one, H2=frozen[1][0], H3=frozen[1][1], H6=-nu, H7=Si, H8=nu),
                            dict(H1=one, H2=frozen[2][0], H3=frozen[2][1], H6=-nu, H7=Se, H8=nu))
    rn = residual(An, Bn, Cn, Dn, fn.u)
    ri = residual(Ueq[0], Ueq[1], Ueq[2], Ueq[3], fi.u, Ueq[4], fe.u)
    re = residual(Weq[0], Weq[1], Weq[2], Weq[3], fe.u, Weq[4], fi.u)
    rms = rms_of([rn, ri, re])
    fn.u = tri(An, Bn, Cn, Dn)
    fi.u, fe.u = pair(Ueq, Weq)
    return rms, profiles_invalid([fn.u, fi.u, fe.u]), frozen


def picard_step_3f_tread(st, it, dt, dx, BCs, nu0, x, a_turb, a_prof, theta=THETA_DEFAULT, tread_lightly=None,
                         tread_initially=None, tri=tridiag_solve, pair=pair_solve):
    """One iteration of the three-field problem with the pseudo-timestep controls of tango/solver.py:198-235,
    :311-372: the previous iterates take the u_{m-1} slots; a step that moves ANY of the three profiles too far
    (:374-402) is redone from the previous iterates with the same turbulent coefficients, the collision frequency of
    the previous iterates (the callbacks are re-evaluated on them) and dtau = dt/2, dt/4, ...
    Returns (rms residual, invalid flag, number of halvings)."""
    for s in st.f:
        s.um1 = s.u
    dtau = dt
    if tread_initially is not None and it < tread_initially['numInitialSteps']:
        dtau = tread_initially['dtau']
    old = [s.u for s in st.f]
    rms, bad, frozen = picard_step_3f(st, dtau, dx, BCs, nu0, x, a_turb, a_prof, theta, tri, pair)
    halvings = 0
    if tread_lightly is not None:
        kw = dict(max_fractional_change=tread_lightly.get('maxFractionalChange', 0.2),
                  inner_fraction=tread_lightly.get('innerFraction', 0.5))
        far = lambda: any(is_unacceptable(o, s.u, **kw) for o, s in zip(old, st.f))
        dtau = dt
        while far():
            dtau /= 2
            halvings += 1
            for s, o in zip(st.f, old):
                s.u = o
            _, bad, _ = picard_step_3f(st, dtau, dx, BCs, nu0, x, a_turb, a_prof, theta, tri, pair, turbulence=False, frozen=frozen)
    return rms, bad, halvings


def run_3f_tread(n0, pi0, pe0, dt, dx, BCs, nu0, x, a_turb, a_prof, tol, max_iter, tread_lightly=None, tread_initially=None,
                 theta=THETA_DEFAULT):
    """One timestep of one three-field profile with pseudo-timestep control; returns ((n, pi, pe), iterations, errHistory,
    all iterates (iterations, 3, N), halvings per iteration)."""
    st = State3F(n0, pi0, pe0)
    err, hist, halv = [], [], []
    it = 0
    done = bad = False
    while it < max_iter and not done and not bad:
        rms, bad, h = picard_step_3f_tread(st, it, dt, dx, BCs, nu0, x, a_turb, a_prof, theta, tread_lightly, tread_initially)
        err.append(rms); hist.append(np.stack([s.u for s in st.f])); halv.append(h)
        done = bool(rms < tol)
        bad = bool(bad)
        it += 1
    return tuple(s.u for s in st.f), it, np.array(err), np.array(hist), np.array(halv)


def run_3f(n0, pi0, pe0, dt, dx, BCs, nu0, x, a_turb, a_prof, tol, max_iter,
           theta=THETA_DEFAULT, inner=0, tri=tridiag_solve, pair=pair_solve,
           fixed_count=False):
    """Reference Solver loop for one three-field profile; `inner` > 0 adds that many inner
    iterations with frozen turbulent coefficients after each outer solve.  Returns
    ((n, pi, pe), iterations, errHistory)."""
    st = State3F(n0, pi0, pe0)
    err = np.zeros(max_iter)
    it = 0
    done = bad = False
    while it < max_iter and not done and not bad:
        rms, bad, frozen = picard_step_3f(st, dt, dx, BCs, nu0, x, a_turb, a_prof, theta,
                                          tri, pair)
        for _ in range(inner):
            picard_step_3f(st, dt, dx, BCs, nu0, x, a_turb, a_prof, theta, tri, pair,
                           turbulence=False, frozen=frozen)
        if inner:
            bad = profiles_invalid([s.u for s in st.f])
        err[it] = rms
        done = (not fixed_count) and bool(rms < tol)
        bad = bool(bad)
        it += 1
    return tuple(s.u for s in st.f), it, err


# --------------------------------------------------------------------------------------
# grid mapping (tango/interfacegrids_gene.py:35-87, :446-484)
# --------------------------------------------------------------------------------------
def spline_resample(xin, yin, xout):
    """Cubic interpolating spline with not-a-knot ends through (xin, yin) at xout, in the second-derivative form:
    the curve scipy.interpolate.InterpolatedUnivariateSpline (FITPACK, k = 3; SciPy is unpinned by the reference,
    1.15.3 here) returns for the same points.  1-D arrays; dense solve."""
    x, y = np.asarray(xin, dtype=float), np.asarray(yin, dtype=float)
    n = len(x)
    h = np.diff(x)
    T = np.zeros((n, n)); r = np.zeros(n)
    for i in range(1, n - 1):
        T[i, i - 1], T[i, i], T[i, i + 1] = h[i - 1], 2 * (h[i - 1] + h[i]), h[i]
        r[i] = 6 * ((y[i + 1] - y[i]) / h[i] - (y[i] - y[i - 1]) / h[i - 1])
    T[0, 0], T[0, 1], T[0, 2] = h[1], -(h[0] + h[1]), h[0]
    T[-1, -1], T[-1, -2], T[-1, -3] = h[-2], -(h[-1] + h[-2]), h[-1]
    M = np.linalg.solve(T, r)
    xo = np.asarray(xout, dtype=float)
    i = np.clip(np.searchsorted(x, xo, side='right') - 1, 0, n - 2)
    hi = h[i]; dl = xo - x[i]; dr = x[i + 1] - xo
    return ((M[i] * dr ** 3 + M[i + 1] * dl ** 3) / (6 * hi) + (y[i] / hi - M[i] * hi / 6) * dr
            + (y[i + 1] / hi - M[i + 1] * hi / 6) * dl)


def extend_with_zeros_both_sides(xSmall, fSmall, xLarge, enforce_positive=False):
    """tango/interfacegrids_gene.py:446-484."""
    xS, xL = np.asarray(xSmall, dtype=float), np.asarray(xLarge, dtype=float)
    out = np.zeros_like(xL)
    lo = np.where(xL > xS[0])[0][0]
    hi = np.where(xL < xS[-1])[0][-1]
    out[lo:hi + 1] = spline_resample(xS, fSmall, xL[lo:hi + 1])
    if enforce_positive:
        out[out < 0] = 0
    return out


def extend_with_zeros_left_side(xIn, fIn, xOut, enforce_positive=False):
    """tango/interfacegrids_gene.py:486-526."""
    xI, xO = np.asarray(xIn, dtype=float), np.asarray(xOut, dtype=float)
    out = np.zeros_like(xO)
    lo = np.where(xO > xI[0])[0][0]
    out[lo:] = spline_resample(xI, fIn, xO[lo:])
    if enforce_positive:
        out[out < 0] = 0
    return out


def extrap_on_left_and_right(xIn, fIn, xOut, inL, inR, outL, outR, degree, enforce_positive=False):
    """tango/interfacegrids_gene.py:300-364: spline up to outR, fitted polynomials left of inL and right of outR."""
    xI, fI, xO = np.asarray(xIn, dtype=float), np.asarray(fIn, dtype=float), np.asarray(xOut, dtype=float)
    out = np.zeros_like(xO)
    ind = (xO >= xI[0]) & (xO <= outR)
    out[ind] = spline_resample(xI, fI, xO[ind])
    z = (xI >= inL) & (xI <= inR)
    out[xO < inL] = np.polyval(np.polyfit(xI[z], fI[z], degree), xO[xO < inL])
    z = (xI >= outL) & (xI <= outR)
    out[xO > outR] = np.polyval(np.polyfit(xI[z], fI[z], degree), xO[xO > outR])
    if enforce_positive:
        out[out < 0] = 0
    return out


def constrained_regression_line(x, y, xEval, side='right', numPts=10):
    """tango/interfacegrids_gene.py:528-576: line through the end point with the least-squares slope of numPts points."""
    x, y = np.asarray(x, dtype=float), np.asarray(y, dtype=float)
    xs, ys, x0, y0 = (x[:numPts], y[:numPts], x[0], y[0]) if side == 'left' else (x[-numPts:], y[-numPts:], x[-1], y[-1])
    a = np.dot(xs - x0, ys - y0) / np.dot(xs - x0, xs - x0)
    return a * (np.asarray(xEval, dtype=float) - x0) + y0


def truncate_on_left_extrapolate_on_right(xIn, fIn, xOut, numPts=10, enforce_positive=False, side='right'):
    """tango/interfacegrids_gene.py:579-634: the regression line, replaced by the spline strictly inside xIn."""
    xI, xO = np.asarray(xIn, dtype=float), np.asarray(xOut, dtype=float)
    out = constrained_regression_line(xI, fIn, xO, side, numPts)
    ind = (xO > xI[0]) & (xO < xI[-1])
    out[ind] = spline_resample(xI, fIn, xO[ind])
    if enforce_positive:
        out[out < 0] = 0
    return out


def truncate_on_left_fixed_slope_on_right(xIn, fIn, xOut, slope, enforce_positive=False):
    """tango/interfacegrids_gene.py:636-692: imposed slope outside xIn, spline on its closed range."""
    xI, fI, xO = np.asarray(xIn, dtype=float), np.asarray(fIn, dtype=float), np.asarray(xOut, dtype=float)
    out = np.zeros_like(xO)
    out[xO < xI[0]] = slope * (xO[xO < xI[0]] - xI[0]) + fI[0]
    out[xO > xI[-1]] = slope * (xO[xO > xI[-1]] - xI[-1]) + fI[-1]
    ind = (xO >= xI[0]) & (xO <= xI[-1])
    out[ind] = spline_resample(xI, fI, xO[ind])
    if enforce_positive:
        out[out < 0] = 0
    return out


# --------------------------------------------------------------------------------------
# synthetic ensembles of the benchmark configurations
# --------------------------------------------------------------------------------------
def shestakov_ensemble(nb, n, seed=12345):
    """Seeded one-field ensemble: member b has right boundary value nL, source amplitude S0
    and width delta, and initial profile a - s*x; member 0 is the exact
    examples/shestakov_basic.py problem (nL=1e-2, S0=1, delta=0.1, n = 1 - 0.5 x)."""
    rng = np.random.default_rng(seed)
    nL = rng.uniform(0.005, 0.02, size=nb)
    S0 = rng.uniform(0.5, 2.0, size=nb)
    delta = rng.uniform(0.05, 0.2, size=nb)
    a = rng.uniform(0.8, 1.2, size=nb)
    s = rng.uniform(0.3, 0.6, size=nb) * a
    nL[0], S0[0], delta[0], a[0], s[0] = 1e-2, 1.0, 0.1, 1.0, 0.5
    dx = 1.0 / (n - 1)
    x = np.arange(n) * dx
    u0 = a[:, None] - s[:, None] * x[None, :]
    return dict(x=x, dx=dx, u0=u0, nL=nL, S0=S0, delta=delta)


def threefield_ensemble(nb, n, seed=2024):
    """Seeded three-field ensemble: nu0 in [1,4] and the three right boundary values within
    +-25 % of (2, 0.1, 0.3); member 0 is the exact example problem (nu0 = 2.2)."""
    rng = np.random.default_rng(seed)
    nu0 = rng.uniform(1.0, 4.0, size=nb)
    bc = np.array([2.0, 0.1, 0.3])[None, :] * rng.uniform(0.75, 1.25, size=(nb, 3))
    nu0[0] = 2.2
    bc[0] = (2.0, 0.1, 0.3)
    dx = 1.0 / (n - 1)
    x = np.arange(n) * dx
    u0 = np.broadcast_to(1 - 0.5 * x, (nb, n)).copy()
    return dict(x=x, dx=dx, u0=u0, nu0=nu0, bc=bc)
