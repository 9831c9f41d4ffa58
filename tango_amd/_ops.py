"""Array-level operations of the transport inner loop, each one a call into libtango_hip.so.

Every function takes NumPy arrays (1-D = one profile, as in the reference, or 2-D (profiles, N))
or device arrays (tango_amd.DeviceArray / CUDA torch tensors, always 2-D) and returns the same kind:
NumPy in -> NumPy out with the input's dimensionality (host -> device -> kernel -> host), device
in -> device out with no host traffic.  Inputs are never modified.  There is no CPU code path."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _hip as H
from ._hip import DeviceArray, check, lib, ptr_of


class _Ctx:
    """Moves a call's inputs to the device and remembers how results must be returned."""

    def __init__(self):
        self.any_device = False
        self.ndim = None
        self.nb = None
        self.n = None
        self.keep = []

    def arr(self, a, name='array'):
        """A per-point array: (n,) / (nb, n) host or (nb, n) device -> device pointer holder."""
        if H.is_device(a):
            self.any_device = True
            shape = tuple(a.shape)
            if len(shape) != 2:
                raise AssertionError(f'{name}: device arrays must be 2-D (profiles, N)')
            self._dims(shape[0], shape[1], 2, name)
            return a
        a = np.asarray(a, dtype=np.float64)
        if a.ndim not in (1, 2):
            raise AssertionError(f'{name}: expected a 1-D or 2-D array')
        shape = (1, a.shape[0]) if a.ndim == 1 else a.shape
        self._dims(shape[0], shape[1], a.ndim, name)
        d = DeviceArray.from_host(a.reshape(shape))
        self.keep.append(d)
        return d

    def _dims(self, nb, n, ndim, name):
        if self.n is None:
            self.n = n
        elif n != self.n:
            raise AssertionError(f'{name}: lengths must be equal')
        if self.nb is None or self.nb == 1:
            self.nb = nb if self.nb is None else max(self.nb, nb)
        elif nb not in (1, self.nb):
            raise AssertionError(f'{name}: number of profiles does not match')
        self.ndim = ndim if self.ndim is None else max(self.ndim, ndim)

    def coeff(self, a, name='coefficient'):
        """A coefficient that may be shared by all profiles: returns (device array or None, stride)."""
        if a is None:
            return None, 0
        d = self.arr(a, name)
        return d, (0 if d.shape[0] == 1 else self.n)

    def full(self, d):
        """Expand a (1, n) device array to (nb, n) when other arguments are batched."""
        if d.shape[0] == self.nb:
            return d
        host = np.broadcast_to(d.to_host(), (self.nb, self.n))
        e = DeviceArray.from_host(host)
        self.keep.append(e)
        return e

    def per_profile(self, v, name='scalar'):
        """A per-profile scalar: Python/NumPy scalar or (nb,) array -> (nb,) device array."""
        if H.is_device(v):
            self.any_device = True
            return v
        v = np.asarray(v, dtype=np.float64)
        if v.ndim == 0:
            v = np.full((self.nb,), float(v))
        elif v.shape != (self.nb,):
            raise AssertionError(f'{name}: expected a scalar or one value per profile')
        d = DeviceArray.from_host(v)
        self.keep.append(d)
        return d

    def new(self, shape=None, dtype=np.float64, zero=False):
        return DeviceArray(shape or (self.nb, self.n), dtype, zero=zero)

    def out(self, *ds):
        """Return results as the caller's kind of array."""
        if self.any_device:
            res = ds
        else:
            res = tuple(None if d is None else (d.to_host()[0] if self.ndim == 1 else d.to_host()) for d in ds)
        return res[0] if len(res) == 1 else res


def _raise_singular(status):
    if status is not None and np.any(status.to_host() & H.ST_SINGULAR):
        raise np.linalg.LinAlgError('singular matrix')


def tridiag_solve(A, B, Cc, D, layout=H.POINT_FASTEST, check_singular=True):
    """tango/tridiag.py:12-31.  With layout=BATCH_FASTEST (dgtsv's sweeps, any N) or BATCH_FASTEST_TILED (the
    point-fastest algorithm on interleaved arrays, N <= 512) the (device) arrays are (N, profiles)."""
    cx = _Ctx()
    if layout in (H.BATCH_FASTEST, H.BATCH_FASTEST_TILED):
        if not all(H.is_device(v) for v in (A, B, Cc, D)):
            raise AssertionError('batch-fastest layout takes device arrays of shape (N, profiles)')
        n, nb = A.shape
        u = DeviceArray((n, nb))
        wb = lib.tg_tridiag_work_bytes(nb, n, layout)
        work = DeviceArray((max(wb // 8, 1),))
        status = DeviceArray((nb,), np.int32, zero=True)
        check(lib.tg_tridiag_solve(ptr_of(A), ptr_of(B), ptr_of(Cc), ptr_of(D), u.ptr, nb, n, layout, status.ptr,
                                   work.ptr, wb, H.stream()))
        if check_singular:
            _raise_singular(status)
        return u
    dA, dB, dC, dD = (cx.arr(v, k) for v, k in ((A, 'A'), (B, 'B'), (Cc, 'C'), (D, 'D')))
    dA, dB, dC, dD = (cx.full(v) for v in (dA, dB, dC, dD))
    u = cx.new()
    status = cx.new((cx.nb,), np.int32, zero=True)
    check(lib.tg_tridiag_solve(ptr_of(dA), ptr_of(dB), ptr_of(dC), ptr_of(dD), u.ptr, cx.nb, cx.n, layout,
                               status.ptr, None, 0, H.stream()))
    if check_singular:
        _raise_singular(status)
    return cx.out(u)


def h_to_matrix(dt, dx, UL, U_mminus1, H1, H2=None, H3=None, H4=None, H6=None, H7=None, H8=None, want_G=False):
    """tango/HToMatrixFD.py:53-117 (and the coupling column of tango/JK_to_matrix_fd.py:119-130)."""
    cx = _Ctx()
    dU = cx.arr(U_mminus1, 'U_mminus1')
    hs = HCoeffsHolder(cx, H1, H2, H3, H4, H6, H7, H8)
    dU = cx.full(dU)
    dUL = cx.per_profile(UL, 'UL')
    dt_arr = None
    if np.ndim(dt) > 0 or H.is_device(dt):
        dt_arr = cx.per_profile(dt, 'dt')
        dt = 0.0
    A, B, Cc, D = cx.new(), cx.new(), cx.new(), cx.new()
    G = cx.new() if want_G else None
    check(lib.tg_h_to_matrix(float(dt), ptr_of(dt_arr), float(dx), ptr_of(dUL), ptr_of(dU), C.byref(hs.c), A.ptr, B.ptr,
                             Cc.ptr, D.ptr, ptr_of(G), cx.nb, cx.n, H.stream()))
    return cx.out(A, B, Cc, D, G) if want_G else cx.out(A, B, Cc, D)


class HCoeffsHolder:
    def __init__(self, cx, H1, H2, H3, H4, H6, H7, H8):
        assert H1 is not None, 'H1 is required'
        self.c = H.HCoeffsC()
        self.keep = []
        for name, idx, val in (('H1', 's1', H1), ('H2', 's2', H2), ('H3', 's3', H3), ('H4', 's4', H4),
                               ('H6', 's6', H6), ('H7', 's7', H7), ('H8', 's8', H8)):
            d, stride = cx.coeff(val, name)
            self.keep.append(d)
            setattr(self.c, name, ptr_of(d))
            setattr(self.c, idx, stride)


def residual(A, B, Cc, f, u, G=None, w=None):
    """tango/fieldgroups.py:139-140 / :247-251.  Returns (normalised residual, sum of its squares)."""
    cx = _Ctx()
    ds = [cx.arr(v, k) for v, k in ((A, 'A'), (B, 'B'), (Cc, 'C'), (f, 'f'), (u, 'u'))]
    dG = dw = None
    if G is not None:
        dG, dw = cx.arr(G, 'G'), cx.arr(w, 'w')
        dG, dw = cx.full(dG), cx.full(dw)
    ds = [cx.full(d) for d in ds]
    r = cx.new()
    ss = cx.new((cx.nb,))
    check(lib.tg_residual(*[ptr_of(d) for d in ds], ptr_of(dG), ptr_of(dw), r.ptr, ss.ptr, cx.nb, cx.n, H.stream()))
    if cx.any_device:
        return r, ss
    rs, s = r.to_host(), ss.to_host()
    return (rs[0], float(s[0])) if cx.ndim == 1 else (rs, s)


def ewma(y, y_prev, a):
    """tango/lodestro_method.py:706-722: a*y + (1-a)*y_prev, y_prev=None seeding with y."""
    cx = _Ctx()
    dy = cx.arr(y, 'y')
    out = cx.new((dy.shape[0], cx.n))
    seeded = y_prev is not None
    if seeded:
        dp = cx.arr(y_prev, 'y_prev')
        out.copy_from(cx.full(dp) if dp.shape[0] != dy.shape[0] else dp)
    check(lib.tg_ewma(ptr_of(dy), out.ptr, float(a), int(seeded), dy.shape[0] * cx.n, H.stream()))
    return cx.out(out)


def flux_split(flux, p, dx, Dmin, Dmax, thr, gxx=None, gradx=None):
    """tango/lodestro_method.py:624-638.  Returns (D, c, DHat, cHat, theta)."""
    cx = _Ctx()
    df, dp = cx.arr(flux, 'flux'), cx.arr(p, 'p')
    df, dp = cx.full(df), cx.full(dp)
    g1 = _shared_n(cx, gxx, 'gxxAvg')
    g2 = _shared_n(cx, gradx, 'gradxAvg')
    outs = [cx.new() for _ in range(5)]
    check(lib.tg_flux_split(ptr_of(df), ptr_of(dp), float(dx), float(Dmin), float(Dmax), float(thr), ptr_of(g1),
                            ptr_of(g2), *[o.ptr for o in outs], cx.nb, cx.n, H.stream()))
    return cx.out(*outs)


def _shared_n(cx, g, name):
    """Geometry factor: None / scalar 1 -> NULL; otherwise an (n,) array shared by all profiles."""
    if g is None or (np.ndim(g) == 0 and float(g) == 1.0):
        return None
    g = np.broadcast_to(np.asarray(g, dtype=np.float64), (cx.n,)) if np.ndim(g) == 0 else np.asarray(g, dtype=np.float64)
    if g.shape != (cx.n,):
        raise AssertionError(f'{name}: expected a scalar or an array of length N')
    d = DeviceArray.from_host(g)
    cx.keep.append(d)
    return d


def ddx(u, dx, edge_order=1):
    """tango/derivatives.py:9-24 (edge_order=1) and :26-41 (edge_order=2)."""
    cx = _Ctx()
    du = cx.arr(u, 'u')
    out = cx.new((du.shape[0], cx.n))
    check(lib.tg_ddx(ptr_of(du), float(dx), int(edge_order), out.ptr, du.shape[0], cx.n, H.stream()))
    return cx.out(out)


def moving_average(x, window):
    """tango/smoother.py:13-48 (window already odd)."""
    if window == 1:
        return x
    assert window % 2 == 1, 'windowSize must be an odd integer.'
    cx = _Ctx()
    dxx = cx.arr(x, 'x')
    out = cx.new((dxx.shape[0], cx.n))
    check(lib.tg_moving_average(ptr_of(dxx), int(window), out.ptr, dxx.shape[0], cx.n, H.stream()))
    return cx.out(out)


def spline_resample(xin, yin, xout, lo=0, hi=None, nonneg=False):
    """Not-a-knot cubic spline through (xin, yin rows) evaluated at xout[lo:hi], zeros elsewhere (the curve of
    scipy.interpolate.InterpolatedUnivariateSpline as used by tango/interfacegrids_gene.py:51-57, :446-484)."""
    xin = np.ascontiguousarray(xin, dtype=np.float64)
    xout = np.ascontiguousarray(xout, dtype=np.float64)
    assert xin.ndim == 1 and xout.ndim == 1 and xin.size >= 4, 'grids must be 1-D with at least 4 points'
    assert np.all(np.diff(xin) > 0), 'the input grid must be strictly increasing'
    hi = xout.size if hi is None else hi
    cx = _Ctx()
    dy = cx.arr(yin, 'values')
    assert cx.n == xin.size, 'values and grid must have the same length'
    nb = dy.shape[0]
    dxin, dxout = DeviceArray.from_host(xin), DeviceArray.from_host(xout)
    out = cx.new((nb, xout.size))
    wb = int(lib.tg_spline_work_bytes(nb, xin.size))
    work = DeviceArray((wb // 8,))
    check(lib.tg_spline_resample(dxin.ptr, ptr_of(dy), nb, xin.size, dxout.ptr, xout.size, int(lo), int(hi), int(bool(nonneg)),
                                 out.ptr, work.ptr, wb, H.stream()))
    return cx.out(out)


def shestakov_flux(n_prof, dx):
    """tango/extras/shestakov_nonlinear_diffusion.py:51-61."""
    cx = _Ctx()
    dn = cx.arr(n_prof, 'n')
    out = cx.new((dn.shape[0], cx.n))
    check(lib.tg_shestakov_flux(ptr_of(dn), float(dx), out.ptr, dn.shape[0], cx.n, H.stream()))
    return cx.out(out)


def pair_solve(U, W, check_singular=True):
    """tango/JK_to_matrix_fd.py:132-166.  U = (AU,BU,CU,DU,GW), W = (AW,BW,CW,DW,GU)."""
    cx = _Ctx()
    dU = [cx.arr(v, 'U coefficient') for v in U]
    dW = [cx.arr(v, 'W coefficient') for v in W]
    dU = [cx.full(v) for v in dU]
    dW = [cx.full(v) for v in dW]
    u, w = cx.new(), cx.new()
    status = cx.new((cx.nb,), np.int32, zero=True)
    check(lib.tg_pair_solve(*[ptr_of(v) for v in dU], *[ptr_of(v) for v in dW], u.ptr, w.ptr, cx.nb, cx.n, status.ptr,
                            H.stream()))
    if check_singular:
        _raise_singular(status)
    return cx.out(u, w)


def block3_solve(Cb, Bb, Ab, d, check_singular=True):
    """3x3 block-tridiagonal solve.  Host arrays: blocks (nb, n, 3, 3) or (n, 3, 3), rhs (nb, n, 3) or (n, 3);
    returns x with the shape of d."""
    Bb = np.asarray(Bb, dtype=np.float64)
    single = Bb.ndim == 3
    def comp_major(a, k):
        a = np.asarray(a, dtype=np.float64)
        if single:
            a = a[None]
        nb, n = a.shape[0], a.shape[1]
        return np.ascontiguousarray(np.moveaxis(a.reshape(nb, n, k), 2, 0)), nb, n
    c, nb, n = comp_major(Cb, 9)
    b, _, _ = comp_major(Bb, 9)
    a, _, _ = comp_major(Ab, 9)
    r, _, _ = comp_major(d, 3)
    dc, db, da, dr = (DeviceArray.from_host(v) for v in (c, b, a, r))
    x = DeviceArray((3, nb, n))
    status = DeviceArray((nb,), np.int32, zero=True)
    check(lib.tg_block3_solve(dc.ptr, db.ptr, da.ptr, dr.ptr, x.ptr, nb, n, status.ptr, H.stream()))
    if check_singular:
        _raise_singular(status)
    out = np.moveaxis(x.to_host(), 0, 2)
    return out[0] if single else out
