"""Seeded synthetic ensembles of the benchmark configurations and a thin device-resident driver for the
fused one-field step (used by bench.py, the smoke test and the multi-GPU path)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _hip as H

THETA_DEFAULT = dict(Dmin=1e-5, Dmax=1e13, dpdxThreshold=10.0)


def shard_range(nb, rank, world):
    """Contiguous range [lo, hi) of profiles owned by `rank` out of `world` ranks."""
    base, extra = divmod(nb, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shestakov_ensemble(nb, n, seed=12345, lo=0, hi=None):
    """One-field Shestakov ensemble: member b has right boundary value nL in [0.005, 0.02], source amplitude
    S0 in [0.5, 2] on x < delta in [0.05, 0.2] and the initial profile a - s x with a in [0.8, 1.2],
    s in [0.3, 0.6] a; member 0 is the problem of examples/shestakov_basic.py:18-39 (nL = 1e-2, S0 = 1,
    delta = 0.1, n = 1 - 0.5 x).  Drawn with numpy.random.default_rng(seed) in the order nL, S0, delta, a, s
    for the whole ensemble; [lo, hi) selects a shard."""
    hi = nb if hi is None else hi
    rng = np.random.default_rng(seed)
    nL = rng.uniform(0.005, 0.02, size=nb)
    S0 = rng.uniform(0.5, 2.0, size=nb)
    delta = rng.uniform(0.05, 0.2, size=nb)
    a = rng.uniform(0.8, 1.2, size=nb)
    s = rng.uniform(0.3, 0.6, size=nb) * a
    nL[0], S0[0], delta[0], a[0], s[0] = 1e-2, 1.0, 0.1, 1.0, 0.5
    dx = 1.0 / (n - 1)
    x = np.arange(n) * dx
    sl = slice(lo, hi)
    u0 = a[sl, None] - s[sl, None] * x[None, :]
    return dict(x=x, dx=dx, u0=u0, nL=nL[sl].copy(), S0=S0[sl].copy(), delta=delta[sl].copy())


def threefield_ensemble(nb, n, seed=2024, lo=0, hi=None):
    """Three-field ensemble around examples/shestakov_threefield_nonlinearnu.py:159-206: collision amplitude nu0
    in [1, 4] and the right boundary values of (n, pi, pe) within 25 % of (2, 0.1, 0.3), initial profiles
    1 - 0.5 x for all three fields; member 0 is the example itself (nu0 = 2.2).  Drawn with
    numpy.random.default_rng(seed) in the order nu0, boundary factors; [lo, hi) selects a shard."""
    hi = nb if hi is None else hi
    rng = np.random.default_rng(seed)
    nu0 = rng.uniform(1.0, 4.0, size=nb)
    bc = np.array([2.0, 0.1, 0.3])[None, :] * rng.uniform(0.75, 1.25, size=(nb, 3))
    nu0[0] = 2.2
    bc[0] = (2.0, 0.1, 0.3)
    dx = 1.0 / (n - 1)
    x = np.arange(n) * dx
    u0 = np.broadcast_to(1 - 0.5 * x, (hi - lo, n)).copy()
    return dict(x=x, dx=dx, u0=u0, nu0=nu0[lo:hi].copy(), bc=bc[lo:hi].copy())


def source_cut(x, delta):
    """Number of grid points with x < delta for every profile (x increasing): the box source S0*[x < delta] of
    tango/extras/shestakov_nonlinear_diffusion.py:63-68 is then S0 for j < cut."""
    x = np.asarray(x, dtype=np.float64)
    if np.any(np.diff(x) <= 0):
        raise ValueError('the radial grid must be strictly increasing')
    return np.searchsorted(x, np.atleast_1d(np.asarray(delta, dtype=np.float64)), side='left').astype(np.int32)


class DeviceView:
    """A typed window into device memory owned by someone else (a pool)."""

    def __init__(self, ptr, shape, dtype, owner):
        self.ptr, self.shape, self.dtype, self.owner = ptr, tuple(shape), np.dtype(dtype), owner
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize

    def data_ptr(self):
        return self.ptr

    def set(self, a):
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=self.dtype), self.shape))
        H.check(H.lib.tg_copy_h2d(self.ptr, a.ctypes.data, self.nbytes, H.stream()))
        H.check(H.lib.tg_stream_sync(H.stream()))

    def zero(self):
        H.check(H.lib.tg_memset(self.ptr, 0, self.nbytes, H.stream()))

    def to_host(self):
        out = np.empty(self.shape, self.dtype)
        H.check(H.lib.tg_copy_d2h(out.ctypes.data, self.ptr, self.nbytes, H.stream()))
        H.check(H.lib.tg_stream_sync(H.stream()))
        return out


# Consecutive state arrays are offset by this many extra bytes.  Arrays of 2^k bytes allocated separately all
# start at the same address modulo every power of two, so the five loads and four stores a wavefront issues for
# one profile hit the same HBM channel and bank; a small odd multiple of 256 B between arrays spreads them
# (measured on MI355X: 0.65 ms -> 0.58 ms per iteration at 65 536 x 512).
POOL_SKEW_BYTES = 17 * 256


class Pool:
    """Carves 256-byte-aligned, skewed views out of one device allocation."""

    def __init__(self, nbytes, alloc=None):
        self.base = alloc(nbytes) if alloc else H.DeviceArray(((nbytes + 7) // 8,))
        self.ptr = H.ptr_of(self.base)
        self.nbytes = nbytes
        self.off = (-self.ptr) % 256

    @staticmethod
    def size_for(shapes_dtypes):
        tot = 256
        for shape, dtype in shapes_dtypes:
            tot += int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize + POOL_SKEW_BYTES + 256
        return tot

    def take(self, shape, dtype=np.float64):
        v = DeviceView(self.ptr + self.off, shape, dtype, self)
        self.off += v.nbytes + POOL_SKEW_BYTES
        self.off += (-self.off - self.ptr) % 256
        assert self.off <= self.nbytes, 'pool exhausted'
        return v


class OneFieldEnsemble:
    """Device-resident state of a one-field ensemble plus the parameter block of tg_picard_1f_step.

    All arrays live in one pool allocation (skewed offsets, see POOL_SKEW_BYTES).  alloc(nbytes) -> object with
    data_ptr() lets a caller own the memory (bench.py passes a torch allocator); default: tg_malloc."""

    def __init__(self, ens, dt=1e4, a_turb=0.3, a_prof=1.0, tol=1e-11, theta=THETA_DEFAULT, max_iter=1000,
                 fixed_count=False, keep_err=True, alloc=None, segmented=True, tread=False):
        u0 = np.ascontiguousarray(ens['u0'], dtype=np.float64)
        self.nb, self.n = nb, n = u0.shape
        f8, i4, i8 = np.float64, np.int32, np.int64
        plan = [((nb, n), f8)] * 5 + [((nb,), i4)] * 3 + [((nb,), f8)] * 2 + [((4,), i8)]
        if keep_err:
            plan.append(((nb, max_iter), f8))
        # work space of the segmented form of the step (profiles of 4096 points and more), 0 bytes otherwise
        # (tread: with room for the previous iterate, for steps under pseudo-timestep control)
        work_bytes = int((H.lib.tg_picard_1f_work_bytes_tread if tread else H.lib.tg_picard_1f_work_bytes)(nb, n)) if segmented else 0
        if work_bytes:
            plan.append(((work_bytes // 8,), f8))
        self.pool = pool = Pool(Pool.size_for(plan), alloc)
        self.u, self.um1, self.uE, self.DE, self.cE = (pool.take((nb, n)) for _ in range(5))
        self.status, self.iters = pool.take((nb,), i4), pool.take((nb,), i4)
        self.UL, self.S0 = pool.take((nb,)), pool.take((nb,))
        self.src_cut = pool.take((nb,), i4)
        self.counts = pool.take((4,), i8)
        self.err = pool.take((nb, max_iter)) if keep_err else None
        self.work = pool.take((work_bytes // 8,)) if work_bytes else None
        self.u.set(u0)
        self.um1.set(u0)
        for v in (self.uE, self.DE, self.cE, self.status, self.iters, self.counts) + ((self.err,) if keep_err else ()):
            v.zero()
        self.UL.set(ens['nL'])
        self.S0.set(ens['S0'])
        self.src_cut.set(source_cut(ens['x'], ens['delta']))
        p = H.ptr_of
        self.state = H.State1F(nb, n, p(self.u), p(self.um1), p(self.uE), p(self.DE), p(self.cE),
                               p(self.status), p(self.iters), p(self.err) if keep_err else None,
                               max_iter if keep_err else 0)
        q = H.Params1F()
        q.dt, q.dx, q.a_turb, q.a_prof = float(dt), float(ens['dx']), float(a_turb), float(a_prof)
        q.Dmin, q.Dmax, q.dpdx_thr, q.tol = float(theta['Dmin']), float(theta['Dmax']), float(theta['dpdxThreshold']), float(tol)
        q.UL, q.S0, q.src_cut = p(self.UL), p(self.S0), p(self.src_cut)
        q.flux_model, q.smooth_window, q.fixed_count, q.frozen_turb = H.FLUX_SHESTAKOV, 1, int(fixed_count), 0
        if self.work is not None:
            q.work, q.work_bytes = p(self.work), work_bytes
        self.params = q
        self.max_iter = max_iter

    def step(self, stream=None, iters=1):
        """`iters` Picard iterations of every active profile in one launch (asynchronous)."""
        self.params.iters_per_launch = int(iters)
        H.check(H.lib.tg_picard_1f_step(C.byref(self.state), C.byref(self.params), stream if stream is not None else H.stream()))

    def reduce_status(self, stream=None, counts_ptr=None):
        """Launch the status reduction; (active, converged, invalid, singular) land in self.counts (or at
        counts_ptr, 4 int64 on the device)."""
        H.check(H.lib.tg_status_reduce(H.ptr_of(self.status), self.nb, counts_ptr or H.ptr_of(self.counts),
                                       stream if stream is not None else H.stream()))

    def reduce_residual(self, out_ptr, stream=None):
        """Launch the residual summary of tg_residual_reduce into 5 device doubles at out_ptr (needs keep_err)."""
        if self.err is None:
            raise ValueError('the residual summary needs the residual history (keep_err=True)')
        H.check(H.lib.tg_residual_reduce(H.ptr_of(self.status), H.ptr_of(self.iters), H.ptr_of(self.err), self.state.err_stride,
                                         self.nb, self.n, out_ptr, stream if stream is not None else H.stream()))


def device_arrays(shapes_dtypes, zero=False):
    """One tango_amd.DeviceArray per (shape, dtype), all carved from ONE allocation with skewed offsets (see
    POOL_SKEW_BYTES): separately allocated power-of-two arrays would all start on the same HBM channel."""
    pool = Pool(Pool.size_for(shapes_dtypes))
    out = []
    for shape, dtype in shapes_dtypes:
        v = pool.take(shape, dtype)
        out.append(H.DeviceArray(shape, dtype, zero=zero, at=v.ptr, owner=pool))
    return out


def pooled(shapes_dtypes, alloc=None):
    """One skewed pool holding an array for every (shape, dtype) of the list; returns (pool, views)."""
    pool = Pool(Pool.size_for(shapes_dtypes), alloc)
    return pool, [pool.take(shape, dtype) for shape, dtype in shapes_dtypes]


class ThreeFieldEnsemble:
    """Device-resident state of a three-field (n, pi, pe) ensemble plus the parameter block of tg_picard_3f_step
    (the problem of examples/shestakov_threefield_nonlinearnu.py:52-206, one collision amplitude nu0 and one set
    of right boundary values per profile).  The fifteen state arrays live in one pool with skewed offsets."""

    def __init__(self, ens, dt=1e6, a_turb=0.3, a_prof=0.3, tol=1e-9, theta=THETA_DEFAULT, max_iter=400,
                 fixed_count=False, keep_err=True, alloc=None):
        from .extras.threefield import source_e, source_i, source_n
        u0 = np.ascontiguousarray(ens['u0'], dtype=np.float64)
        self.nb, self.n = nb, n = u0.shape
        f8, i4, i8 = np.float64, np.int32, np.int64
        plan = [((nb, n), f8)] * 15 + [((nb,), i4)] * 2 + [((nb,), f8)] * 4 + [((n,), f8)] * 3 + [((4,), i8)]
        if keep_err:
            plan.append(((nb, max_iter), f8))
        self.pool, v = pooled(plan, alloc)
        self.u, self.um1, self.uE, self.DE, self.cE = (v[3 * k:3 * k + 3] for k in range(5))
        self.status, self.iters = v[15], v[16]
        self.UL, self.nu0, self.H7, self.counts = v[17:20], v[20], v[21:24], v[24]
        self.err = v[25] if keep_err else None
        for k in range(3):
            self.u[k].set(u0)
            self.um1[k].set(u0)
            self.UL[k].set(np.ascontiguousarray(ens['bc'][:, k]))
        for a in self.uE + self.DE + self.cE + [self.status, self.iters, self.counts] + ([self.err] if keep_err else []):
            a.zero()
        self.nu0.set(ens['nu0'])
        for a, src in zip(self.H7, (source_n, source_i, source_e)):
            a.set(src(np.asarray(ens['x'], dtype=np.float64)))
        P3 = H._p * 3
        ptrs = lambda arrs: P3(*[a.ptr for a in arrs])
        self.state = H.State3F(nb, n, ptrs(self.u), ptrs(self.um1), ptrs(self.uE), ptrs(self.DE), ptrs(self.cE),
                               self.status.ptr, self.iters.ptr, self.err.ptr if keep_err else None,
                               max_iter if keep_err else 0)
        q = H.Params3F()
        q.dt, q.dx, q.a_turb, q.a_prof = float(dt), float(ens['dx']), float(a_turb), float(a_prof)
        q.Dmin, q.Dmax, q.dpdx_thr, q.tol = float(theta['Dmin']), float(theta['Dmax']), float(theta['dpdxThreshold']), float(tol)
        q.UL, q.H7, q.nu0 = ptrs(self.UL), ptrs(self.H7), self.nu0.ptr
        q.fixed_count, q.frozen_turb, q.outer_count = int(fixed_count), 0, 0
        self.params = q

    def step(self, stream=None, iters=1):
        """`iters` Picard iterations of every active profile, one launch each (asynchronous)."""
        for _ in range(int(iters)):
            H.check(H.lib.tg_picard_3f_step(C.byref(self.state), C.byref(self.params), stream if stream is not None else H.stream()))

    def reduce_status(self, stream=None, counts_ptr=None):
        H.check(H.lib.tg_status_reduce(H.ptr_of(self.status), self.nb, counts_ptr or H.ptr_of(self.counts),
                                       stream if stream is not None else H.stream()))

    def reduce_residual(self, out_ptr, stream=None):
        """Launch the residual summary of tg_residual_reduce into 5 device doubles at out_ptr (needs keep_err); a
        profile's rms is taken over the 3 n residual entries of its three fields (tango/solver.py:464-470)."""
        if self.err is None:
            raise ValueError('the residual summary needs the residual history (keep_err=True)')
        H.check(H.lib.tg_residual_reduce(H.ptr_of(self.status), H.ptr_of(self.iters), H.ptr_of(self.err), self.state.err_stride,
                                         self.nb, 3 * self.n, out_ptr, stream if stream is not None else H.stream()))
