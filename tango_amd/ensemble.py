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


class OneFieldEnsemble:
    """Device-resident state of a one-field ensemble plus the parameter block of tg_picard_1f_step.

    alloc(shape, dtype) -> object with data_ptr(): where the arrays live (default: DeviceArray; bench.py
    passes a torch allocator so that torch owns the memory and the stream)."""

    def __init__(self, ens, dt=1e4, a_turb=0.3, a_prof=1.0, tol=1e-11, theta=THETA_DEFAULT, max_iter=1000,
                 fixed_count=False, keep_err=True, upload=None):
        u0 = np.ascontiguousarray(ens['u0'], dtype=np.float64)
        self.nb, self.n = u0.shape
        up = upload or (lambda a: H.DeviceArray.from_host(a))
        zeros = (lambda shape, dtype=np.float64: up(np.zeros(shape, dtype)))
        self.u, self.um1 = up(u0), up(u0)
        self.uE, self.DE, self.cE = zeros(u0.shape), zeros(u0.shape), zeros(u0.shape)
        self.status = zeros((self.nb,), np.int32)
        self.iters = zeros((self.nb,), np.int32)
        self.err = zeros((self.nb, max_iter)) if keep_err else None
        self.counts = zeros((4,), np.int64)
        self.UL, self.S0, self.delta = up(ens['nL']), up(ens['S0']), up(ens['delta'])
        self.x = up(np.ascontiguousarray(ens['x'], dtype=np.float64))
        p = H.ptr_of
        self.state = H.State1F(self.nb, self.n, p(self.u), p(self.um1), p(self.uE), p(self.DE), p(self.cE),
                               p(self.status), p(self.iters), p(self.err), max_iter if keep_err else 0)
        q = H.Params1F()
        q.dt, q.dx, q.a_turb, q.a_prof = float(dt), float(ens['dx']), float(a_turb), float(a_prof)
        q.Dmin, q.Dmax, q.dpdx_thr, q.tol = float(theta['Dmin']), float(theta['Dmax']), float(theta['dpdxThreshold']), float(tol)
        q.UL, q.S0, q.delta, q.x = p(self.UL), p(self.S0), p(self.delta), p(self.x)
        q.flux_model, q.smooth_window, q.fixed_count, q.frozen_turb = H.FLUX_SHESTAKOV, 1, int(fixed_count), 0
        self.params = q
        self.max_iter = max_iter

    def step(self, stream=None):
        """One Picard iteration of every active profile (asynchronous)."""
        H.check(H.lib.tg_picard_1f_step(C.byref(self.state), C.byref(self.params), stream if stream is not None else H.stream()))

    def reduce_status(self, stream=None):
        """Launch the status reduction; counts = (active, converged, invalid, singular) lands in self.counts."""
        H.check(H.lib.tg_status_reduce(H.ptr_of(self.status), self.nb, H.ptr_of(self.counts),
                                       stream if stream is not None else H.stream()))
