"""Three-field analytic problem n, pi, pe with nonlinear collisional coupling, packaged from
examples/shestakov_threefield_nonlinearnu.py:52-149 of the reference:

    dn/dt  + dG/dx  = Sn          G  = -D dn/dx,   D = (dpi/dx)^2 / pi^2
    dpi/dt + dQi/dx = Si - nu (pi - pe)
    dpe/dt + dQe/dx = Se - nu (pe - pi)       nu = nu0 / (pi/n + pe/n)^(3/2)

n is solved on its own, pi and pe as a pair-coupled system.  The classes below are the example's flux model and
compute_all_H callbacks; used together with solver.Solver they are recognised and run by the fused
three-field kernel (tg_picard_3f_step), one launch per Picard iteration for a whole ensemble (nu0 and the three
right boundary values may be one value per profile)."""
import ctypes as C

import numpy as np

from .. import _hip as H
from .. import derivatives, multifield

LABELS = ('n', 'pi', 'pe')


class ShestakovThreeFieldFluxModel(object):
    fused_flux_model = 'threefield_nonlinear_nu'

    def __init__(self, dx):
        self.dx = dx

    def get_flux(self, profiles):
        n, pi, pe = profiles['n'], profiles['pi'], profiles['pe']
        d = derivatives.dx_centered_difference_edge_first_order
        dndx, dpidx, dpedx = d(n, self.dx), d(pi, self.dx), d(pe, self.dx)
        D = dpidx ** 2 / pi ** 2
        return {'n': -D * dndx, 'pi': -D * dpidx, 'pe': -D * dpedx}


def _box(x, S0, delta):
    S = np.zeros_like(x)
    S[x < delta] = S0
    return S


def source_n(x, S0=8, delta=0.3):
    return _box(x, S0, delta)


def source_i(x, S0=1, delta=0.1):
    return _box(x, S0, delta)


def source_e(x, S0=3, delta=0.4):
    S = np.zeros_like(x)
    ind = x < delta
    S[ind] = S0 * x[ind]
    return S


def calc_nu(nu0, n, pi, pe):
    nu0 = np.asarray(nu0, dtype=float)
    if nu0.ndim:
        nu0 = nu0[..., None]
    return nu0 / ((pi / n + pe / n) ** (3 / 2))


class ComputeAllH_n(object):
    fused_role = 'n'

    def __call__(self, t, x, profiles, HCoeffsTurb):
        return multifield.HCoefficients(H1=np.ones_like(x), H7=source_n(x)) + HCoeffsTurb


class _ComputeAllH_p(object):
    source = None

    def __init__(self, nu0):
        self.nu0 = nu0

    def __call__(self, t, x, profiles, HCoeffsTurb):
        nu = calc_nu(self.nu0, profiles['n'], profiles['pi'], profiles['pe'])
        H = multifield.HCoefficients(H1=np.ones_like(x), H6=-nu, H7=type(self).source(x), H8=nu)
        return H + HCoeffsTurb


class ComputeAllH_pi(_ComputeAllH_p):
    fused_role = 'pi'
    source = staticmethod(source_i)


class ComputeAllH_pe(_ComputeAllH_p):
    fused_role = 'pe'
    source = staticmethod(source_e)


def plan(s):
    """Fused-path runner for solver.Solver `s`, or None if the fields are not exactly this problem."""
    by_label = {f.label: f for f in s.fields}
    if sorted(by_label) != sorted(LABELS):
        return None
    fn, fi, fe = (by_label[k] for k in LABELS)
    roles = [getattr(f.compute_all_H, 'fused_role', None) for f in (fn, fi, fe)]
    if roles != list(LABELS) or fn.coupledTo is not None or fi.coupledTo != 'pe' or fe.coupledTo != 'pi':
        return None
    if not np.array_equal(np.asarray(fi.compute_all_H.nu0), np.asarray(fe.compute_all_H.nu0)):
        return None
    th = s.compute_all_H_all_fields.turbhandler
    if getattr(th.fluxModel, 'dx', None) != s.dx or th.dxTurbGrid != s.dx:
        return None      # the kernel differentiates on the solver's own grid
    params = [f.lodestroMethod.get_ewma_params() for f in (fn, fi, fe)]
    thetas = [f.lodestroMethod._fluxSplitter.thetaParams for f in (fn, fi, fe)]
    if len(set(params)) != 1 or any(dict(th) != dict(thetas[0]) for th in thetas):
        return None
    # EWMA memories handed over by a restart must be there for all three fields or for none
    preset = [f.lodestroMethod._EWMAProfile._yEWMA_lminus1 is not None for f in (fn, fi, fe)]
    if any(preset) and not all(preset):
        return None
    return Fused3F(s, (fn, fi, fe))


class Fused3F(object):
    """Device-resident three-field ensemble stepped by tg_picard_3f_step."""

    def __init__(self, s, fields):
        self._alloc_state(s, fields)
        D = H.DeviceArray
        self.nu0 = D.from_host(np.broadcast_to(np.asarray(fields[1].compute_all_H.nu0, dtype=np.float64), (self.nb,)))
        x = np.asarray(s.x, dtype=np.float64)
        self.H7 = [D.from_host(src(x)) for src in (source_n, source_i, source_e)]

    def _alloc_state(self, s, fields):
        """The device-resident state of the three fields (order: the field solved on its own, then the pair)."""
        self.fields = fields
        u0 = [np.atleast_2d(np.asarray(s.profiles[f.label], dtype=np.float64)) for f in fields]
        self.nb, self.n = u0[0].shape
        nb, n = self.nb, self.n
        D = H.DeviceArray
        from ..ensemble import device_arrays      # the fifteen state arrays: one allocation, skewed offsets
        arrs = device_arrays([((nb, n), np.float64)] * 15, zero=True)
        self.u, self.um1, self.uE, self.DE, self.cE = (arrs[3 * k:3 * k + 3] for k in range(5))
        for k, f in enumerate(fields):
            self.u[k].set(u0[k])
            self.um1[k].set(np.atleast_2d(np.asarray(f.profile_mminus1, dtype=np.float64)))
        self.status = D((nb,), np.int32, zero=True)
        # restart: EWMA memories supplied by the caller (tango/restart.py:90-98) seed the three fields
        if fields[0].lodestroMethod._EWMAProfile._yEWMA_lminus1 is not None:
            for k, f in enumerate(fields):
                m = f.lodestroMethod
                self.uE[k].set(np.atleast_2d(m._EWMAProfile._yEWMA_lminus1))
                self.DE[k].set(np.atleast_2d(m._EWMATurbD._yEWMA_lminus1))
                self.cE[k].set(np.atleast_2d(m._EWMATurbc._yEWMA_lminus1))
            self.status.set(np.full((nb,), H.ST_SEEDED, np.int32))
        self.iters = D((nb,), np.int32, zero=True)
        self.err = D((nb, s.maxIterations), zero=True)
        self.counts = D((4,), np.int64, zero=True)
        self.UL = [D.from_host(np.broadcast_to(np.asarray(f.rightBC, dtype=np.float64), (nb,))) for f in fields]
        P3 = H._p * 3
        ptrs = lambda arrs: P3(*[a.ptr for a in arrs])
        self.state = H.State3F(nb, n, ptrs(self.u), ptrs(self.um1), ptrs(self.uE), ptrs(self.DE), ptrs(self.cE),
                               self.status.ptr, self.iters.ptr, self.err.ptr, s.maxIterations)
        self._ptrs = ptrs

    def params(self, s):
        m = self.fields[0].lodestroMethod
        a_turb, a_prof = m.get_ewma_params()
        tp = m._fluxSplitter.thetaParams
        q = H.Params3F()
        q.dt, q.dx, q.a_turb, q.a_prof = float(s.dt), float(s.dx), float(a_turb), float(a_prof)
        q.Dmin, q.Dmax, q.dpdx_thr, q.tol = float(tp['Dmin']), float(tp['Dmax']), float(tp['dpdxThreshold']), float(s.tol)
        q.UL, q.H7, q.nu0 = self._ptrs(self.UL), self._ptrs(self.H7), self.nu0.ptr
        q.fixed_count, q.frozen_turb, q.outer_count = 0, 0, 0
        # pseudo-timestep control (tango/solver.py:198-216, 311-372) runs inside the kernel
        q.pseudo_time = int(bool(s.useTreadLightly or s.useTreadInitially))
        if s.useTreadLightly:
            q.tread_lightly = 1
            q.tread_max_change = float(s.treadLightlyParams.get('maxFractionalChange', 0.2))
            q.tread_inner_count = int(s.treadLightlyParams.get('innerFraction', 0.5) * self.n)
        if s.useTreadInitially:
            q.tread_initial_steps = int(s.treadInitiallyParams['numInitialSteps'])
            q.tread_initial_dtau = float(s.treadInitiallyParams['dtau'])
        return q

    def run_timestep(self, s):
        q = self.params(s)
        if s.m > 1:
            for a, b in zip(self.um1, self.u):
                a.copy_from(b)
            self.status.set((self.status.to_host() & H.ST_SEEDED).astype(np.int32))
            H.check(H.lib.tg_memset(self.iters.ptr, 0, self.iters.nbytes, H.stream()))
            H.check(H.lib.tg_memset(self.err.ptr, 0, self.err.nbytes, H.stream()))
        inner = s.innerIterationMaxCount if s.useInnerIteration else 0
        hist = s.profilesAllIterations
        done = start = self.start = s.iterationNumber
        # The iterate history, when kept, is collected on the device (one copy per field and iteration, no host round
        # trip) and read back once at the end; beyond 256 MiB it is streamed to the host iterate by iterate instead.
        rows = s.maxIterations - done
        snaps = None
        if hist is not None and rows > 0 and 3 * rows * self.u[0].nbytes <= (256 << 20):
            snaps = [H.DeviceArray((rows, self.nb, self.n)) for _ in self.u]
        while done < s.maxIterations:
            q.frozen_turb, q.outer_count = 0, 0
            H.check(H.lib.tg_picard_3f_step(C.byref(self.state), C.byref(q), H.stream()))
            done += 1
            q.frozen_turb, q.outer_count = 1, done
            for _ in range(inner):      # inner iteration with frozen turbulent coefficients (tango/solver.py:274-309)
                H.check(H.lib.tg_picard_3f_step(C.byref(self.state), C.byref(q), H.stream()))
            if snaps is not None:       # a profile that has stopped keeps its values: its later rows repeat them
                for dst, d in zip(snaps, self.u):
                    H.check(H.lib.tg_copy_d2d(dst.ptr + (done - 1 - start) * d.nbytes, d.ptr, d.nbytes, H.stream()))
            elif hist is not None:
                for f, d in zip(self.fields, self.u):
                    snap = d.to_host()
                    hist[f.label][done - 1 - start] = snap if s.batched else snap[0]
            if (done - start) % s.iterationsPerCheck == 0 or done == s.maxIterations:
                H.check(H.lib.tg_status_reduce(self.status.ptr, self.nb, self.counts.ptr, H.stream()))
                if int(self.counts.to_host()[0]) == 0:
                    break
        if snaps is not None:
            top = done - start
            for f, dst in zip(self.fields, snaps):
                got = np.empty((top, self.nb, self.n))
                if top:
                    H.check(H.lib.tg_copy_d2h(got.ctypes.data, dst.ptr, got.nbytes, H.stream()))
                    H.check(H.lib.tg_stream_sync(H.stream()))
                hist[f.label][:top] = got if s.batched else got[:, 0]
        self.finish(s)

    def finish(self, s):
        status, iters = self.status.to_host(), self.iters.to_host()
        top = int(iters.max()) if iters.size else 0
        err = np.zeros(self.err.shape)                 # only the columns that were written cross PCIe
        err[:, :top] = self.err.columns_to_host(top)
        s.iterationsPerProfile = iters.astype(np.int64)
        s.countStoredIterations = int(iters.max()) if iters.size else 0
        s.iterationNumber = getattr(self, 'start', 0) + s.countStoredIterations
        s.converged = bool(np.all(status & H.ST_CONVERGED))
        s.solutionError = bool(np.any(status & H.ST_INVALID))
        s.profileStatus = status
        pick = (lambda a: a) if s.batched else (lambda a: a[0])
        s.errHistory = err if s.batched else err[0].copy()
        s.profiles = {f.label: pick(d.to_host()) for f, d in zip(self.fields, self.u)}
        for k, f in enumerate(self.fields):
            if s.batched:      # an ensemble's memories stay on the device until someone reads them
                f.lodestroMethod.set_ewma_iterates(self.uE[k].to_host, self.DE[k].to_host, self.cE[k].to_host)
            else:
                f.lodestroMethod.set_ewma_iterates(self.uE[k].to_host()[0], self.DE[k].to_host()[0], self.cE[k].to_host()[0])
