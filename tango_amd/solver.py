"""Time-step / Picard-iteration driver with the interface of tango/solver.py:32-595.

Two execution paths sit behind the same Solver class:

* generic path -- any user compute_all_H callbacks and any flux model, as in the reference.  Every
  numerical step (EWMA, flux split, assembly, residual, tridiagonal / pair-coupled solve) is a HIP kernel
  called through tango_amd's modules; arrays cross the host boundary around each call.  With 1-D profiles
  the control flow and results are the reference's; with (profiles, N) arrays all profiles advance in
  lock-step until every one of them has converged.

* fused path -- chosen automatically for the problems whose flux model and source are built into the fused
  kernels (one-field Shestakov ensembles: extras.shestakov_nonlinear_diffusion.AnalyticFluxModel +
  ComputeAllH; the three-field nonlinear-nu ensemble of extras.threefield).  The state stays on the device,
  one kernel launch advances every active profile by one full iteration, and each profile is frozen on the
  iteration on which the reference would have stopped it (its residual, evaluated before the solve, is below
  tol; the solve of that iteration still happens, tango/solver.py:150-154, 219-228).
"""
from __future__ import division, absolute_import

import ctypes as C
import time

import numpy as np

from . import _hip as H
from . import fieldgroups, tango_logging
from . import multifield as _mf
from . import lodestro_method as _lm
from . import smoother as _smoother


def duration_as_hms(seconds):
    """'h:mm:ss.s' rendering of a duration (tango/utilities/util.py:3-20)."""
    m, s = divmod(seconds, 60)
    h, m = divmod(m, 60)
    return '%d:%02d:%04.1f' % (h, m, s)


class Solver(object):
    def __init__(self, L, x, tArray, maxIterations, tol, compute_all_H_all_fields, fields,
                 startIterationNumber=0, profiles=None, maxIterationsPerSet=np.inf,
                 useTreadLightly=False, treadLightlyParams=None,
                 useTreadInitially=False, treadInitiallyParams=None,
                 useInnerIteration=False, innerIterationMaxCount=20,
                 user_control_func=None,
                 saveFluxesInMemory=False,
                 iterationsPerCheck=1, saveProfileHistory=None, detectStaticH=True, fusedPaths=True):
        """Arguments up to saveFluxesInMemory are those of tango/solver.py:33-39.

        iterationsPerCheck  (fused path) iterations per kernel launch and between two host reads of the
                            convergence counters.  Each profile still stops on its own iteration (the test is made
                            every iteration on the device); with k > 1 the profiles stay in registers between
                            iterations, which removes most of the HBM traffic.
        saveProfileHistory  keep profilesAllIterations (profiles x maxIterations x N values); default: only
                            for single-profile runs.
        detectStaticH       one-field problems: find out by two trial calls (different profiles, first and last
                            time) whether the field's compute_all_H callback contributes the same H1/H7 whatever
                            the profiles and the time, as the callbacks of the example scripts do
                            (examples/shestakov_basic.py:41-53); if so the iteration runs on the fused path with the
                            flux model called on the host.  A callback can state the answer itself with an attribute
                            static_H = True / False, which suppresses the trial calls.
        fusedPaths          False keeps the run on the generic path (one kernel per reference function, host arrays in
                            between) whatever the problem.
        """
        self.L = L
        self.x = x
        self.N = len(x)
        self.dx = x[1] - x[0]
        self.tArray = tArray
        self.maxIterations = maxIterations
        self.tol = tol
        self.compute_all_H_all_fields = compute_all_H_all_fields
        self.fields = fields
        self.useTreadLightly = useTreadLightly
        self.treadLightlyParams = treadLightlyParams if treadLightlyParams is not None else {}
        self.useTreadInitially = useTreadInitially
        self.treadInitiallyParams = treadInitiallyParams if treadInitiallyParams is not None else {}
        self.useInnerIteration = useInnerIteration
        self.innerIterationMaxCount = innerIterationMaxCount
        self.saveFluxesInMemory = saveFluxesInMemory
        if user_control_func is not None:
            assert callable(user_control_func), "user_control_func must be callable (e.g., a function)"
        self.user_control_func = user_control_func

        if profiles is None:
            self.profiles = {field.label: field.profile_mminus1 for field in fields}
        else:
            self.profiles = profiles
        self.fileHandlerExecutor = _NoHandlers()

        self.solutionError = False
        self.t = 0
        self.dt = None
        self.countStoredIterations = 0
        self.iterationNumber = startIterationNumber
        self.maxIterationsPerSet = maxIterationsPerSet
        self.converged = None
        self.errHistory = np.zeros(self.maxIterations)
        self.errHistoryFinal = None
        self.tFinal = tArray[-1]
        self.m = 0
        self.reachedEnd = False
        self.profilesAllIterations = None
        self.fluxesAllIterations = None
        self.normalizedResids = None
        self.datadict = None
        self.iterationsPerProfile = None
        self.countLastTimestep = None
        self.restart_profile_mminus1 = None

        first = np.asarray(self.profiles[fields[0].label])
        self.batched = first.ndim == 2
        self.numProfiles = first.shape[0] if self.batched else 1
        self.saveProfileHistory = (not self.batched) if saveProfileHistory is None else saveProfileHistory
        self.iterationsPerCheck = max(1, int(iterationsPerCheck))
        self.detectStaticH = bool(detectStaticH)
        self.fusedPaths = bool(fusedPaths)

        self.profilesAllTimesteps = {}
        self.fluxesAllTimesteps = {}
        for field in self.fields:
            shape = (len(tArray),) + np.shape(self.profiles[field.label])
            self.profilesAllTimesteps[field.label] = np.zeros(shape)
            self.fluxesAllTimesteps[field.label] = np.zeros(shape)
            self.profilesAllTimesteps[field.label][self.m] = self.profiles[field.label]

        self.is_unacceptable_one_profile = is_unacceptable_one_profile
        self._fused = _plan_fused(self)

    # ------------------------------------------------------------------ time step
    def take_timestep(self):
        """One implicit time step: iterate until converged, out of iterations, or invalid."""
        self.m += 1
        self.converged = False
        (self.t, self.dt) = self._get_tdt(self.tArray, self.m)
        self.errHistory[:] = 0
        self._alloc_history()
        # what a restart inside this timestep needs besides the profiles (tango_amd/restart.py)
        self.restart_profile_mminus1 = {f.label: np.array(f.profile_mminus1, dtype=np.float64) for f in self.fields}
        tango_logging.info("Timestep m={}:  Beginning iteration loop ...".format(self.m))

        if self._fused is not None:
            self._fused.run_timestep(self)
        else:
            while (self.iterationNumber < self.maxIterations
                   and self.countStoredIterations < self.maxIterationsPerSet
                   and not self.converged
                   and not self.solutionError):
                self.compute_next_iteration()

        if self.converged:
            tango_logging.info("Timestep m={}:  Converged!  Successfully found the solution for t={}.  Rms error={}.  "
                               "Took {} iterations.".format(self.m, self.t, self._last_err(), self.iterationNumber))
            if self.m >= len(self.tArray) - 1:
                tango_logging.info("Reached the final timestep m={} at t={}.  Simulation ending...".format(self.m, self.t))
                self.reachedEnd = True

        k = self.countStoredIterations
        self.errHistoryFinal = self.errHistory[..., 0:k]
        for field in self.fields:
            if self.profilesAllIterations is not None:
                self.profilesAllIterations[field.label] = self.profilesAllIterations[field.label][0:k]
                self.fluxesAllIterations[field.label] = self.fluxesAllIterations[field.label][0:k]
            self.profilesAllTimesteps[field.label][self.m] = self.profiles[field.label]
            if self.profilesAllIterations is not None and k > 0:
                self.fluxesAllTimesteps[field.label][self.m] = self.fluxesAllIterations[field.label][-1]
                if self.m == 1:
                    self.fluxesAllTimesteps[field.label][0] = self.fluxesAllIterations[field.label][0]
        for field in self.fields:
            field.profile_mminus1 = self.profiles[field.label]
        self.countLastTimestep = self.iterationNumber      # iterations completed, counted from startIterationNumber
        self.iterationNumber = 0
        self.countStoredIterations = 0

    def _alloc_history(self):
        if not self.saveProfileHistory:
            self.profilesAllIterations = None
            self.fluxesAllIterations = None
            return
        shape = (self.maxIterations,) + np.shape(self.profiles[self.fields[0].label])
        self.profilesAllIterations = {f.label: np.zeros(shape) for f in self.fields}
        self.fluxesAllIterations = {f.label: np.zeros(shape) for f in self.fields}

    def _last_err(self):
        i = self.iterationNumber - 1
        return self.errHistory[..., i] if i >= 0 else None

    # ------------------------------------------------------------------ generic path
    def _assemble(self, HCoeffsAllFields, dtau):
        groups = fieldgroups.fields_to_fieldgroups(self.fields, HCoeffsAllFields)
        for g in groups:
            g.matrixEqn = g.Hcoeffs_to_matrix_eqn(dtau, self.dx, g.rightBC, g.psi_mminus1, g.HCoeffs)
        return groups

    @staticmethod
    def _solve(groups):
        for g in groups:
            g.profileSolution = g.solve_matrix_eqn(g.matrixEqn)
        return fieldgroups.fieldgroups_to_profiles(groups)

    def compute_next_iteration(self):
        """One Picard iteration of the generic path (tango/solver.py:188-272)."""
        startTime = time.time()
        if self.user_control_func is not None:
            self.user_control_func(self)
        index = self.countStoredIterations

        if self.useTreadLightly or self.useTreadInitially:
            for field in self.fields:     # the "previous timestep" slot holds the previous iterate
                field.profile_mminus1 = self.profiles[field.label]

        (HCoeffsAllFields, HCoeffsTurbAllFields, extradataAllFields) = \
            self.compute_all_H_all_fields(self.t, self.x, self.profiles)

        dtau = self.dt
        if self.useTreadInitially and self.iterationNumber < self.treadInitiallyParams['numInitialSteps']:
            dtau = self.treadInitiallyParams['dtau']

        groups = self._assemble(HCoeffsAllFields, dtau)
        (self.converged, rmsError, normalizedResids) = self.check_convergence(groups, self.profiles, self.tol)
        self.errHistory[index] = np.max(rmsError)
        self.normalizedResids = normalizedResids
        self.profiles = self._solve(groups)

        if self.useTreadLightly:
            self.tread_lightly(HCoeffsTurbAllFields)
        if self.useInnerIteration:
            (HCoeffsAllFields, _, _) = self.perform_inner_iteration(HCoeffsTurbAllFields)

        tango_logging.debug("...iteration {} took a wall time of {}".format(
            self.iterationNumber, duration_as_hms(time.time() - startTime)))

        self.datadict = self._pkgdata(HCoeffsAllFields=HCoeffsAllFields, extradataAllFields=extradataAllFields,
                                      profiles=self.profiles, normalizedResids=normalizedResids,
                                      rmsError=rmsError, iterationNumber=self.iterationNumber)
        if self.profilesAllIterations is not None:
            for field in self.fields:
                self.profilesAllIterations[field.label][index] = self.profiles[field.label]
                if self.saveFluxesInMemory and extradataAllFields is not None:
                    self.fluxesAllIterations[field.label][index] = extradataAllFields[field.label]['fluxTurbGrid']
        self.fileHandlerExecutor.execute_scheduled(self.datadict, self.iterationNumber)
        self.check_profiles_are_valid(self.profiles)

        self.countStoredIterations += 1
        self.iterationNumber += 1
        if self.iterationNumber >= self.maxIterations:
            tango_logging.info("Timestep m={} and time t={}:  maxIterations ({}) reached.  Error={} while tol={}".format(
                self.m, self.t, self.maxIterations, rmsError, self.tol))
        if self.countStoredIterations >= self.maxIterationsPerSet:
            tango_logging.info("Timestep m={} and time t={}:  maxIterationsPerSet ({}) reached.  Error={} while tol={}.  "
                               "Shutting down...".format(self.m, self.t, self.maxIterationsPerSet, rmsError, self.tol))

    def perform_inner_iteration(self, HCoeffsTurbAllFields):
        """A fixed number of iterations with the turbulent coefficients frozen (tango/solver.py:274-309)."""
        count = 0
        while True:
            (HCoeffsAllFields, _, _) = self.compute_all_H_all_fields(
                self.t, self.x, self.profiles, computeTurbulence=False, HCoeffsTurbAllFields=HCoeffsTurbAllFields)
            groups = self._assemble(HCoeffsAllFields, self.dt)
            (_, rmsError, normalizedResids) = self.check_convergence(groups, self.profiles, self.tol)
            self.profiles = self._solve(groups)
            count += 1
            if count >= self.innerIterationMaxCount:
                break
        return (HCoeffsAllFields, normalizedResids, rmsError)

    def tread_lightly(self, HCoeffsTurbAllFields):
        """Halve the pseudo-timestep, restarting from the previous iterate with the turbulent coefficients
        frozen, until no profile changes too much in one iteration (tango/solver.py:311-372)."""
        previous = {field.label: field.profile_mminus1 for field in self.fields}
        if not self.is_unacceptable(previous, self.profiles):
            return

        def trial(dtau):
            (HCoeffsAllFields, _, _) = self.compute_all_H_all_fields(
                self.t, self.x, previous, computeTurbulence=False, HCoeffsTurbAllFields=HCoeffsTurbAllFields)
            return self._solve(self._assemble(HCoeffsAllFields, dtau))

        dtau = self.dt
        while True:
            dtau /= 2
            trialProfiles = trial(dtau)
            if not self.is_unacceptable(previous, trialProfiles):
                break
        tango_logging.info(f'used tread lightly on iteration {self.iterationNumber}.  dtau={dtau}')
        self.profiles = trialProfiles

    def is_unacceptable(self, oldProfiles, newProfiles):
        maxFractionalChange = self.treadLightlyParams.get('maxFractionalChange', 0.2)
        innerFraction = self.treadLightlyParams.get('innerFraction', 0.5)
        for field in self.fields:
            if self.is_unacceptable_one_profile(oldProfiles[field.label], newProfiles[field.label],
                                                maxFractionalChange=maxFractionalChange, innerFraction=innerFraction):
                return True
        return False

    @property
    def ok(self):
        """True until a stop condition is met (tango/solver.py:404-418)."""
        if self.solutionError:
            return False
        if self.m >= len(self.tArray) - 1 or self.t >= self.tFinal:
            return False
        if self.iterationNumber >= self.maxIterations:
            return False
        if self.countStoredIterations >= self.maxIterationsPerSet:
            return False
        return True

    def check_profiles_are_valid(self, profiles):
        """NaN, Inf or negative values stop the solve gracefully (tango/solver.py:420-439)."""
        for label in profiles:
            profile = np.asarray(profiles[label])
            if not np.all(np.isfinite(profile)):
                tango_logging.error("field {}: NaN or Inf detected in profile at l={}.  Aborting...".format(label, self.iterationNumber))
                self.solutionError = True
                break
            if np.any(profile < 0):
                tango_logging.error("field {}: Negative value detected in profile at l={}.  Aborting...".format(label, self.iterationNumber))
                self.solutionError = True
                break

    def check_convergence(self, fieldGroups, profiles, tol):
        """rms over all fields of the normalised residuals (tango/solver.py:441-475); one value per profile
        for ensembles, converged when every profile is below tol."""
        normalizedResids = {}
        for g in fieldGroups:
            normalizedResids.update(g.calculate_residual(g.matrixEqn, profiles))
        totalLength = 0
        sumOfSquaresResid = 0
        for label in normalizedResids:
            resid = normalizedResids[label]
            totalLength += np.shape(resid)[-1]
            sumOfSquaresResid = sumOfSquaresResid + np.sum(resid ** 2, axis=-1)
        rmsError = np.sqrt(1 / totalLength * sumOfSquaresResid)
        converged = bool(np.all(rmsError < tol))
        return (converged, rmsError, normalizedResids)

    @staticmethod
    def _get_tdt(tArray, timestepNumber):
        tNew = tArray[timestepNumber]
        return tNew, tNew - tArray[timestepNumber - 1]

    def _pkgdata(self, HCoeffsAllFields=None, extradataAllFields=None, profiles=None, normalizedResids=None,
                 rmsError=None, iterationNumber=None):
        """Per-iteration data in the layout the reference hands to its file handlers (tango/solver.py:497-532)."""
        data = {'errHistory': rmsError, 'iterationNumber': iterationNumber}
        for label in HCoeffsAllFields:
            entry = {'profile': profiles[label], 'normalizedResid': normalizedResids[label]}
            if extradataAllFields is not None:
                entry = self._merge_two_dicts(entry, extradataAllFields[label])
            data[label] = entry
            self._pkgdata_HCoeffs_helper(data, HCoeffsAllFields, label)
        return data

    def _pkgdata_HCoeffs_helper(self, data, HCoeffsAllFields, label):
        for name in ('H1', 'H2', 'H3', 'H4', 'H6', 'H7', 'H8'):
            Hj = getattr(HCoeffsAllFields[label], name)
            data[label][name] = Hj if Hj is not None else np.zeros_like(HCoeffsAllFields[label].H1)

    @staticmethod
    def _merge_two_dicts(x, y):
        z = dict(x)
        z.update(y)
        return z


class _NoHandlers(object):
    """Stand-in for the reference's file-handler executor (tango/handlers.py:15-83): HDF5 output is outside this
    package; the per-iteration data is available as Solver.datadict."""

    def execute_scheduled(self, datadict, iterationNumber):
        pass

    def add_handler(self, handler):
        raise NotImplementedError('file handlers are not part of tango_amd; read Solver.datadict instead')


def is_unacceptable_one_profile(oldProfile, newProfile, maxFractionalChange=0.2, innerFraction=0.5):
    """True if any point of the inner part of the profile changed by at least maxFractionalChange
    (tango/solver.py:564-595)."""
    oldProfile, newProfile = np.asarray(oldProfile), np.asarray(newProfile)
    k = int(innerFraction * oldProfile.shape[-1])
    change = (newProfile[..., :k] - oldProfile[..., :k]) / oldProfile[..., :k]
    return bool(np.any(np.abs(change) >= maxFractionalChange))


# ---------------------------------------------------------------------- fused path
def _plan_fused(s):
    """Return a fused-path runner when the problem is one the fused kernels implement, else None."""
    if not s.fusedPaths or s.user_control_func is not None or s.saveFluxesInMemory or s.maxIterationsPerSet != np.inf:
        return None
    if s.useTreadLightly and s.is_unacceptable_one_profile is not is_unacceptable_one_profile:
        return None      # a user-supplied acceptance test runs on the host
    if s.useTreadInitially and s.iterationNumber != 0:
        return None
    call = s.compute_all_H_all_fields
    if not isinstance(call, _mf.ComputeAllHAllFields) or type(call.turbhandler) is not _lm.TurbulenceHandler:
        return None
    th = call.turbhandler
    if call.fields is not s.fields and list(call.fields) != list(s.fields):
        return None
    if len(s.fields) == 1 and type(s.fields[0].gridMapper) is not _mf.GridsNull:
        return _FusedMapped1F.plan(s)      # a turbulence grid of its own: the device-resident pipeline, or the generic path
    for f in s.fields:
        m = f.lodestroMethod
        if type(f.gridMapper) is not _mf.GridsNull or type(m) is not _lm.lm:
            return None
        if m._fluxSplitter.ftheta != m._fluxSplitter._default_ftheta:
            return None

    def plain(sp):
        return not (np.ndim(sp.gxxAvg) or np.ndim(sp.gradxAvg) or sp.gxxAvg != 1 or sp.gradxAvg != 1)

    inner, decorators = _unwrap_flux_model(th.fluxModel)
    model = getattr(th.fluxModel, 'fused_flux_model', None)
    n_points = np.shape(s.x)[-1]
    # the packaged moving average runs inside the fused kernel (grids of up to 4096 points); any other smoother on the host
    fused_smoothing = th.doSmoothing and type(th.fluxSmoother) is _smoother.Smoother and n_points <= 4096 \
        and th.fluxSmoother.windowSize <= n_points
    one = len(s.fields) == 1 and s.fields[0].coupledTo is None and not s.useInnerIteration and th.dxTurbGrid == s.dx
    builtin = getattr(inner, 'fused_flux_model', None)
    # the flux models built into the kernel, bare or behind the packaged decorators (taper, device-generated noise), which
    # then act as the kernel's flux multiplier
    decorated_ok = not decorators or (n_points <= 4096 and sum(d.flux_decorator == 'noise' for d in decorators) <= 1
                                      and all(d.flux_decorator == 'taper' or d.device_seed is not None for d in decorators))
    if one and builtin in ('shestakov', 'power') and decorated_ok and (builtin == 'shestakov' or n_points <= 4096) \
            and (not th.doSmoothing or fused_smoothing) and not th.isNonCartesian and plain(s.fields[0].lodestroMethod._fluxSplitter) \
            and getattr(s.fields[0].compute_all_H, 'fused_source', None) is not None and inner.dx == s.dx:
        return _Fused1F(s, inner, decorators)
    if one and _callback_is_static(s):
        plan = _FusedExternal1F.plan(s)
        if plan is not None:
            return plan
    if th.isNonCartesian or not all(plain(f.lodestroMethod._fluxSplitter) for f in s.fields):
        return None
    if (s.useTreadLightly or s.useTreadInitially) and s.useInnerIteration:
        return None      # the kernels do not combine pseudo-timestep control with inner iterations
    if model == 'threefield_nonlinear_nu' and len(s.fields) == 3 and not th.doSmoothing:
        from .extras import threefield
        runner = threefield.plan(s)
        if runner is not None:
            return runner
    if len(s.fields) == 3:
        return _FusedExternal3F.plan(s)      # a field on its own + a coupled pair with a Python flux model and callbacks
    return None


def _unwrap_flux_model(model):
    """(innermost flux model, the packaged decorators around it, outermost first)."""
    from .extras import bufferzone, noisyflux
    chain = []
    while type(model) in (bufferzone.BufferZone, noisyflux.NoisyFlux):
        chain.append(model)
        model = model.fluxModel
    return model, chain


def _callback_is_static(s):
    """Does the one field's compute_all_H callback contribute coefficients that depend on x only?  Its attribute
    static_H answers when present; otherwise (detectStaticH) the callback is called for the first and the last time of
    tArray with two different sets of profiles and no turbulent coefficients, and the answer is yes when every
    coefficient comes back bit for bit the same."""
    f = s.fields[0]
    cb = f.compute_all_H
    flag = getattr(cb, 'static_H', None)
    s._static_by_trial = False
    if flag is not None:
        return bool(flag)
    if not s.detectStaticH:
        return False
    s._static_by_trial = True
    try:
        p0 = np.array(s.profiles[f.label], dtype=np.float64)
        p1 = p0 * 1.37 + 0.11 * (1.0 + np.asarray(s.x, dtype=np.float64))
        a = cb(s.tArray[0], s.x, {f.label: p0}, _mf.HCoefficients())
        b = cb(s.tArray[-1], s.x, {f.label: p1}, _mf.HCoefficients())
    except Exception:
        return False
    if not isinstance(a, _mf.HCoefficients) or not isinstance(b, _mf.HCoefficients):
        return False
    for k in ('H1', 'H2', 'H3', 'H4', 'H6', 'H7', 'H8'):
        va, vb = getattr(a, k), getattr(b, k)
        if (va is None) != (vb is None):
            return False
        if va is not None and not np.array_equal(np.asarray(va), np.asarray(vb)):
            return False
    tango_logging.info('compute_all_H of field %r returned the same coefficients for two different profiles and times: its H1 and '
                       'H7 are evaluated once and the run takes the fused path (checked again at every convergence check; set '
                       'static_H on the callback, or detectStaticH=False, to decide yourself)' % f.label)
    return True


class _Fused1F(object):
    """Device-resident one-field ensemble stepped by tg_picard_1f_step."""

    def __init__(self, s, model=None, decorators=()):
        f = s.fields[0]
        self.label = f.label
        u0 = np.atleast_2d(np.asarray(s.profiles[f.label], dtype=np.float64))
        self.nb, self.n = u0.shape
        nb = self.nb
        self.model = model
        # flux-model decorators: the tapers multiply into one (n,) window, a noise decorator fills an (nb, n) multiplier
        # on the device before every iteration
        self.taper = self.mult = self.noise = None
        tapers = [d for d in decorators if d.flux_decorator == 'taper']
        if tapers:
            from .extras.bufferzone import tukey_window
            w = np.ones(self.n)
            for d in tapers:
                w = w * tukey_window(self.n, 2 * d.taperwidth)
            self.taper = H.DeviceArray.from_host(w)
        for d in decorators:
            if d.flux_decorator == 'noise':
                self.noise = d
                self.mult = H.DeviceArray((nb, self.n))
        # the five state arrays come from one allocation with skewed offsets (ensemble.device_arrays)
        from .ensemble import device_arrays
        self.u, self.um1, self.uE, self.DE, self.cE = device_arrays([((nb, self.n), np.float64)] * 5, zero=True)
        self.u.set(u0)
        self.um1.set(np.atleast_2d(np.asarray(f.profile_mminus1, dtype=np.float64)))
        self.status = H.DeviceArray((nb,), np.int32, zero=True)
        self.iters = H.DeviceArray((nb,), np.int32, zero=True)
        self.err = H.DeviceArray((nb, s.maxIterations), zero=True)
        self.counts = H.DeviceArray((4,), np.int64, zero=True)
        bc = np.broadcast_to(np.asarray(f.rightBC, dtype=np.float64), (nb,))
        _, S0, delta = f.compute_all_H.fused_source
        self.UL = H.DeviceArray.from_host(bc)
        from .ensemble import source_cut
        self.S0 = H.DeviceArray.from_host(np.broadcast_to(np.asarray(S0, dtype=np.float64), (nb,)))
        cut = np.broadcast_to(source_cut(s.x, delta), (nb,))
        self.src_cut = H.DeviceArray.from_host(np.ascontiguousarray(cut, dtype=np.int32))
        m = f.lodestroMethod
        if m._EWMAProfile._yEWMA_lminus1 is not None:    # restart: EWMA memories supplied by the caller
            self.uE.set(np.atleast_2d(m._EWMAProfile._yEWMA_lminus1))
            self.DE.set(np.atleast_2d(m._EWMATurbD._yEWMA_lminus1))
            self.cE.set(np.atleast_2d(m._EWMATurbc._yEWMA_lminus1))
            self.status.set(np.full((nb,), H.ST_SEEDED, np.int32))
        self.state = H.State1F(nb, self.n, self.u.ptr, self.um1.ptr, self.uE.ptr, self.DE.ptr, self.cE.ptr,
                               self.status.ptr, self.iters.ptr, self.err.ptr, s.maxIterations)
        # segmented form of the step for long profiles; under pseudo-timestep control with room for the previous iterate
        sized = H.lib.tg_picard_1f_work_bytes_tread if (s.useTreadLightly or s.useTreadInitially) else H.lib.tg_picard_1f_work_bytes
        wb = int(sized(nb, self.n))
        self.work = H.DeviceArray((wb // 8,)) if wb else None

    def params(self, s):
        m = s.fields[0].lodestroMethod
        a_turb, a_prof = m.get_ewma_params()
        tp = m._fluxSplitter.thetaParams
        q = H.Params1F()
        q.dt, q.dx, q.a_turb, q.a_prof = float(s.dt), float(s.dx), float(a_turb), float(a_prof)
        q.Dmin, q.Dmax, q.dpdx_thr, q.tol = float(tp['Dmin']), float(tp['Dmax']), float(tp['dpdxThreshold']), float(s.tol)
        ptr = lambda a: a.ptr if a is not None else None
        q.UL, q.S0, q.src_cut = self.UL.ptr, ptr(self.S0), ptr(self.src_cut)
        if getattr(self, 'work', None) is not None:
            q.work, q.work_bytes = self.work.ptr, self.work.nbytes
        q.flux_model, q.smooth_window, q.fixed_count, q.frozen_turb = H.FLUX_SHESTAKOV, self._smooth_window(s), 0, 0
        if getattr(getattr(self, 'model', None), 'fused_flux_model', None) == 'power':
            q.flux_model, q.flux_power = H.FLUX_POWER, float(self.model.p)
        if getattr(self, 'mult', None) is not None:
            q.flux_mult, q.flux_mult_stride = self.mult.ptr, self.n
        elif getattr(self, 'taper', None) is not None:
            q.flux_mult, q.flux_mult_stride = self.taper.ptr, 0
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

    @staticmethod
    def _smooth_window(s):
        """Window of the packaged flux smoother when the kernel does the smoothing (tango/lodestro_method.py:359-362), else 1."""
        th = s.compute_all_H_all_fields.turbhandler
        if th.doSmoothing and type(th.fluxSmoother) is _smoother.Smoother and np.shape(s.x)[-1] <= 4096 \
                and th.fluxSmoother.windowSize <= np.shape(s.x)[-1]:
            return int(th.fluxSmoother.windowSize)
        return 1

    def run_timestep(self, s):
        q = self.params(s)
        if s.m > 1:      # new timestep: previous solution becomes u_{m-1}; keep the EWMA memories
            self.um1.copy_from(self.u)
            st = self.status.to_host() & H.ST_SEEDED
            self.status.set(st.astype(np.int32))
            H.check(H.lib.tg_memset(self.iters.ptr, 0, self.iters.nbytes, H.stream()))
            H.check(H.lib.tg_memset(self.err.ptr, 0, self.err.nbytes, H.stream()))
        done = self.start = s.iterationNumber
        hist = s.profilesAllIterations
        # The iterate history, when kept, is written by the kernel into a device buffer (tg_params_1f.iterates) and
        # read back once at the end; a history beyond 256 MiB is streamed to the host iterate by iterate instead.
        rows = s.maxIterations - done
        snaps = None
        if hist is not None and rows > 0 and rows * self.u.nbytes <= (256 << 20):
            snaps = H.DeviceArray((rows, self.nb, self.n))
            q.iterates, q.iterates_rows = snaps.ptr, rows
        stream_hist = hist is not None and snaps is None
        while done < s.maxIterations:
            k = 1 if (stream_hist or self.noise is not None) else min(s.iterationsPerCheck, s.maxIterations - done)
            q.iters_per_launch = k        # k iterations in one launch, the profiles held in registers in between
            if self.noise is not None:    # fresh noise for every evaluation of the flux model (tango/extras/noisyflux.py:43-47)
                self.noise.device_multiplier(self.nb, self.n, out=self.mult, taper=self.taper)
            H.check(H.lib.tg_picard_1f_step(C.byref(self.state), C.byref(q), H.stream()))
            done += k
            if stream_hist:
                snap = self.u.to_host()
                hist[self.label][done - 1 - self.start] = snap if s.batched else snap[0]
            H.check(H.lib.tg_status_reduce(self.status.ptr, self.nb, self.counts.ptr, H.stream()))
            active, conv, invalid, _ = (int(v) for v in self.counts.to_host())
            if active == 0:
                break
        if snaps is not None:
            its = self.iters.to_host()
            top = int(its.max())
            got = np.empty((top, self.nb, self.n))      # only the rows that were written
            if top:
                H.check(H.lib.tg_copy_d2h(got.ctypes.data, snaps.ptr, got.nbytes, H.stream()))
                H.check(H.lib.tg_stream_sync(H.stream()))
            if s.batched:         # rows after a profile's last iteration repeat its final iterate
                final = self.u.to_host()
                late = np.arange(top)[:, None] >= its[None, :]
                got[:top][late] = np.broadcast_to(final, (top,) + final.shape)[late]
                hist[self.label][:top] = got[:top]
            else:
                hist[self.label][:top] = got[:top, 0]
        self.finish(s)

    def finish(self, s):
        status = self.status.to_host()
        iters = self.iters.to_host()
        top = int(iters.max()) if iters.size else 0
        err = np.zeros(self.err.shape)                 # only the columns that were written cross PCIe
        err[:, :top] = self.err.columns_to_host(top)
        u = self.u.to_host()
        s.iterationsPerProfile = iters.astype(np.int64)
        s.countStoredIterations = int(iters.max())
        s.iterationNumber = getattr(self, 'start', 0) + s.countStoredIterations
        s.converged = bool(np.all(status & H.ST_CONVERGED))
        s.solutionError = bool(np.any(status & H.ST_INVALID))
        s.profileStatus = status
        if s.batched:
            s.errHistory = err
            s.profiles = {self.label: u}
        else:
            s.errHistory = err[0].copy()
            s.profiles = {self.label: u[0]}
        m = s.fields[0].lodestroMethod
        if s.batched:      # an ensemble's memories stay on the device until someone reads them
            m.set_ewma_iterates(self.uE.to_host, self.DE.to_host, self.cE.to_host)
        else:
            m.set_ewma_iterates(self.uE.to_host()[0], self.DE.to_host()[0], self.cE.to_host()[0])


def _torch_view(array, device):
    import torch
    return torch.as_tensor(array, device=device)


class _FusedExternal1F(_Fused1F):
    """One field whose turbulent flux comes from a Python flux model (the place of a turbulence code): per
    iteration the profile EWMA is advanced and handed to fluxModel.get_flux on the host, the (optionally smoothed)
    flux goes back to the device and one launch of tg_picard_1f_step does the split, the EWMA of D and c, geometry
    factors, assembly, residual, solve and validity check (tango/lodestro_method.py:327-408 + tango/solver.py:188-272).
    Needs a compute_all_H callback marked `static_H = True`: its own H coefficients (H1 and optionally H7) depend on x
    only, so they are evaluated once."""

    _turb_on_transport_grid = True      # the flux model, the split and the EWMA memories live on the transport grid

    @classmethod
    def plan(cls, s):
        f = s.fields[0]
        Hs = f.compute_all_H(s.tArray[0], s.x, dict(s.profiles), _mf.HCoefficients())
        if Hs.H1 is None or any(getattr(Hs, k) is not None for k in ('H2', 'H3', 'H4', 'H6', 'H8')):
            return None
        return cls(s, Hs)

    def __init__(self, s, Hs):
        f = s.fields[0]
        self.label = f.label
        u0 = np.atleast_2d(np.asarray(s.profiles[f.label], dtype=np.float64))
        self.nb, self.n = nb, n = u0.shape
        D = H.DeviceArray
        from .ensemble import device_arrays
        self.u, self.um1, self.uE, self.DE, self.cE, self.flux = device_arrays([((nb, n), np.float64)] * 6, zero=True)
        self.u.set(u0)
        self.um1.set(np.atleast_2d(np.asarray(f.profile_mminus1, dtype=np.float64)))
        self.status = D((nb,), np.int32, zero=True)
        self.iters = D((nb,), np.int32, zero=True)
        self.err = D((nb, s.maxIterations), zero=True)
        self.counts = D((4,), np.int64, zero=True)
        self.UL = D.from_host(np.broadcast_to(np.asarray(f.rightBC, dtype=np.float64), (nb,)))
        self.S0 = self.src_cut = None
        self.uE_host = None
        m = f.lodestroMethod
        if self._turb_on_transport_grid and m._EWMAProfile._yEWMA_lminus1 is not None:
            self.uE_host = np.atleast_2d(np.array(m._EWMAProfile._yEWMA_lminus1, dtype=np.float64))
            self.uE.set(self.uE_host)
            self.DE.set(np.atleast_2d(m._EWMATurbD._yEWMA_lminus1))
            self.cE.set(np.atleast_2d(m._EWMATurbc._yEWMA_lminus1))
            self.status.set(np.full((nb,), H.ST_SEEDED, np.int32))
        self.state = H.State1F(nb, n, self.u.ptr, self.um1.ptr, self.uE.ptr, self.DE.ptr, self.cE.ptr,
                               self.status.ptr, self.iters.ptr, self.err.ptr, s.maxIterations)

        def grid_array(v):       # (n,) device array of a coefficient shared by all profiles
            v = np.asarray(v, dtype=np.float64)
            if v.ndim > 1:
                raise ValueError('geometry and H1 coefficients must be one array for the whole ensemble')
            return D.from_host(np.broadcast_to(v, (n,)).copy())

        self._frozen_H = Hs
        self.H1 = grid_array(Hs.H1)
        self.H7, self.h7_stride = None, 0
        if Hs.H7 is not None:
            h7 = np.asarray(Hs.H7, dtype=np.float64)
            self.H7 = D.from_host(np.ascontiguousarray(h7))
            self.h7_stride = n if h7.ndim == 2 else 0
        th = s.compute_all_H_all_fields.turbhandler
        sp = m._fluxSplitter
        self.Vprime = self.gxx = self.gradx = None
        if th.isNonCartesian:
            self.Vprime, self.gxx, self.gradx = grid_array(th.VprimeTango), grid_array(th.gxxAvgTango), grid_array(th.gradxAvgTango)
        self.gxx_turb = self.gradx_turb = None
        if self._turb_on_transport_grid:
            self.gxx_turb = grid_array(sp.gxxAvg) if (np.ndim(sp.gxxAvg) or sp.gxxAvg != 1) else None
            self.gradx_turb = grid_array(sp.gradxAvg) if (np.ndim(sp.gradxAvg) or sp.gradxAvg != 1) else None

    def _verify_static(self, s):
        """A callback that was found static by trial calls only (no static_H attribute) is called again at every convergence
        check with the current time and profiles: if its own coefficients are not the ones frozen at the start, the
        trial misjudged it (a threshold, a switch in time, internal state) and the run stops rather than go on with
        coefficients the reference would not use."""
        if not getattr(s, '_static_by_trial', False):
            return
        f = s.fields[0]
        u = self.u.to_host()
        now = f.compute_all_H(s.t, s.x, {self.label: u if s.batched else u[0]}, _mf.HCoefficients())
        for k in ('H1', 'H2', 'H3', 'H4', 'H6', 'H7', 'H8'):
            a, b = getattr(now, k), getattr(self._frozen_H, k)
            if (a is None) != (b is None) or (a is not None and not np.array_equal(np.asarray(a), np.asarray(b))):
                raise RuntimeError('compute_all_H of field %r looked independent of the profile and of time in two trial calls, but '
                                   'its %s has changed at t = %g: construct the Solver with detectStaticH=False (or give the '
                                   'callback an attribute static_H = False)' % (self.label, k, s.t))

    def params(self, s):
        q = _Fused1F.params(self, s)
        p = lambda a: a.ptr if a is not None else None
        q.S0 = q.src_cut = None
        q.H1, q.H7, q.h7_stride = p(self.H1), p(self.H7), self.h7_stride
        if self.H7 is None:          # no source: a zero array (the kernel's box source needs S0 and src_cut)
            self.H7 = H.DeviceArray((self.n,), zero=True)
            q.H7, q.h7_stride = self.H7.ptr, 0
        q.Vprime, q.gxx, q.gradx = p(self.Vprime), p(self.gxx), p(self.gradx)
        q.gxx_turb, q.gradx_turb = p(self.gxx_turb), p(self.gradx_turb)
        q.flux_model, q.flux = H.FLUX_EXTERNAL, self.flux.ptr
        return q

    def run_timestep(self, s):
        q = self.params(s)
        q.iters_per_launch = 1
        th = s.compute_all_H_all_fields.turbhandler
        a_prof = s.fields[0].lodestroMethod.get_ewma_params()[1]
        if s.m > 1:
            self.um1.copy_from(self.u)
            self.status.set((self.status.to_host() & H.ST_SEEDED).astype(np.int32))
            H.check(H.lib.tg_memset(self.iters.ptr, 0, self.iters.nbytes, H.stream()))
            H.check(H.lib.tg_memset(self.err.ptr, 0, self.err.nbytes, H.stream()))
        done = self.start = s.iterationNumber
        hist = s.profilesAllIterations
        host_smoothing = th.doSmoothing and q.smooth_window <= 1       # a smoother the kernel does not implement
        if getattr(th.fluxModel, 'device_flux', False) and not host_smoothing:
            return self._run_timestep_device_flux(s, q, th, a_prof)
        pick = (lambda a: a) if s.batched else (lambda a: a[0])
        # Host side of the protocol, per iteration: the profile EWMA of the members that still iterate is advanced on the
        # device (tg_ewma_profiles), comes to the host through a page-locked buffer, the flux model (and a smoother the
        # kernel does not implement) runs on it, the flux goes back through another page-locked buffer.  Whether anything
        # still iterates is looked up every iterationsPerCheck iterations only.
        if getattr(self, '_pinned', None) is None:
            self._pinned = (H.PinnedArray((self.nb, self.n)), H.PinnedArray((self.nb, self.n)))
        uE_pin, flux_pin = self._pinned
        while done < s.maxIterations:
            if (done - self.start) % s.iterationsPerCheck == 0:
                H.check(H.lib.tg_status_reduce(self.status.ptr, self.nb, self.counts.ptr, H.stream()))
                if int(self.counts.to_host()[0]) == 0:
                    break
                self._verify_static(s)
            H.check(H.lib.tg_ewma_profiles(self.u.ptr, self.uE.ptr, float(a_prof), self.status.ptr, self.nb, self.n, H.stream()))
            uE_host = uE_pin.download(self.uE)
            flux = np.asarray(th.fluxModel.get_flux({self.label: pick(uE_host)})[self.label], dtype=np.float64)
            if host_smoothing:
                flux = np.asarray(th.fluxSmoother.smooth(flux), dtype=np.float64)
            flux_pin.array[...] = np.atleast_2d(flux)
            flux_pin.upload(self.flux)
            H.check(H.lib.tg_picard_1f_step(C.byref(self.state), C.byref(q), H.stream()))
            done += 1
            if hist is not None:
                hist[self.label][done - 1 - self.start] = pick(self.u.to_host())
                if s.fluxesAllIterations is not None:
                    s.fluxesAllIterations[self.label][done - 1 - self.start] = flux
        self._uE_on_device = True
        self.finish(s)

    def _run_timestep_device_flux(self, s, q, th, a_prof):
        """The same loop for a flux model that lives on the GPU (attribute device_flux = True, e.g. a surrogate model
        in PyTorch): get_flux receives the EWMA of the profiles as a CUDA torch tensor viewing this solver's memory
        ((profiles, N), or (N,) for a single profile) and returns the flux as a CUDA tensor; nothing crosses PCIe
        inside the iteration.  The kernel is launched on torch's current stream for the duration of the timestep."""
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError('a flux model with device_flux = True needs torch to see the GPU: import torch before tango_amd, '
                               'so that the library shares the HIP runtime torch ships with')
        dev = torch.device('cuda', torch.cuda.current_device())
        u_t, uE_t, fl_t, st_t = (_torch_view(a, dev) for a in (self.u, self.uE, self.flux, self.status))
        hist = s.profilesAllIterations
        rows = s.maxIterations - self.start
        snaps = None
        if hist is not None and rows > 0 and rows * self.u.nbytes <= (256 << 20):
            snaps = H.DeviceArray((rows, self.nb, self.n))
        previous = H.stream()
        H.set_stream(torch.cuda.current_stream().cuda_stream)
        try:
            done = self.start
            while done < s.maxIterations:
                if (done - self.start) % s.iterationsPerCheck == 0:
                    H.check(H.lib.tg_status_reduce(self.status.ptr, self.nb, self.counts.ptr, H.stream()))
                    if int(self.counts.to_host()[0]) == 0:
                        break
                    self._verify_static(s)
                # EWMA of the profiles that still iterate (tango/lodestro_method.py:345), the first one seeding the memory:
                # one kernel on the same stream, predicated on the status word
                H.check(H.lib.tg_ewma_profiles(self.u.ptr, self.uE.ptr, float(a_prof), self.status.ptr, self.nb, self.n, H.stream()))
                self._uE_on_device = True
                flux = th.fluxModel.get_flux({self.label: uE_t if s.batched else uE_t[0]})[self.label]
                fl_t.copy_(torch.as_tensor(flux, device=dev).reshape(self.nb, self.n))
                H.check(H.lib.tg_picard_1f_step(C.byref(self.state), C.byref(q), H.stream()))
                done += 1
                if snaps is not None:
                    H.check(H.lib.tg_copy_d2d(snaps.ptr + (done - 1 - self.start) * self.u.nbytes, self.u.ptr, self.u.nbytes, H.stream()))
                elif hist is not None:
                    snap = self.u.to_host()
                    hist[self.label][done - 1 - self.start] = snap if s.batched else snap[0]
            if snaps is not None:
                top = done - self.start
                got = np.empty((top, self.nb, self.n))
                if top:
                    H.check(H.lib.tg_copy_d2h(got.ctypes.data, snaps.ptr, got.nbytes, H.stream()))
                    H.check(H.lib.tg_stream_sync(H.stream()))
                hist[self.label][:top] = got if s.batched else got[:, 0]
            self.finish(s)
        finally:
            H.set_stream(previous)


class _FusedMapped1F(_FusedExternal1F):
    """One field whose turbulence lives on a grid of its own (gridMapper = interfacegrids_gene.GridInterfaceTangoOutside,
    TangoOutsideExtrapCoeffs or TangoOutsideExtrapCoeffsBothSides: the transport grid reaches beyond the turbulence grid at
    both ends; the latter two replace the coefficients beyond the trusted part of the turbulence grid by a fitted polynomial;
    or GridInterfaceTangoInside / GridInterfaceTangoInsideFixedOutside: the turbulence grid reaches beyond the outer end of
    the transport grid and the profile is continued there by a straight line of fitted or imposed slope).  Every step of tango/lodestro_method.py:327-388 stays
    on the device: spline of the profile onto the turbulence grid (tg_spline_resample), its EWMA (tg_ewma_profiles), the
    flux model (built-in Shestakov flux, a model on the GPU, or a Python model on the host through page-locked buffers),
    smoothing (tg_moving_average), the split (tg_flux_split), the EWMA of D and c on the turbulence grid, the spline of
    both back onto the transport grid with zeros outside the turbulence grid, and the fused step with those coefficients
    (tg_picard_1f_step, frozen_turb: assembly, residual, solve, validity check).  The EWMA memories live on the
    turbulence grid, as in the reference."""

    _turb_on_transport_grid = False

    @classmethod
    def plan(cls, s):
        from . import interfacegrids_gene as _ig
        f = s.fields[0]
        th = s.compute_all_H_all_fields.turbhandler
        m = f.lodestroMethod
        if type(f.gridMapper) not in (_ig.GridInterfaceTangoOutside, _ig.TangoOutsideExtrapCoeffs, _ig.TangoOutsideExtrapCoeffsBothSides,
                                      _ig.GridInterfaceTangoInside, _ig.GridInterfaceTangoInsideFixedOutside) or type(m) is not _lm.lm:
            return None
        if m._fluxSplitter.ftheta != m._fluxSplitter._default_ftheta or f.coupledTo is not None:
            return None
        if s.useInnerIteration or s.useTreadLightly or s.useTreadInitially:
            return None
        if not _callback_is_static(s):
            return None
        xg = np.asarray(f.gridMapper.get_x_turbulence_grid(), dtype=np.float64)
        if xg.size < 4 or np.shape(s.x)[-1] < 4 or not np.array_equal(np.asarray(f.gridMapper.get_x_transport_grid()), np.asarray(s.x)):
            return None
        if abs((xg[1] - xg[0]) - th.dxTurbGrid) > 1e-12 * abs(th.dxTurbGrid):
            return None
        Hs = f.compute_all_H(s.tArray[0], s.x, dict(s.profiles), _mf.HCoefficients())
        if Hs.H1 is None or any(getattr(Hs, k) is not None for k in ('H2', 'H3', 'H4', 'H6', 'H8')):
            return None
        return cls(s, Hs)

    def __init__(self, s, Hs):
        _FusedExternal1F.__init__(self, s, Hs)
        f = s.fields[0]
        D = H.DeviceArray
        xT = np.ascontiguousarray(s.x, dtype=np.float64)
        xG = np.ascontiguousarray(f.gridMapper.get_x_turbulence_grid(), dtype=np.float64)
        self.nt = nt = xG.size
        self.xT, self.xG = D.from_host(xT), D.from_host(xG)
        self.lo, self.hi, self.zones = self._back_map(f.gridMapper, xT, xG)
        self.fwd_lo, self.fwd_hi, self.fwd_nonneg, self.fwd_zones = self._forward_map(f.gridMapper, xT, xG)
        from .ensemble import device_arrays
        self.pT, self.uE_t, self.flux_t, self.D_t, self.c_t, self.DE_t, self.cE_t = device_arrays([((self.nb, nt), np.float64)] * 7, zero=True)
        wb = max(int(H.lib.tg_spline_work_bytes(self.nb, self.n)), int(H.lib.tg_spline_work_bytes(self.nb, nt)))
        self.spline_work = D((wb // 8,))
        sp = f.lodestroMethod._fluxSplitter
        arr = lambda v: D.from_host(np.broadcast_to(np.asarray(v, dtype=np.float64), (nt,)).copy())
        self.gxx_t = arr(sp.gxxAvg) if (np.ndim(sp.gxxAvg) or sp.gxxAvg != 1) else None
        self.gradx_t = arr(sp.gradxAvg) if (np.ndim(sp.gradxAvg) or sp.gradxAvg != 1) else None
        self.gxx_turb = self.gradx_turb = None      # the split runs on the turbulence grid, outside the step kernel
        m = f.lodestroMethod
        if m._EWMAProfile._yEWMA_lminus1 is not None:      # restart: memories on the turbulence grid
            self.uE_t.set(np.atleast_2d(m._EWMAProfile._yEWMA_lminus1))
            self.DE_t.set(np.atleast_2d(m._EWMATurbD._yEWMA_lminus1))
            self.cE_t.set(np.atleast_2d(m._EWMATurbc._yEWMA_lminus1))
        self.seeded = m._EWMAProfile._yEWMA_lminus1 is not None

    @staticmethod
    def _forward_map(mapper, xT, xG):
        """How the profile goes onto the turbulence grid: the range [lo, hi) of turbulence points that take the spline
        through the transport-grid values, whether negative values are cut off, and for a turbulence grid that reaches
        beyond the transport grid the stretches that continue the profile by the straight line through its last point with
        the least-squares slope of its last ten points (tango/interfacegrids_gene.py:545-604): that line is linear in the
        data, value(x) = y_last + (x - x_last) sum_z d_z (y_z - y_last) / sum_z d_z^2 with d_z = x_z - x_last, so it is
        applied as a fixed matrix over those ten points.  Zones: (first input point, input points, matrix, first output point,
        output points, bias or None)."""
        from . import interfacegrids_gene as _ig
        D = H.DeviceArray
        if type(mapper) is _ig.GridInterfaceTangoInsideFixedOutside:
            # spline on the closed range of the transport grid, straight lines of the imposed slope through the end points
            # outside it (:636-662): the end point's value plus slope * (x - x_end)
            inside = np.flatnonzero((xG >= xT[0]) & (xG <= xT[-1]))
            lo, hi = (int(inside[0]), int(inside[-1]) + 1) if inside.size else (0, 0)
            zones = []
            for out, end in ((np.arange(0, lo), 0), (np.arange(hi, xG.size), xT.size - 1)):
                if out.size:
                    bias = mapper.outwardSlopeGene * (xG[out] - xT[end])
                    zones.append((end, 1, D.from_host(np.ones((out.size, 1))), int(out[0]), int(out.size), D.from_host(np.ascontiguousarray(bias))))
            return lo, hi, True, zones
        if type(mapper) is not _ig.GridInterfaceTangoInside:
            return 0, xG.size, False, []
        inside = np.flatnonzero((xG > xT[0]) & (xG < xT[-1]))
        lo, hi = (int(inside[0]), int(inside[-1]) + 1) if inside.size else (0, 0)
        pts = min(10, xT.size)
        d = xT[-pts:] - xT[-1]
        w = d / np.dot(d, d)
        zones = []
        for out in (np.arange(0, lo), np.arange(hi, xG.size)):
            if out.size:
                M = (xG[out] - xT[-1])[:, None] * w[None, :]
                M[:, -1] += 1.0 - (xG[out] - xT[-1]) * w.sum()
                zones.append((xT.size - pts, pts, D.from_host(np.ascontiguousarray(M)), int(out[0]), int(out.size), None))
        return lo, hi, True, zones

    @staticmethod
    def _back_map(mapper, xT, xG):
        """How D and c come back onto the transport grid: the range [lo, hi) of transport points that take the spline
        through the turbulence-grid values (zero elsewhere), and for the extrapolating mappers the stretches that take a
        polynomial fitted to the points of a zone instead, each as (first zone point, zone points, matrix on the device,
        first output point, output points).  The fit + evaluation of np.polyfit / np.polyval is linear in the data: its
        matrix is built here once, by the same two NumPy routines applied to unit vectors."""
        from . import interfacegrids_gene as _ig

        def fit_matrix(z0, z1, out):
            zone = np.flatnonzero((xG >= z0) & (xG <= z1))
            if zone.size == 0 or out.size == 0:
                return None
            coef = np.polyfit(xG[zone], np.eye(zone.size), mapper.polynomialDegree)        # (degree + 1, zone points)
            M = np.stack([np.polyval(coef[:, z], xT[out]) for z in range(zone.size)], axis=1)
            return (int(zone[0]), int(zone.size), H.DeviceArray.from_host(np.ascontiguousarray(M)), int(out[0]), int(out.size))

        if type(mapper) is _ig.GridInterfaceTangoOutside:
            # the transport points strictly inside the turbulence grid (tango/interfacegrids_gene.py:446-484)
            return int(np.where(xT > xG[0])[0][0]), int(np.where(xT < xG[-1])[0][-1]) + 1, []
        if type(mapper) in (_ig.GridInterfaceTangoInside, _ig.GridInterfaceTangoInsideFixedOutside):
            # the transport points to the right of the first turbulence point (:486-526)
            return int(np.where(xT > xG[0])[0][0]), xT.size, []
        if type(mapper) is _ig.TangoOutsideExtrapCoeffs:
            right_edge = mapper.xExtrapZoneRight
            fits = [(mapper.xExtrapZoneLeft, mapper.xExtrapZoneRight, np.flatnonzero(xT > right_edge))]
        else:       # both sides: the inner fit is applied after the spline and replaces it where the two overlap (:349-360)
            right_edge = mapper.xOuterExtrapZoneRight
            fits = [(mapper.xInnerExtrapZoneLeft, mapper.xInnerExtrapZoneRight, np.flatnonzero(xT < mapper.xInnerExtrapZoneLeft)),
                    (mapper.xOuterExtrapZoneLeft, mapper.xOuterExtrapZoneRight, np.flatnonzero(xT > right_edge))]
        inside = np.flatnonzero((xT >= xG[0]) & (xT <= right_edge))
        lo, hi = (int(inside[0]), int(inside[-1]) + 1) if inside.size else (0, 0)
        zones = [z for z in (fit_matrix(*fit) for fit in fits) if z is not None]
        return lo, hi, zones

    def _onto_transport_grid(self, src, nonneg, out):
        """D or c from the turbulence grid onto the transport grid (tango/lodestro_method.py:372)."""
        self._resample(self.xG, src, self.nt, self.xT, self.n, self.lo, self.hi, nonneg, out)
        for z_lo, z_n, M, o_lo, rows in self.zones:
            H.check(H.lib.tg_zone_apply(src.ptr, self.nb, self.nt, z_lo, z_n, M.ptr, None, out.ptr, self.n, o_lo, rows, int(nonneg), H.stream()))

    def params(self, s):
        q = _FusedExternal1F.params(self, s)
        q.flux_model, q.flux, q.smooth_window = H.FLUX_SHESTAKOV, None, 1      # the step gets D and c ready-made
        q.frozen_turb, q.iters_per_launch = 1, 1
        return q

    def _resample(self, xin, yin, nin, xout, nout, lo, hi, nonneg, out):
        H.check(H.lib.tg_spline_resample(xin.ptr, yin.ptr, self.nb, nin, xout.ptr, nout, lo, hi, int(nonneg), out.ptr,
                                         self.spline_work.ptr, self.spline_work.nbytes, H.stream()))

    def run_timestep(self, s):
        q = self.params(s)
        th = s.compute_all_H_all_fields.turbhandler
        f = s.fields[0]
        m = f.lodestroMethod
        a_turb, a_prof = m.get_ewma_params()
        tp = m._fluxSplitter.thetaParams
        if s.m > 1:
            self.um1.copy_from(self.u)
            self.status.set((self.status.to_host() & H.ST_SEEDED).astype(np.int32))
            H.check(H.lib.tg_memset(self.iters.ptr, 0, self.iters.nbytes, H.stream()))
            H.check(H.lib.tg_memset(self.err.ptr, 0, self.err.nbytes, H.stream()))
        if self.seeded:
            H.check(H.lib.tg_status_or(self.status.ptr, self.nb, H.ST_SEEDED, H.stream()))
        done = self.start = s.iterationNumber
        hist = s.profilesAllIterations
        pick = (lambda a: a) if s.batched else (lambda a: a[0])
        model = th.fluxModel
        builtin = getattr(model, 'fused_flux_model', None) == 'shestakov' and model.dx == th.dxTurbGrid
        on_device = getattr(model, 'device_flux', False)
        window = int(th.fluxSmoother.windowSize) if (th.doSmoothing and type(th.fluxSmoother) is _smoother.Smoother) else 1
        host_smoothing = th.doSmoothing and window <= 1
        if on_device:
            import torch
            dev = torch.device('cuda', torch.cuda.current_device())
            uE_view, fl_view = _torch_view(self.uE_t, dev), _torch_view(self.flux_t, dev)
            previous = H.stream()
            H.set_stream(torch.cuda.current_stream().cuda_stream)
        elif not builtin and getattr(self, '_pinned_t', None) is None:
            self._pinned_t = (H.PinnedArray((self.nb, self.nt)), H.PinnedArray((self.nb, self.nt)))
        try:
            while done < s.maxIterations:
                if (done - self.start) % s.iterationsPerCheck == 0:
                    H.check(H.lib.tg_status_reduce(self.status.ptr, self.nb, self.counts.ptr, H.stream()))
                    if int(self.counts.to_host()[0]) == 0:
                        break
                    self._verify_static(s)
                # profile onto the turbulence grid and its EWMA there (tango/lodestro_method.py:344-345)
                self._resample(self.xT, self.u, self.n, self.xG, self.nt, self.fwd_lo, self.fwd_hi, self.fwd_nonneg, self.pT)
                for z_lo, z_n, M, o_lo, rows, bias in self.fwd_zones:
                    H.check(H.lib.tg_zone_apply(self.u.ptr, self.nb, self.n, z_lo, z_n, M.ptr, H.ptr_of(bias), self.pT.ptr, self.nt, o_lo, rows,
                                                int(self.fwd_nonneg), H.stream()))
                H.check(H.lib.tg_ewma_profiles(self.pT.ptr, self.uE_t.ptr, float(a_prof), self.status.ptr, self.nb, self.nt, H.stream()))
                # flux on the turbulence grid (:350), smoothing (:359-362)
                if builtin:
                    H.check(H.lib.tg_shestakov_flux(self.uE_t.ptr, float(th.dxTurbGrid), self.flux_t.ptr, self.nb, self.nt, H.stream()))
                    flux_dev = self.flux_t
                elif on_device:
                    flux = model.get_flux({self.label: uE_view if s.batched else uE_view[0]})[self.label]
                    fl_view.copy_(torch.as_tensor(flux, device=dev).reshape(self.nb, self.nt))
                    flux_dev = self.flux_t
                else:
                    uE_pin, fl_pin = self._pinned_t
                    flux = np.asarray(model.get_flux({self.label: pick(uE_pin.download(self.uE_t))})[self.label], dtype=np.float64)
                    if host_smoothing:
                        flux = np.asarray(th.fluxSmoother.smooth(flux), dtype=np.float64)
                    fl_pin.array[...] = np.atleast_2d(flux)
                    fl_pin.upload(self.flux_t)
                    flux_dev = self.flux_t
                if window > 1:
                    H.check(H.lib.tg_moving_average(flux_dev.ptr, window, self.D_t.ptr, self.nb, self.nt, H.stream()))
                    H.check(H.lib.tg_copy_d2d(self.flux_t.ptr, self.D_t.ptr, self.flux_t.nbytes, H.stream()))
                # split and EWMA of D and c on the turbulence grid (:365-369)
                H.check(H.lib.tg_flux_split(self.flux_t.ptr, self.uE_t.ptr, float(th.dxTurbGrid), float(tp['Dmin']), float(tp['Dmax']),
                                            float(tp['dpdxThreshold']), H.ptr_of(self.gxx_t), H.ptr_of(self.gradx_t), self.D_t.ptr,
                                            self.c_t.ptr, None, None, None, self.nb, self.nt, H.stream()))
                H.check(H.lib.tg_ewma_profiles(self.D_t.ptr, self.DE_t.ptr, float(a_turb), self.status.ptr, self.nb, self.nt, H.stream()))
                H.check(H.lib.tg_ewma_profiles(self.c_t.ptr, self.cE_t.ptr, float(a_turb), self.status.ptr, self.nb, self.nt, H.stream()))
                # back onto the transport grid (:372), D kept non-negative (tango/interfacegrids_gene.py:59-64)
                self._onto_transport_grid(self.DE_t, True, self.DE)
                self._onto_transport_grid(self.cE_t, False, self.cE)
                if not self.seeded:
                    H.check(H.lib.tg_status_or(self.status.ptr, self.nb, H.ST_SEEDED, H.stream()))
                    self.seeded = True
                H.check(H.lib.tg_picard_1f_step(C.byref(self.state), C.byref(q), H.stream()))
                done += 1
                if hist is not None:
                    hist[self.label][done - 1 - self.start] = pick(self.u.to_host())
        finally:
            if on_device:
                H.set_stream(previous)
        self.finish(s)

    def finish(self, s):
        keep = (self.uE, self.DE, self.cE)
        self.uE, self.DE, self.cE = self.uE_t, self.DE_t, self.cE_t      # the memories the caller sees live on the turbulence grid
        try:
            _Fused1F.finish(self, s)
        finally:
            self.uE, self.DE, self.cE = keep


class _FusedExternal3F(object):
    """One field solved on its own plus a pair of coupled fields (the grouping of tango/fieldgroups.py:275-331 for e.g.
    n, pi <-> pe) whose flux model and compute_all_H callbacks are arbitrary Python: per iteration the EWMA of the three
    profiles is advanced on the device (tg_ewma_profiles) and handed to fluxModel.get_flux on the host, the callbacks are
    called with the current profiles for their H6, H7, H8 (e.g. the collision terms H6 = -nu, H8 = +nu of
    examples/shestakov_threefield_nonlinearnu.py:108-149), fluxes and coefficients go back through page-locked buffers,
    and ONE launch of tg_picard_3f_step (TG_FLUX_EXTERNAL) does the three splits, the EWMA of D and c, the assembly, the
    residual, the tridiagonal and the 2x2 block-tridiagonal solve and the validity check
    (tango/lodestro_method.py:327-408, tango/fieldgroups.py:172-256, tango/JK_to_matrix_fd.py:52-166,
    tango/solver.py:188-272), where the generic path makes about ten launches per field group with host arrays in between.
    Callbacks marked static_H = True are evaluated once.  Inner iterations (tango/solver.py:274-309) call the callbacks
    again with the new profiles and relaunch with the turbulent coefficients frozen.

    A flux model on the GPU (device_flux = True) and callbacks marked device_H = True are handed CUDA torch tensors that
    view this solver's memory ((profiles, N), or (N,) for a single profile) and may return CUDA tensors; with both, nothing
    crosses PCIe inside the iteration.

    Taken when: three fields, one uncoupled and two coupled to each other; the same EWMA and theta parameters for all;
    Cartesian geometry on the solver's own grid; callbacks that return H1 = 1 and no H2, H3, H4; no pseudo-timestep
    control.  Anything else stays on the generic path."""

    KEYS = ('H6', 'H7', 'H8')

    @classmethod
    def plan(cls, s):
        from .extras import threefield
        if len(s.fields) != 3 or s.useTreadLightly or s.useTreadInitially:
            return None
        alone = [f for f in s.fields if f.coupledTo is None]
        pair = [f for f in s.fields if f.coupledTo is not None]
        if len(alone) != 1 or len(pair) != 2 or pair[0].coupledTo != pair[1].label or pair[1].coupledTo != pair[0].label:
            return None
        fields = (alone[0], pair[0], pair[1])
        th = s.compute_all_H_all_fields.turbhandler
        if th.dxTurbGrid != s.dx or (getattr(th.fluxModel, 'device_flux', False) and th.doSmoothing):
            return None
        params = [f.lodestroMethod.get_ewma_params() for f in fields]
        thetas = [dict(f.lodestroMethod._fluxSplitter.thetaParams) for f in fields]
        if len(set(params)) != 1 or any(t != thetas[0] for t in thetas):
            return None
        preset = [f.lodestroMethod._EWMAProfile._yEWMA_lminus1 is not None for f in fields]
        if any(preset) and not all(preset):
            return None
        runner = cls.__new__(cls)
        runner._base = threefield.Fused3F.__new__(threefield.Fused3F)
        runner.device_model = bool(getattr(th.fluxModel, 'device_flux', False))
        runner.device_cb = [bool(getattr(f.compute_all_H, 'device_H', False)) for f in fields]
        host_profiles = {f.label: np.asarray(s.profiles[f.label], dtype=np.float64) for f in s.fields}
        dev_profiles = None
        if runner.device_model or any(runner.device_cb):
            import torch
            if not torch.cuda.is_available():
                return None
            dev_profiles = {lab: torch.as_tensor(v, device='cuda') for lab, v in host_profiles.items()}
        try:
            coeffs = runner._call_callbacks(s, fields, host_profiles, s.tArray[0], dev_profiles)
        except Exception:
            return None
        if coeffs is None:
            return None
        runner._init(s, fields, coeffs)
        return runner

    def _call_callbacks(self, s, fields, profiles, t, dev_profiles=None, only=None):
        """The fields' own coefficients for `profiles` (tango/multifield.py:202 without the turbulent part): a list of
        {'H6', 'H7', 'H8'} dicts of float64 arrays (CUDA tensors from callbacks marked device_H) or None, or None if a
        callback returns something the kernel does not take (H1 other than 1; H2, H3, H4; an H8 on the field that is solved
        alone).  only: the indices of the fields to call (the others get None entries)."""
        out = []
        for k, f in enumerate(fields):
            if only is not None and k not in only:
                out.append(None)
                continue
            Hc = f.compute_all_H(t, s.x, dev_profiles if self.device_cb[k] else profiles, _mf.HCoefficients())
            if not isinstance(Hc, _mf.HCoefficients) or Hc.H1 is None:
                return None
            ones = (Hc.H1 == 1.0).all() if hasattr(Hc.H1, 'is_cuda') else np.all(np.asarray(Hc.H1) == 1.0)
            if not bool(ones):
                return None
            if any(getattr(Hc, name) is not None for name in ('H2', 'H3', 'H4')) or (k == 0 and Hc.H8 is not None):
                return None
            row = {}
            for name in self.KEYS:
                v = getattr(Hc, name)
                if v is not None and not hasattr(v, 'is_cuda'):
                    v = np.ascontiguousarray(v, dtype=np.float64)
                row[name] = v
            out.append(row)
        return out

    def _init(self, s, fields, coeffs):
        b = self._base
        b._alloc_state(s, fields)
        self.fields, self.nb, self.n = fields, b.nb, b.n
        nb, n = self.nb, self.n
        self.static_cb = [getattr(f.compute_all_H, 'static_H', False) is True for f in fields]
        self.static = all(self.static_cb)
        from .ensemble import device_arrays
        self.flux = device_arrays([((nb, n), np.float64)] * 3, zero=True)
        # one device array and one page-locked staging buffer per coefficient the callbacks return, in the shape they
        # return it: (n,) for all profiles or (profiles, n)
        self.coef = [{} for _ in fields]
        for k, c in enumerate(coeffs):
            for name, v in c.items():
                if v is None:
                    continue
                shape = (n,) if v.ndim == 1 else (nb, n)
                if v.shape != shape and not (v.ndim == 1 and nb == 1 and v.shape == (n,)):
                    v = np.broadcast_to(v, shape)
                self.coef[k][name] = (H.DeviceArray(shape), H.PinnedArray(shape))
        self._upload_coeffs(coeffs)
        self._pin_in = [H.PinnedArray((nb, n)) for _ in range(3)]
        self._pin_flux = None if self.device_model else [H.PinnedArray((nb, n)) for _ in range(3)]

    def _upload_coeffs(self, coeffs):
        for k, c in enumerate(coeffs):
            if c is None:
                continue
            if sorted(name for name, v in c.items() if v is not None) != sorted(self.coef[k]):
                raise RuntimeError('compute_all_H of field %r returned another set of coefficients than at the start' % self.fields[k].label)
            for name, (dev, pin) in self.coef[k].items():
                v = c[name]
                if hasattr(v, 'is_cuda'):          # a CUDA tensor: device to device on torch's stream (= the library's for this run)
                    import torch
                    _torch_view(dev, v.device).copy_(torch.broadcast_to(v.to(torch.float64), dev.shape))
                else:
                    pin.array[...] = np.broadcast_to(v, pin.shape)
                    pin.upload(dev)

    def params(self, s):
        b = self._base
        m = self.fields[0].lodestroMethod
        a_turb, a_prof = m.get_ewma_params()
        tp = m._fluxSplitter.thetaParams
        q = H.Params3F()
        q.dt, q.dx, q.a_turb, q.a_prof = float(s.dt), float(s.dx), float(a_turb), float(a_prof)
        q.Dmin, q.Dmax, q.dpdx_thr, q.tol = float(tp['Dmin']), float(tp['Dmax']), float(tp['dpdxThreshold']), float(s.tol)
        q.UL = b._ptrs(b.UL)
        q.flux_model = H.FLUX_EXTERNAL
        q.flux = b._ptrs(self.flux)
        for k in range(3):
            for name, stride, ptrs in (('H6', q.h6_stride, q.H6), ('H7', q.h7_stride, q.H7), ('H8', q.h8_stride, q.H8)):
                if name in self.coef[k]:
                    dev = self.coef[k][name][0]
                    ptrs[k] = dev.ptr
                    stride[k] = self.n if len(dev.shape) == 2 else 0
        return q

    def _profiles_on_host(self, s, arrays):
        """The three device arrays as the dict of host arrays the reference's callables take ((N,) for a single profile)."""
        out = {}
        for f, pin, d in zip(self.fields, self._pin_in, arrays):
            a = pin.download(d)
            out[f.label] = a if s.batched else a[0]
        return out

    def _refresh_coeffs(self, s):
        todo = [k for k in range(3) if not self.static_cb[k]]
        on_host = any(not self.device_cb[k] for k in todo)
        coeffs = self._call_callbacks(s, self.fields, self._profiles_on_host(s, self._base.u) if on_host else None, s.t,
                                      getattr(self, '_dev_u', None), only=todo)
        if coeffs is None:
            raise RuntimeError('a compute_all_H callback now returns coefficients the fused three-field step does not take '
                               '(H1 other than 1, or H2, H3, H4): construct the Solver with fusedPaths=False')
        self._upload_coeffs(coeffs)

    def run_timestep(self, s):
        b = self._base
        q = self.params(s)
        th = s.compute_all_H_all_fields.turbhandler
        a_prof = self.fields[0].lodestroMethod.get_ewma_params()[1]
        if s.m > 1:
            for a, c in zip(b.um1, b.u):
                a.copy_from(c)
            b.status.set((b.status.to_host() & H.ST_SEEDED).astype(np.int32))
            H.check(H.lib.tg_memset(b.iters.ptr, 0, b.iters.nbytes, H.stream()))
            H.check(H.lib.tg_memset(b.err.ptr, 0, b.err.nbytes, H.stream()))
        inner = s.innerIterationMaxCount if s.useInnerIteration else 0
        hist = s.profilesAllIterations
        done = start = b.start = s.iterationNumber
        pick = (lambda a: a) if s.batched else (lambda a: a[0])
        previous_stream = H.stream()
        if self.device_model or any(self.device_cb):
            # tensors that view the solver's memory, for a flux model and callbacks that live on the GPU; the library
            # works on torch's current stream for the duration of the timestep
            import torch
            dev = torch.device('cuda', torch.cuda.current_device())
            self._dev_u = {f.label: pick(_torch_view(a, dev)) for f, a in zip(self.fields, b.u)}
            dev_uE = {f.label: pick(_torch_view(a, dev)) for f, a in zip(self.fields, b.uE)}
            dev_flux = [_torch_view(a, dev) for a in self.flux]
            H.set_stream(torch.cuda.current_stream().cuda_stream)
        try:
            self._iterate(s, b, q, th, a_prof, inner, hist, done, start, pick, dev_uE if self.device_model else None,
                          dev_flux if self.device_model else None)
        finally:
            H.set_stream(previous_stream)

    def _iterate(self, s, b, q, th, a_prof, inner, hist, done, start, pick, dev_uE, dev_flux):
        while done < s.maxIterations:
            if (done - start) % s.iterationsPerCheck == 0:
                H.check(H.lib.tg_status_reduce(b.status.ptr, self.nb, b.counts.ptr, H.stream()))
                if int(b.counts.to_host()[0]) == 0:
                    break
            # EWMA of the profiles that still iterate, the flux model on it (tango/lodestro_method.py:342-362)
            for k in range(3):
                H.check(H.lib.tg_ewma_profiles(b.u[k].ptr, b.uE[k].ptr, float(a_prof), b.status.ptr, self.nb, self.n, H.stream()))
            if dev_uE is not None:
                import torch
                fluxes = th.fluxModel.get_flux(dev_uE)
                for f, view in zip(self.fields, dev_flux):
                    view.copy_(torch.as_tensor(fluxes[f.label], device=view.device).reshape(self.nb, self.n))
            else:
                fluxes = th.fluxModel.get_flux(self._profiles_on_host(s, b.uE))
                for f, pin, dev in zip(self.fields, self._pin_flux, self.flux):
                    flux = np.asarray(fluxes[f.label], dtype=np.float64)
                    if th.doSmoothing:
                        flux = np.asarray(th.fluxSmoother.smooth(flux), dtype=np.float64)
                    if hist is not None and s.fluxesAllIterations is not None:
                        s.fluxesAllIterations[f.label][done - start] = flux
                    pin.array[...] = np.atleast_2d(flux)
                    pin.upload(dev)
            # the fields' own coefficients for the current profiles (tango/multifield.py:202)
            if not self.static:
                self._refresh_coeffs(s)
            q.frozen_turb, q.outer_count = 0, 0
            H.check(H.lib.tg_picard_3f_step(C.byref(b.state), C.byref(q), H.stream()))
            done += 1
            q.frozen_turb, q.outer_count = 1, done
            for _ in range(inner):      # tango/solver.py:274-309: turbulent coefficients frozen, the callbacks see the new profiles
                if not self.static:
                    self._refresh_coeffs(s)
                H.check(H.lib.tg_picard_3f_step(C.byref(b.state), C.byref(q), H.stream()))
            if hist is not None:
                for f, d in zip(self.fields, b.u):
                    hist[f.label][done - 1 - start] = pick(d.to_host())
        H.synchronize()
        b.finish(s)
