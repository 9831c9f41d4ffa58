"""Flux-model decorators that multiply the flux by 1 + h with h a random perturbation that vanishes at both ends of
the domain (tango/extras/noisyflux.py:13-161): NoisyFlux draws h from an AR(1) process with a given correlation
length, UniformNoisyFlux uses one random number for the whole profile.  Random numbers come from numpy's global
generator in the same order as the reference's, so a seeded run reproduces the reference's noise.  For
(profiles, N) fluxes every profile gets its own, independent, draw (in profile order)."""
import numpy as np


class NoisyFlux(object):
    """device_seed=None (default): the noise is drawn on the host from numpy's global generator, in the reference's
    order.  device_seed=<int>: the noise is generated on the GPU (tg_ar1_noise: counter-based generator keyed by the
    seed, the call number, the profile and the grid point; statistically the same process, not the reference's numbers);
    with a flux model that is built into the fused Picard kernel solver.Solver then keeps the whole iteration on the
    device, the noise entering as the kernel's flux multiplier."""
    flux_decorator = 'noise'

    def __init__(self, fluxModel, amplitude, tac_x, dx, device_seed=None):
        self.fluxModel = fluxModel
        self.amplitude = amplitude
        self.tac = tac_x / dx        # correlation length in grid points
        self.device_seed = device_seed
        self.calls = 0               # sequence number of the next device draw

    @property
    def device_flux(self):
        return self.device_seed is not None and getattr(self.fluxModel, 'device_flux', False)

    def device_multiplier(self, nb, n, out=None, taper=None):
        """(nb, n) device array of 1 + noise (times an optional (n,) device taper) for the next call."""
        from .. import _hip as H
        out = out if out is not None else H.DeviceArray((nb, n))
        H.check(H.lib.tg_ar1_noise(int(self.device_seed), self.calls, float(self.amplitude), float(self.tac), int(0.03 * n),
                                   H.ptr_of(taper), H.ptr_of(out), nb, n, H.stream()))
        self.calls += 1
        return out

    def get_flux(self, profiles):
        fluxes = self.fluxModel.get_flux(profiles)
        if self.device_seed is not None:
            out = {}
            for label, v in fluxes.items():
                if hasattr(v, 'is_cuda') and v.is_cuda:
                    import torch
                    v2 = v.reshape(-1, v.shape[-1])
                    m = torch.as_tensor(self.device_multiplier(v2.shape[0], v2.shape[1]), device=v.device)
                    out[label] = (v2 * m).reshape(v.shape)
                else:
                    v = np.asarray(v, dtype=np.float64)
                    v2 = np.atleast_2d(v)
                    out[label] = (v2 * self.device_multiplier(v2.shape[0], v2.shape[1]).to_host()).reshape(v.shape)
            return out
        return {label: self._add_noise(fluxes[label], self.amplitude, self.tac) for label in fluxes}

    @staticmethod
    def _add_noise(v, amplitude, tac):
        v = np.asarray(v)
        if v.ndim == 2:
            return np.stack([NoisyFlux._add_noise(row, amplitude, tac) for row in v])
        h = dampen_sides(amplitude * ar1noise(len(v), tac))
        return (1 + h) * v


class UniformNoisyFlux(object):
    def __init__(self, fluxModel, amplitude):
        self.fluxModel = fluxModel
        self.amplitude = amplitude

    def get_flux(self, profile):
        flux = self.fluxModel.get_flux(profile)
        if isinstance(flux, dict):       # the reference's version predates dict-valued fluxes (noisyflux.py:77-80)
            return {label: self._add_noise(flux[label], self.amplitude) for label in flux}
        return self._add_noise(flux, self.amplitude)

    @staticmethod
    def _add_noise(v, amplitude):
        v = np.asarray(v)
        if v.ndim == 2:
            return np.stack([UniformNoisyFlux._add_noise(row, amplitude) for row in v])
        h = dampen_sides(np.random.normal(scale=amplitude) * np.ones_like(v))
        return (1 + h) * v


def ar1noise(numSamples, tac):
    """numSamples values of the unit-variance AR(1) process y_{n+1} = lambda y_n + a_n, lambda = (tac-1)/(tac+1),
    after discarding the first m values so that the start value is forgotten (lambda^(2m) < 1/1000)."""
    lamb = (tac - 1) / (tac + 1)
    m = int(-np.log(1000) / (2 * np.log(lamb))) + 1
    white = np.random.normal(scale=np.sqrt(1 - lamb ** 2), size=numSamples + m)
    noise = np.empty_like(white)
    acc = 0.0
    for k in range(white.size):          # the recursion of scipy.signal.lfilter([1], [1, -lambda], white)
        acc = white[k] + lamb * acc
        noise[k] = acc
    return noise[m:]


def dampen_sides(v, numPts=None):
    """Scale v linearly down to zero over numPts points at either end (default 3 % of len(v))."""
    N = len(v)
    if numPts is None:
        numPts = int(0.03 * N)
    assert len(v) > 2 * numPts, 'numPts is too large'
    out = v.copy()
    out[0:numPts] = out[0:numPts] * np.linspace(0, 1, numPts)
    if numPts > 0:
        out[-numPts:] = out[-numPts:] * np.linspace(1, 0, numPts)
    return out
