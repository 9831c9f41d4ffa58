"""Flux-model decorator that damps the flux near both ends of the domain, in the manner of a gyrokinetic code's buffer
zones (tango/extras/bufferzone.py:15-53).  Works on (N,) and (profiles, N) fluxes; the taper is a Tukey window
(the reference calls scipy.signal.tukey, which recent SciPy only keeps as scipy.signal.windows.tukey; the window is
written out here)."""
import numpy as np


def tukey_window(n, alpha):
    """Symmetric Tukey (tapered cosine) window of n points: cosine lobes over the first and last alpha/2 of the
    span, 1 in between; alpha <= 0 is a boxcar, alpha >= 1 a Hann window."""
    if n <= 1:
        return np.ones(n)
    if alpha <= 0:
        return np.ones(n)
    alpha = min(float(alpha), 1.0)
    k = np.arange(n, dtype=float)
    width = int(np.floor(alpha * (n - 1) / 2.0))
    w = np.ones(n)
    left = k[:width + 1]
    w[:width + 1] = 0.5 * (1 + np.cos(np.pi * (-1 + 2.0 * left / alpha / (n - 1))))
    right = k[n - width - 1:]
    w[n - width - 1:] = 0.5 * (1 + np.cos(np.pi * (-2.0 / alpha + 1 + 2.0 * right / alpha / (n - 1))))
    return w


class BufferZone(object):
    def __init__(self, fluxModel, taperwidth):
        """taperwidth: width of the damped region at each boundary as a fraction of the domain."""
        self.fluxModel = fluxModel
        self.taperwidth = taperwidth

    def get_flux(self, profiles):
        fluxes = self.fluxModel.get_flux(profiles)
        return {label: _damp_flux(fluxes[label], self.taperwidth) for label in fluxes}


def _damp_flux(flux, taperwidth):
    flux = np.asarray(flux)
    return flux * tukey_window(flux.shape[-1], 2 * taperwidth)
