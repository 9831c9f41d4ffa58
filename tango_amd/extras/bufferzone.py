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


def tukey_window_device(n, alpha):
    """The same window formed on the GPU (tg_tukey_window): an (n,) tango_amd.DeviceArray."""
    from .. import _hip as H
    w = H.DeviceArray((int(n),))
    H.check(H.lib.tg_tukey_window(w.ptr, int(n), float(alpha), H.stream()))
    return w


class BufferZone(object):
    """With a flux model that is built into the fused Picard kernel (or one that lives on the GPU), solver.Solver
    applies the taper on the device: the window becomes the kernel's flux multiplier (tg_params_1f.flux_mult)."""
    flux_decorator = 'taper'

    def __init__(self, fluxModel, taperwidth):
        """taperwidth: width of the damped region at each boundary as a fraction of the domain."""
        self.fluxModel = fluxModel
        self.taperwidth = taperwidth

    @property
    def device_flux(self):
        return getattr(self.fluxModel, 'device_flux', False)

    def get_flux(self, profiles):
        fluxes = self.fluxModel.get_flux(profiles)
        return {label: _damp_flux(fluxes[label], self.taperwidth) for label in fluxes}


def _is_cuda_tensor(a):
    return hasattr(a, 'is_cuda') and a.is_cuda


def _damp_flux(flux, taperwidth):
    if _is_cuda_tensor(flux):           # a flux model on the GPU: the window is formed there too
        import torch
        w = tukey_window_device(flux.shape[-1], 2 * taperwidth)
        return flux * torch.as_tensor(w, device=flux.device)
    flux = np.asarray(flux)
    return flux * tukey_window(flux.shape[-1], 2 * taperwidth)
