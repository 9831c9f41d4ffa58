"""Analytic test problem of Shestakov et al. (2003): nonlinear diffusion with flux
Gamma[n] = -(dn/dx)^3 / n^2, a box source S0 on x < delta, and a closed-form steady state.

Same functions as tango/extras/shestakov_nonlinear_diffusion.py:7-106.  AnalyticFluxModel.get_flux runs
the tg_shestakov_flux kernel; `ComputeAllH` is the H-coefficient callback of examples/shestakov_basic.py:41-53
(H1 = 1, H7 = source) packaged so that solver.Solver can recognise the problem and run it with the fused
device-resident Picard kernel for ensembles with per-profile S0, delta."""
import numpy as np

from .. import _ops, multifield


class AnalyticFluxModel(object):
    """Flux model for the field labelled 'n'; n(1) = nL > 0 keeps the flux finite at the boundary."""
    fused_flux_model = 'shestakov'   # built into the fused Picard kernel under this name

    def __init__(self, dx):
        self.dx = dx

    def get_flux(self, profiles):
        return {'n': get_flux(profiles['n'], self.dx)}


class PowerLawFluxModel(object):
    """Gamma = -|(dn/dx)/n|^p dn/dx, the FluxModel of examples/stripped_down_tango_example.py:55-65 (p = 2 is the
    Shestakov flux).  Built into the fused Picard kernel (TG_FLUX_POWER); get_flux is the host-array form."""
    fused_flux_model = 'power'

    def __init__(self, dx, p=2):
        self.dx = dx
        self.p = p

    def get_flux(self, profiles):
        n = np.asarray(profiles['n'], dtype=float)
        dndx = _ops.ddx(n, self.dx, 1)
        return {'n': -np.abs((dndx / n) ** self.p) * dndx}


def H7contrib_Source(x):
    return GetSource(x)


def get_flux(n, dx):
    return _ops.shestakov_flux(n, dx)


def GetSource(x, S0=1, delta=0.1):
    """S0 where x < delta, 0 elsewhere; S0 and delta may be (profiles,) arrays -> (profiles, N)."""
    x = np.asarray(x, dtype=float)
    S0, delta = np.asarray(S0, dtype=float), np.asarray(delta, dtype=float)
    if S0.ndim or delta.ndim:
        S0 = S0[..., None] if S0.ndim else S0
        delta = delta[..., None] if delta.ndim else delta
    return np.where(x < delta, S0, 0.0) * np.ones_like(x)


def _dxCenteredDifference(u, dx):
    return _ops.ddx(u, dx, 1)


def steady_state_solution(x, nL, S0=1, delta=0.1):
    x = np.asarray(x, dtype=float)
    k = 1 / 3 * (S0 * delta) ** (1 / 3)
    nright = (nL ** (1 / 3) + k * (1 - x)) ** 3
    nleft = (nL ** (1 / 3) + k * (1 - delta + (3 / 4) * (delta - x ** (4 / 3) / delta ** (1 / 3)))) ** 3
    return np.where(x < delta, nleft, nright)


def GetSteadyStateSolution(x, nL, S0=1, delta=0.1):
    return steady_state_solution(x, nL, S0=S0, delta=delta)


class ComputeAllH(object):
    """compute_all_H callback of the one-field problem: H1 = 1, H7 = S0 [x < delta], plus the turbulent
    H2, H3.  S0 and delta may be (profiles,) arrays."""

    static_H = True      # H1 and H7 depend on x only: the fused paths may evaluate them once

    def __init__(self, S0=1, delta=0.1):
        self.S0, self.delta = S0, delta
        self.fused_source = ('box', S0, delta)

    def __call__(self, t, x, profiles, HCoeffsTurb):
        H1 = np.ones_like(x)
        H7 = GetSource(x, self.S0, self.delta)
        return multifield.HCoefficients(H1=H1, H7=H7) + HCoeffsTurb
