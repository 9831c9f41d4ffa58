"""The three-field problem with its flux model and its collision terms on the GPU: a flux model marked device_flux and
compute_all_H callbacks marked device_H receive CUDA tensors that view the solver's memory and return CUDA tensors, the
fused three-field step takes them as outside inputs, and nothing crosses PCIe inside the iteration.  Single profile
against the reference's trajectory, an ensemble against the built-in closure.  Run by
tests/test_gpu_coupled.py::test_threefield_outside_inputs_on_the_device in its own process (torch has to be imported
before the library so that the two share one HIP runtime)."""
import os
import sys

import numpy as np
import torch          # before tango_amd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import tango_amd as tango                      # noqa: E402
from conftest import relerr                    # noqa: E402
from oracle import tango_oracle as orc         # noqa: E402
from tango_amd.extras import threefield as T   # noqa: E402
from test_gpu_coupled import TOL, make_3f_solver     # noqa: E402

calls = {'flux': 0, 'cb': 0}


def ddx(u, dx):
    d = torch.empty_like(u)
    d[..., 1:-1] = (u[..., 2:] - u[..., :-2]) / (2 * dx)
    d[..., 0] = (u[..., 1] - u[..., 0]) / dx
    d[..., -1] = (u[..., -1] - u[..., -2]) / dx
    return d


class TorchFlux(object):
    device_flux = True

    def __init__(self, dx):
        self.dx = dx

    def get_flux(self, p):
        assert all(isinstance(v, torch.Tensor) and v.is_cuda for v in p.values())
        calls['flux'] += 1
        dn, di, de = (ddx(p[k], self.dx) for k in ('n', 'pi', 'pe'))
        D = di ** 2 / p['pi'] ** 2
        return {'n': -D * dn, 'pi': -D * di, 'pe': -D * de}


class SourceOnly(object):
    static_H = True

    def __call__(self, t, x, profiles, HT):
        return tango.multifield.HCoefficients(H1=np.ones_like(x), H7=T.source_n(x)) + HT


class Collisional(object):
    device_H = True

    def __init__(self, nu0, source):
        self.nu0, self.source = nu0, source

    def __call__(self, t, x, p, HT):
        assert p['n'].is_cuda
        calls['cb'] += 1
        nu0 = torch.as_tensor(np.asarray(self.nu0, dtype=float), device=p['n'].device)
        if nu0.ndim:
            nu0 = nu0[..., None]
        nu = nu0 / ((p['pi'] / p['n'] + p['pe'] / p['n']) ** 1.5)
        return tango.multifield.HCoefficients(H1=np.ones_like(x), H6=-nu, H7=self.source(x), H8=nu) + HT


def build(N, nu0, BCs, ic=None, **kw):
    dx = 1 / (N - 1); x = np.arange(N) * dx
    th = {'Dmin': 1e-5, 'Dmax': 1e13, 'dpdxThreshold': 10}
    ic = (1 - 0.5 * x) if ic is None else ic
    lms = [tango.lodestro_method.lm(0.3, 0.3, th) for _ in range(3)]
    F = tango.multifield.Field
    fields = [F(label='n', rightBC=BCs[0], profile_mminus1=1.0 * ic, compute_all_H=SourceOnly(), lodestroMethod=lms[0]),
              F(label='pi', rightBC=BCs[1], profile_mminus1=1.0 * ic, compute_all_H=Collisional(nu0, T.source_i), lodestroMethod=lms[1], coupledTo='pe'),
              F(label='pe', rightBC=BCs[2], profile_mminus1=1.0 * ic, compute_all_H=Collisional(nu0, T.source_e), lodestroMethod=lms[2], coupledTo='pi')]
    tango.multifield.check_fields_initialize(fields)
    call = tango.multifield.ComputeAllHAllFields(fields, tango.lodestro_method.TurbulenceHandler(dx, x, TorchFlux(dx)))
    return tango.solver.Solver(1, x, np.array([0, 1e6]), 1000, 1e-9, call, fields, **kw), lms


def main():
    g = np.load(os.path.join(HERE, 'golden', 'g9_threefield.npz'))
    s, lms = build(500, 2.2, (2, 0.1, 0.3))
    f = s._fused
    assert type(f).__name__ == '_FusedExternal3F' and f.device_model and f.device_cb == [False, True, True] and f.static_cb == [True, False, False]
    before = dict(calls)
    while s.ok:
        s.take_timestep()
    k = int(g['N500_iters'])
    assert s.converged and len(s.errHistoryFinal) == k
    assert calls['flux'] - before['flux'] == k and calls['cb'] - before['cb'] == 2 * k
    assert np.allclose(s.errHistoryFinal, g['N500_err'], rtol=1e-6, atol=5e-12)
    for lab in ('n', 'pi', 'pe'):
        assert relerr(s.profiles[lab], g[f'N500_{lab}']) < TOL
        assert relerr(s.profilesAllIterations[lab][10], g[f'N500_it10_{lab}']) < TOL
    # an ensemble against the built-in closure
    N, nb = 256, 5
    ens = orc.threefield_ensemble(nb, N)
    bcs = tuple(ens['bc'][:, q].copy() for q in range(3))
    sd, _ = build(N, ens['nu0'], bcs, ic=ens['u0'], iterationsPerCheck=4)
    assert sd.batched and type(sd._fused).__name__ == '_FusedExternal3F'
    sd.take_timestep()
    sb, _ = make_3f_solver(tango, N, ens['nu0'], bcs, ic=ens['u0'], iterationsPerCheck=4)
    assert type(sb._fused).__name__ == 'Fused3F'
    sb.take_timestep()
    assert sd.converged and sb.converged and np.array_equal(sd.iterationsPerProfile, sb.iterationsPerProfile)
    for lab in ('n', 'pi', 'pe'):
        assert relerr(sd.profiles[lab], sb.profiles[lab]) < TOL
    print('device coupled case ok')


if __name__ == '__main__':
    main()
