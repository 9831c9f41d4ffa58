"""A flux model that lives on the GPU (here the Shestakov flux written with torch operations, standing in for a
surrogate model): the profiles' EWMA reaches it as a CUDA tensor viewing the solver's memory, the flux comes back as a
CUDA tensor, nothing crosses PCIe inside the iteration.  Single profile against the reference's trajectory, an ensemble
against the built-in fused path.  Run by tests/test_gpu_picard.py::test_flux_model_on_the_device in its own process."""
import os
import sys

import numpy as np
import torch          # before tango_amd: the library then shares torch's HIP runtime

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import tango_amd as tango                      # noqa: E402
from conftest import relerr                    # noqa: E402
from oracle import tango_oracle as orc         # noqa: E402
from test_gpu_picard import TOL, make_solver, same_stop     # noqa: E402


def main():
    g = np.load(os.path.join(HERE, 'golden', 'g8_onefield.npz'))
    sh = tango.extras.shestakov_nonlinear_diffusion
    seen = []

    class TorchFlux(object):
        device_flux = True

        def __init__(self, dx):
            self.dx = dx

        def get_flux(self, profiles):
            n = profiles['n']
            assert isinstance(n, torch.Tensor) and n.is_cuda and n.dtype == torch.float64
            seen.append(tuple(n.shape))
            d = torch.empty_like(n)
            d[..., 1:-1] = (n[..., 2:] - n[..., :-2]) / (2 * self.dx)
            d[..., 0] = (n[..., 1] - n[..., 0]) / self.dx
            d[..., -1] = (n[..., -1] - n[..., -2]) / self.dx
            return {'n': -(d * d * d) / (n * n)}

    def build(N, nL, S0, delta, n0, a_prof=1.0, **kw):
        dx = 1 / (N - 1); x = np.arange(N) * dx
        lm = tango.lodestro_method.lm(0.3, a_prof, {'Dmin': 1e-5, 'Dmax': 1e13, 'dpdxThreshold': 10})
        f = tango.multifield.Field(label='n', rightBC=nL, profile_mminus1=n0, compute_all_H=sh.ComputeAllH(S0, delta), lodestroMethod=lm)
        tango.multifield.check_fields_initialize([f])
        call = tango.multifield.ComputeAllHAllFields([f], tango.lodestro_method.TurbulenceHandler(dx, x, TorchFlux(dx)))
        return tango.solver.Solver(1, x, np.array([0, 1e4]), 1000, 1e-11, call, [f], **kw), lm

    N = 500
    x = np.arange(N) / (N - 1)
    s, lm = build(N, 1e-2, 1.0, 0.1, 1 - 0.5 * x)
    assert type(s._fused).__name__ == '_FusedExternal1F'
    s.take_timestep()
    k = int(g['N500_p1_iters'])
    assert s.converged and len(s.errHistoryFinal) == k and seen[0] == (N,)
    assert np.allclose(s.errHistoryFinal, g['N500_p1_err'], rtol=1e-6, atol=2e-12)
    assert relerr(s.profiles['n'], g['N500_p1_final']) < TOL
    assert relerr(s.profilesAllIterations['n'][10], g['N500_p1_it10_u']) < TOL
    assert relerr(lm._EWMATurbD._yEWMA_lminus1, g['N500_p1_last_DE']) < TOL
    # an ensemble, profile relaxation on: against the built-in fused path (same arithmetic except the flux expression)
    ens = orc.shestakov_ensemble(6, 512)
    del seen[:]
    sd, _ = build(512, ens['nL'], ens['S0'], ens['delta'], ens['u0'], a_prof=0.4, iterationsPerCheck=4)
    sd.take_timestep()
    sb, _ = make_solver(tango, 512, ens['nL'], ens['S0'], ens['delta'], ens['u0'], 0.3, 0.4, 1e-11, 1000, iterationsPerCheck=4)
    assert type(sb._fused).__name__ == '_Fused1F'
    sb.take_timestep()
    assert sd.converged and sb.converged and seen[0] == (6, 512)
    for b in range(6):
        assert same_stop(int(sd.iterationsPerProfile[b]), int(sb.iterationsPerProfile[b]), sb.errHistory[b], 1e-11, sd.errHistory[b]), b
        assert relerr(sd.profiles['n'][b], sb.profiles['n'][b]) < TOL, b
    print('device flux case ok')


if __name__ == '__main__':
    main()
