"""Per-iteration wall time of the Solver paths whose flux model lives outside the kernel, at the metric ensemble
(65 536 x 512): a NumPy flux model on the host (two page-locked PCIe copies per iteration) and a torch flux model on
the GPU.  Prints one JSON line per path."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.getcwd())
import tango_amd as tango
from tango_amd.ensemble import shestakov_ensemble

nb, n = 65536, 512
ens = shestakov_ensemble(nb, n)
sh = tango.extras.shestakov_nonlinear_diffusion


class HostFlux(object):
    def __init__(self, dx):
        self.dx = dx

    def get_flux(self, profiles):
        p = profiles['n']
        d = np.empty_like(p)
        d[..., 1:-1] = (p[..., 2:] - p[..., :-2]) / (2 * self.dx)
        d[..., 0] = (p[..., 1] - p[..., 0]) / self.dx
        d[..., -1] = (p[..., -1] - p[..., -2]) / self.dx
        return {'n': -(d * d * d) / (p * p)}


class TorchFlux(HostFlux):
    device_flux = True

    def get_flux(self, profiles):
        p = profiles['n']
        d = torch.empty_like(p)
        d[..., 1:-1] = (p[..., 2:] - p[..., :-2]) / (2 * self.dx)
        d[..., 0] = (p[..., 1] - p[..., 0]) / self.dx
        d[..., -1] = (p[..., -1] - p[..., -2]) / self.dx
        return {'n': -(d * d * d) / (p * p)}


for name, Model, its in (('numpy flux model on the host', HostFlux, 16), ('torch flux model on the GPU', TorchFlux, 16),
                         ('torch flux model on the GPU', TorchFlux, 128)):
    lm = tango.lodestro_method.lm(0.3, 0.3, {'Dmin': 1e-5, 'Dmax': 1e13, 'dpdxThreshold': 10})
    f = tango.multifield.Field(label='n', rightBC=ens['nL'], profile_mminus1=ens['u0'], compute_all_H=sh.ComputeAllH(ens['S0'], ens['delta']),
                               lodestroMethod=lm)
    tango.multifield.check_fields_initialize([f])
    model = Model(ens['dx'])
    call = tango.multifield.ComputeAllHAllFields([f], tango.lodestro_method.TurbulenceHandler(ens['dx'], ens['x'], model))
    s = tango.solver.Solver(1, ens['x'], np.array([0, 1e4]), its, 1e-30, call, [f], iterationsPerCheck=8, saveProfileHistory=False)
    assert type(s._fused).__name__ == '_FusedExternal1F'
    t_model = 0.0
    inner = model.get_flux

    def timed_get_flux(profiles, inner=inner):
        global t_model
        if not getattr(model, 'device_flux', False):
            t0 = time.perf_counter()
            out = inner(profiles)
            t_model += time.perf_counter() - t0
            return out
        return inner(profiles)
    model.get_flux = timed_get_flux
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s.take_timestep()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    print(json.dumps({'path': name, 'profiles': nb, 'N': n, 'iterations': its, 'ms_per_iteration': round(wall / its * 1e3, 2),
                      'of_which_flux_model_on_host_ms': round(t_model / its * 1e3, 2)}), flush=True)
