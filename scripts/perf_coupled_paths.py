"""Per-iteration wall time of the three paths a three-field (n, pi <-> pe) run can take, on one ensemble: the built-in
closure (one launch per iteration, nothing crosses PCIe), the fused step with outside inputs (a Python flux model and
Python compute_all_H callbacks on the host, fluxes and H6/H7/H8 through page-locked buffers, one launch per iteration),
and the generic path (one kernel per reference function, host arrays in between).  Prints one JSON line per path.

    python scripts/perf_coupled_paths.py [profiles] [N] [iterations]"""
import json, os, sys, time
import numpy as np
import torch          # before tango_amd: the device path below shares torch's HIP runtime with the library
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import tango_amd as tango
from tango_amd import _hip as H
from tango_amd.ensemble import threefield_ensemble
from tango_amd.extras import threefield as T

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
its = int(sys.argv[3]) if len(sys.argv) > 3 else 8
ens = threefield_ensemble(nb, n)
th = {'Dmin': 1e-5, 'Dmax': 1e13, 'dpdxThreshold': 10}


class UserFlux(object):
    def __init__(self, dx):
        self.inner, self.seconds = T.ShestakovThreeFieldFluxModel(dx), 0.0

    def get_flux(self, profiles):
        t0 = time.perf_counter()
        out = self.inner.get_flux(profiles)
        self.seconds += time.perf_counter() - t0
        return out


import device_coupled_case as DC          # a torch flux model and torch collision terms (tests/device_coupled_case.py)

for name, external, kw in (('built-in closure', False, {}), ('outside inputs, fused step', True, {}), ('generic path', True, {'fusedPaths': False}),
                           ('outside inputs on the GPU (torch flux model and callbacks), fused step', 'device', {})):
    if external == 'device':
        s, _ = DC.build(n, ens['nu0'], tuple(ens['bc'][:, k].copy() for k in range(3)), ic=ens['u0'], iterationsPerCheck=8, saveProfileHistory=False)
        s.maxIterations, s.tol = its, 1e-30
        s.errHistory = np.zeros(its)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s.take_timestep()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        print(json.dumps({'path': name, 'runner': type(s._fused).__name__, 'profiles': nb, 'N': n, 'iterations': its,
                          'ms_per_iteration': round(wall / its * 1e3, 2)}), flush=True)
        continue
    cbs = [T.ComputeAllH_n(), T.ComputeAllH_pi(ens['nu0']), T.ComputeAllH_pe(ens['nu0'])]
    spent = [0.0]
    if external:
        def wrap(cb):
            def call(t, x, p, HT):
                t0 = time.perf_counter()
                out = cb(t, x, p, tango.multifield.HCoefficients())
                spent[0] += time.perf_counter() - t0
                return out + HT
            return call
        cbs = [wrap(cb) for cb in cbs]
    lms = [tango.lodestro_method.lm(0.3, 0.3, th) for _ in range(3)]
    F = tango.multifield.Field
    bc = [ens['bc'][:, k].copy() for k in range(3)]
    fields = [F(label='n', rightBC=bc[0], profile_mminus1=ens['u0'].copy(), compute_all_H=cbs[0], lodestroMethod=lms[0]),
              F(label='pi', rightBC=bc[1], profile_mminus1=ens['u0'].copy(), compute_all_H=cbs[1], lodestroMethod=lms[1], coupledTo='pe'),
              F(label='pe', rightBC=bc[2], profile_mminus1=ens['u0'].copy(), compute_all_H=cbs[2], lodestroMethod=lms[2], coupledTo='pi')]
    tango.multifield.check_fields_initialize(fields)
    model = UserFlux(ens['dx']) if external else T.ShestakovThreeFieldFluxModel(ens['dx'])
    call = tango.multifield.ComputeAllHAllFields(fields, tango.lodestro_method.TurbulenceHandler(ens['dx'], ens['x'], model))
    s = tango.solver.Solver(1, ens['x'], np.array([0, 1e6]), its, 1e-30, call, fields, iterationsPerCheck=8, saveProfileHistory=False, **kw)
    H.synchronize()
    t0 = time.perf_counter()
    s.take_timestep()
    H.synchronize()
    wall = time.perf_counter() - t0
    print(json.dumps({'path': name, 'runner': type(s._fused).__name__ if s._fused is not None else None, 'profiles': nb, 'N': n, 'iterations': its,
                      'ms_per_iteration': round(wall / its * 1e3, 2),
                      'of_which_python_flux_model_ms': round(getattr(model, 'seconds', 0.0) / its * 1e3, 2),
                      'of_which_python_callbacks_ms': round(spent[0] / its * 1e3, 2)}), flush=True)
