import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'scripts'))
import numpy as np, torch
from tango_amd import _hip as H
from tango_amd.ensemble import OneFieldEnsemble, shestakov_ensemble
import bench_configs as B
H.set_stream(torch.cuda.current_stream().cuda_stream)
st = torch.cuda.current_stream().cuda_stream
nb, n = 65536, 512
ens = shestakov_ensemble(nb, n)
x = ens['x']
for mode in ('builtin', 'geometry', 'external', 'external+arrays', 'external+arrays+smooth', 'external+geometry', 'tread'):
    E = OneFieldEnsemble(ens, a_turb=0.3, a_prof=1.0, fixed_count=True, keep_err=False)
    q = E.params
    keep = []
    if 'geometry' in mode:
        Vp = H.DeviceArray.from_host(1.0 + x); gxx = H.DeviceArray.from_host(1.0 + 0.2 * x ** 2); gr = H.DeviceArray.from_host(np.sqrt(1.0 + 0.2 * x ** 2))
        keep += [Vp, gxx, gr]
        q.H1, q.Vprime, q.gxx, q.gradx, q.gxx_turb, q.gradx_turb = Vp.ptr, Vp.ptr, gxx.ptr, gr.ptr, gxx.ptr, gr.ptr
    if 'external' in mode:
        fl = H.DeviceArray((nb, n), zero=True); keep.append(fl)
        fl.set(np.full((nb, n), 0.3))
        E.uE.set(ens['u0'])
        q.flux_model, q.flux = H.FLUX_EXTERNAL, fl.ptr
    if 'arrays' in mode:        # H1 and H7 as arrays, as Solver's external-flux path passes them
        h1 = H.DeviceArray.from_host(np.ones(n)); h7 = H.DeviceArray.from_host(np.where(x < 0.1, 1.0, 0.0))
        keep += [h1, h7]
        q.H1, q.H7, q.h7_stride = h1.ptr, h7.ptr, 0
    if 'smooth' in mode:        # a five-point flux smoother inside the kernel, as the reference's GENE drivers use
        q.smooth_window = 5
    if mode == 'tread':
        q.pseudo_time, q.tread_lightly, q.tread_inner_count, q.tread_max_change = 1, 1, n // 2, 0.2
    for _ in range(150):         # long enough for the GPU's clock to come up (50 ms of launches)
        E.step(st)
    ms = B.timed(lambda: E.step(st), 48)
    B.report(mode, nb, n, 72 + (8 if 'external' in mode else 0), ms)
