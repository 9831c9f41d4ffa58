"""Time the fused one-field step with HIP events on the launch stream (torch supplies stream + events)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get('TG_LIB'):
    import tango_amd._hip as _h0  # noqa
from tango_amd import _hip as H
from tango_amd import ensemble as orc
C = H.C

def main(nb=65536, n=512, iters=30, warm=5, a_prof=1.0, kper=1):
    dev = torch.device('cuda:0')
    H.set_stream(torch.cuda.current_stream().cuda_stream)
    ens = orc.shestakov_ensemble(nb, n)
    t = lambda a, dt=torch.float64: torch.as_tensor(np.ascontiguousarray(a), dtype=dt).to(dev)
    skew = int(os.environ.get('TG_SKEW', '0'))       # bytes of extra offset between consecutive state arrays
    pool = torch.zeros(9 * (nb * n + skew // 8 + 64), dtype=torch.float64, device=dev)
    def carve(k):
        o = k * (nb * n + skew // 8)
        return pool[o:o + nb * n].view(nb, n)
    u, um1, uE, DE, cE = (carve(k) for k in range(5))
    u.copy_(t(ens['u0'])); um1.copy_(u)
    outs = [carve(k) for k in range(5, 9)]
    ins5 = (H._p * 5)(*[x.data_ptr() for x in (u, um1, uE, DE, cE)]); outs4 = (H._p * 4)(*[x.data_ptr() for x in outs])
    st0 = torch.cuda.current_stream().cuda_stream
    for _ in range(3): H.check(H.lib.tg_stream_probe(ins5, outs4, nb * n, st0))
    torch.cuda.synchronize(); p0, p1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    p0.record()
    for _ in range(10): H.check(H.lib.tg_stream_probe(ins5, outs4, nb * n, st0))
    p1.record(); torch.cuda.synchronize()
    pms = p0.elapsed_time(p1) / 10
    print(f'stream probe 5r+4w (skew {skew} B): {pms:.4f} ms  {nb * n * 72 / pms / 1e9:.3f} TB/s', flush=True)
    status = torch.zeros(nb, dtype=torch.int32, device=dev); its = torch.zeros(nb, dtype=torch.int32, device=dev)
    UL, S0, dl = t(ens['nL']), t(ens['S0']), t(np.searchsorted(ens['x'], ens['delta']).astype(np.int32), torch.int32)
    S = H.State1F(nb, n, u.data_ptr(), um1.data_ptr(), uE.data_ptr(), DE.data_ptr(), cE.data_ptr(), status.data_ptr(), its.data_ptr(), None, 0)
    Q = H.Params1F()
    Q.dt, Q.dx, Q.a_turb, Q.a_prof = 1e4, ens['dx'], 0.3, a_prof
    Q.Dmin, Q.Dmax, Q.dpdx_thr, Q.tol = 1e-5, 1e13, 10.0, 1e-11
    Q.UL, Q.S0, Q.src_cut = UL.data_ptr(), S0.data_ptr(), dl.data_ptr()
    Q.flux_model, Q.smooth_window, Q.fixed_count, Q.frozen_turb = 0, 1, 1, 0
    Q.iters_per_launch = kper
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(warm):
        H.check(H.lib.tg_picard_1f_step(C.byref(S), C.byref(Q), st))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        H.check(H.lib.tg_picard_1f_step(C.byref(S), C.byref(Q), st))
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters / kper
    print(f'nb={nb} n={n} k={kper}: {ms:.4f} ms/iter  {1e3 / ms:.1f} it/s  {nb * n * 72 / ms / 1e9:.3f} TB/s algorithmic ({nb * n * 72 / ms / 1e9 / 8 * 100:.1f}% of 8 TB/s); invalid={(status & 2).ne(0).sum().item()}', flush=True)
    # copy bandwidth probe for reference
    a = torch.empty(nb * n * 4, dtype=torch.float64, device=dev); b2 = torch.empty_like(a)
    for _ in range(3): b2.copy_(a)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): b2.copy_(a)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f'copy probe: {2 * a.numel() * 8 / ms / 1e9:.3f} TB/s', flush=True)

if __name__ == '__main__':
    if len(sys.argv) > 2:
        main(int(sys.argv[1]), int(sys.argv[2]), kper=int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    else:
        main()
