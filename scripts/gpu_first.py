"""First GPU check: tridiagonal solves and the fused one-field step against the CPU oracle."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tango_amd import _hip as H
from oracle import tango_oracle as orc
C = H.C

def relerr(a, b):
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))

print('devices', H.device_count(), flush=True)
rng = np.random.default_rng(0)
for n in (12, 64, 100, 256, 500, 512, 513, 1000, 2048, 8192):
    nb = 7
    A = rng.uniform(-1, 1, (nb, n)); Cc = rng.uniform(-1, 1, (nb, n))
    B = np.abs(A) + np.abs(Cc) + rng.uniform(0.01, 1, (nb, n)); D = rng.normal(size=(nb, n))
    ref = orc.tridiag_solve(A, B, Cc, D)
    dA, dB, dC, dD = (H.DeviceArray.from_host(v) for v in (A, B, Cc, D))
    du = H.DeviceArray((nb, n)); st = H.DeviceArray((nb,), np.int32, zero=True)
    H.check(H.lib.tg_tridiag_solve(dA.ptr, dB.ptr, dC.ptr, dD.ptr, du.ptr, nb, n, 0, st.ptr, None, 0, None))
    e_pf = relerr(du.to_host(), ref)
    # batch-fastest
    tA, tB, tC, tD = (H.DeviceArray.from_host(np.ascontiguousarray(v.T)) for v in (A, B, Cc, D))
    tu = H.DeviceArray((n, nb)); wb = H.lib.tg_tridiag_work_bytes(nb, n, 1); w = H.DeviceArray((wb // 8,))
    H.check(H.lib.tg_tridiag_solve(tA.ptr, tB.ptr, tC.ptr, tD.ptr, tu.ptr, nb, n, 1, st.ptr, w.ptr, wb, None))
    e_bf = relerr(tu.to_host().T, ref)
    print(f'tridiag n={n}: point-fastest err {e_pf:.2e}  batch-fastest err {e_bf:.2e} status {st.to_host().max()}', flush=True)

def run_fused(nb, n, iters, a_prof, fixed=True, tol=1e-11):
    ens = orc.shestakov_ensemble(nb, n)
    u = H.DeviceArray.from_host(ens['u0']); um1 = H.DeviceArray.from_host(ens['u0'])
    uE = H.DeviceArray((nb, n), zero=True); DE = H.DeviceArray((nb, n), zero=True); cE = H.DeviceArray((nb, n), zero=True)
    status = H.DeviceArray((nb,), np.int32, zero=True); its = H.DeviceArray((nb,), np.int32, zero=True)
    err = H.DeviceArray((nb, iters), zero=True)
    UL = H.DeviceArray.from_host(ens['nL']); S0 = H.DeviceArray.from_host(ens['S0']); dl = H.DeviceArray.from_host(np.searchsorted(ens['x'], ens['delta']).astype(np.int32))
    S = H.State1F(nb, n, u.ptr, um1.ptr, uE.ptr, DE.ptr, cE.ptr, status.ptr, its.ptr, err.ptr, iters)
    Q = H.Params1F()
    Q.dt, Q.dx, Q.a_turb, Q.a_prof = 1e4, ens['dx'], 0.3, a_prof
    Q.Dmin, Q.Dmax, Q.dpdx_thr, Q.tol = 1e-5, 1e13, 10.0, tol
    Q.UL, Q.S0, Q.src_cut = UL.ptr, S0.ptr, dl.ptr
    Q.flux_model, Q.smooth_window, Q.fixed_count, Q.frozen_turb = 0, 1, int(fixed), 0
    keep = (u, um1, uE, DE, cE, status, its, err, UL, S0, dl)
    t0 = time.time()
    for _ in range(iters):
        H.check(H.lib.tg_picard_1f_step(C.byref(S), C.byref(Q), None))
    H.synchronize()
    return ens, u.to_host(), uE.to_host(), DE.to_host(), cE.to_host(), err.to_host(), status.to_host(), its.to_host(), time.time() - t0, keep

for n, a_prof in ((512, 1.0), (500, 0.3), (256, 1.0), (40, 1.0), (1000, 0.3), (2048, 0.3)):
    nb, iters = 5, 12
    ens, u, uE, DE, cE, err, status, its, _, _ = run_fused(nb, n, iters, a_prof)
    H7 = orc.shestakov_source(ens['x'], ens['S0'], ens['delta'])
    worst = 0; errs = []
    for b in range(nb):
        st = orc.State1F(ens['u0'][b])
        for it in range(iters):
            rms, bad = orc.picard_step_1f(st, 1e4, ens['dx'], ens['nL'][b], np.ones(n), H7[b], 0.3, a_prof)
            errs.append(abs(err[b, it] - rms) / rms)
        worst = max(worst, relerr(u[b], st.u), relerr(DE[b], st.DE), relerr(cE[b], st.cE), relerr(uE[b], st.uE))
    print(f'fused n={n} a_prof={a_prof}: worst rel err after {iters} its {worst:.2e}; rms hist err {max(errs):.2e}; status {status}, iters {its}', flush=True)

# quick timing at the metric size
nb, n, iters = 65536, 512, 20
ens, u, uE, DE, cE, err, status, its, dtm, keep = run_fused(nb, n, iters, 1.0)
print(f'timing nb={nb} n={n}: {dtm / iters * 1e3:.3f} ms/iter (incl. launch), {nb * n * 72 / (dtm / iters) / 1e12:.3f} TB/s algorithmic; invalid={int((status & 2).astype(bool).sum())}', flush=True)
