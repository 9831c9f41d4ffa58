"""Throughput of the standalone kernels (tridiagonal solve, EWMA, residual, flux split) at the benchmark sizes."""
import sys, os, ctypes as C
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'scripts'))
import numpy as np, torch
from tango_amd import _hip as H
import bench_configs as B
H.set_stream(torch.cuda.current_stream().cuda_stream)
st = H.stream()
for nb, n in ((65536, 512), (65536, 256), (16384, 2048)):
    rng = np.random.default_rng(0)
    a = -rng.uniform(0.5, 1, (nb, n)); c = -rng.uniform(0.5, 1, (nb, n)); b = 2.5 + rng.uniform(0, 1, (nb, n)); d = rng.standard_normal((nb, n))
    dA, dB, dC, dD = (H.DeviceArray.from_host(v) for v in (a, b, c, d))
    u = H.DeviceArray((nb, n)); status = H.DeviceArray((nb,), np.int32, zero=True)
    f = lambda: H.check(H.lib.tg_tridiag_solve(dA.ptr, dB.ptr, dC.ptr, dD.ptr, u.ptr, nb, n, 0, status.ptr, None, 0, st))
    for _ in range(3): f()
    ms = B.timed(f, 20)
    print(f'tridiag_solve pf {nb}x{n}: {ms:.4f} ms  {nb*n*40/ms/1e6:.0f} GB/s ({nb*n*40/ms/1e6/80:.1f}% of 8 TB/s)', flush=True)
    # ewma
    y = dD; ye = H.DeviceArray.from_host(b)
    f2 = lambda: H.check(H.lib.tg_ewma(y.ptr, ye.ptr, 0.3, 1, nb * n, st))
    for _ in range(3): f2()
    ms = B.timed(f2, 20)
    print(f'ewma {nb}x{n}: {ms:.4f} ms  {nb*n*24/ms/1e6:.0f} GB/s', flush=True)
    # residual
    res = H.DeviceArray((nb, n)); ss = H.DeviceArray((nb,))
    f3 = lambda: H.check(H.lib.tg_residual(dA.ptr, dB.ptr, dC.ptr, dD.ptr, u.ptr, None, None, res.ptr, ss.ptr, nb, n, st))
    for _ in range(3): f3()
    ms = B.timed(f3, 20)
    print(f'residual {nb}x{n}: {ms:.4f} ms  {nb*n*48/ms/1e6:.0f} GB/s', flush=True)
    Dd, cc = H.DeviceArray((nb, n)), H.DeviceArray((nb, n))
    f4 = lambda: H.check(H.lib.tg_flux_split(dD.ptr, dB.ptr, 1.0 / (n - 1), 1e-5, 1e13, 10.0, None, None, Dd.ptr, cc.ptr, None, None, None, nb, n, st))
    for _ in range(3): f4()
    ms = B.timed(f4, 20)
    print(f'flux_split {nb}x{n}: {ms:.4f} ms  {nb*n*32/ms/1e6:.0f} GB/s', flush=True)
    del dA, dB, dC, dD, u, res, Dd, cc, ye
