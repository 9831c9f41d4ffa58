"""Throughput of the standalone tridiagonal solve in both layouts at 65 536 x 512 (40 algorithmic bytes per point:
four diagonals in, solution out).  Prints one JSON line per layout.  Meant to be run under scripts/profile_cmd.sh too."""
import json, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'scripts'))
import numpy as np, torch
from tango_amd import _hip as H
import bench_configs as B
H.set_stream(torch.cuda.current_stream().cuda_stream)
st = H.stream()
nb, n = int(os.environ.get('TG_NB', 65536)), 512
rng = np.random.default_rng(0)
a = -rng.uniform(0.5, 1, (nb, n)); c = -rng.uniform(0.5, 1, (nb, n)); b = 2.5 + rng.uniform(0, 1, (nb, n)); d = rng.standard_normal((nb, n))
for layout, name in ((0, 'point-fastest (wave per profile, chunked elimination + PCR)'), (1, 'batch-fastest (lane per profile, dgtsv sweeps)'),
                     (2, 'batch-fastest, tiles of 16 profiles through LDS, chunked elimination + PCR')):
    arrs = [H.DeviceArray.from_host(v if layout == 0 else np.ascontiguousarray(v.T)) for v in (a, b, c, d)]
    u = H.DeviceArray((nb, n)); status = H.DeviceArray((nb,), np.int32, zero=True)
    wb = H.lib.tg_tridiag_work_bytes(nb, n, layout)
    work = H.DeviceArray((max(wb // 8, 1),))
    f = lambda: H.check(H.lib.tg_tridiag_solve(arrs[0].ptr, arrs[1].ptr, arrs[2].ptr, arrs[3].ptr, u.ptr, nb, n, layout, status.ptr, work.ptr, wb, st))
    for _ in range(3):
        f()
    ms = B.timed(f, 20)
    print(json.dumps({'kernel': 'tg_tridiag_solve', 'layout': name, 'profiles': nb, 'N': n, 'ms': round(ms, 4),
                      'algorithmic_GBps': round(nb * n * 40 / ms / 1e6, 1), 'frac_of_8TBps': round(nb * n * 40 / ms / 1e6 / 8000, 4)}), flush=True)
    del arrs, u, work
