"""Throughput of the standalone coupled solves at the three-field benchmark size (16 384 x 512), against their
algorithmic bytes: the pair-coupled solve (tango/JK_to_matrix_fd.py:132-166: ten coefficient arrays in, two solutions
out = 96 B/point; with the scalar solve of the uncoupled field, 40 B/point, the reference's three-field structure moves
136 B/point) and the dense 3 x 3 block-tridiagonal solve (27 block entries + 3 right-hand sides in, 3 solutions out =
264 B/point).  Prints one JSON line per kernel.  Meant to be run under scripts/profile_cmd.sh too."""
import json, os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'scripts'))
import numpy as np, torch
from tango_amd import _hip as H
import bench_configs as B
H.set_stream(torch.cuda.current_stream().cuda_stream)
st = H.stream()
nb, n = 16384, 512
rng = np.random.default_rng(0)
D = H.DeviceArray


def line(kernel, ms, bpp, note=''):
    print(json.dumps({'kernel': kernel, 'profiles': nb, 'N': n, 'ms': round(ms, 4), 'bytes_per_point': bpp,
                      'algorithmic_GBps': round(nb * n * bpp / ms / 1e6, 1), 'frac_of_8TBps': round(nb * n * bpp / ms / 1e6 / 8000, 4), 'note': note}), flush=True)


off = lambda: -rng.uniform(0.5, 1, (nb, n))
diag = lambda: 3.0 + rng.uniform(0, 1, (nb, n))
# scalar solve of the uncoupled field (point-fastest)
A, Bm, Cm, Dm = (D.from_host(v) for v in (off(), diag(), off(), rng.standard_normal((nb, n))))
u = D((nb, n)); status = D((nb,), np.int32, zero=True)
f = lambda: H.check(H.lib.tg_tridiag_solve(A.ptr, Bm.ptr, Cm.ptr, Dm.ptr, u.ptr, nb, n, 0, status.ptr, None, 0, st))
for _ in range(3):
    f()
t_tri = B.timed(f, 20)
line('tg_tridiag_solve (point-fastest)', t_tri, 40)
# pair-coupled solve
arrs = [D.from_host(v) for v in (off(), diag(), off(), rng.standard_normal((nb, n)), -rng.uniform(0, 0.5, (nb, n)),
                                  off(), diag(), off(), rng.standard_normal((nb, n)), -rng.uniform(0, 0.5, (nb, n)))]
U, W = D((nb, n)), D((nb, n))
g = lambda: H.check(H.lib.tg_pair_solve(*[a.ptr for a in arrs], U.ptr, W.ptr, nb, n, status.ptr, st))
for _ in range(3):
    g()
t_pair = B.timed(g, 20)
line('tg_pair_solve', t_pair, 96)
line('tg_tridiag_solve + tg_pair_solve', t_tri + t_pair, 136, "the reference's three-field structure: one field on its own, two as a pair")
del arrs, U, W, A, Bm, Cm, Dm, u
# dense 3 x 3 blocks, component-major
plane = nb * n
Cb = -rng.uniform(0.1, 0.3, (9, nb, n)); Ab = -rng.uniform(0.1, 0.3, (9, nb, n))
Bb = rng.uniform(-0.2, 0.2, (9, nb, n))
for r in range(3):
    Bb[4 * r] = 4.0 + rng.uniform(0, 1, (nb, n))
dCb, dBb, dAb = D.from_host(Cb), D.from_host(Bb), D.from_host(Ab)
d3, x3 = D.from_host(rng.standard_normal((3, nb, n))), D((3, nb, n))
h = lambda: H.check(H.lib.tg_block3_solve(dCb.ptr, dBb.ptr, dAb.ptr, d3.ptr, x3.ptr, nb, n, status.ptr, st))
for _ in range(3):
    h()
line('tg_block3_solve (dense 3x3 blocks)', B.timed(h, 20), 264)
