"""The C restatement (oracle/tango_oracle.c, the CPU baseline of the benchmark) against the golden vectors."""
import numpy as np

from conftest import relerr
from oracle import c_oracle, tango_oracle as orc


def test_c_oracle_trajectory(golden):
    g = golden('g8_onefield')
    for N, tag, ap in ((500, 'p1', 1.0), (512, 'p03', 0.3), (256, 'p1', 1.0)):
        key = f'N{N}_{tag}'
        x = np.arange(N) * (1 / (N - 1))
        ens = dict(x=x, dx=1 / (N - 1), u0=(1 - 0.5 * x)[None], nL=np.array([1e-2]), S0=np.array([1.0]), delta=np.array([0.1]))
        E = c_oracle.Ensemble1F(ens, a_prof=ap, max_iter=200)
        iters = int(g[f'{key}_iters'])
        for it in range(iters + 2):
            E.step(1)
            if it in (0, 1, 2, 10):
                assert relerr(E.u[0], g[f'{key}_it{it}_u']) < 1e-12
                assert relerr(E.DE[0], g[f'{key}_it{it}_DE']) < 1e-12 and relerr(E.cE[0], g[f'{key}_it{it}_cE']) < 1e-11
        assert abs(int(E.iters[0]) - iters) <= 1 and (E.status[0] & 1)
        assert relerr(E.u[0], g[f'{key}_final']) < 1e-10


def test_c_oracle_ensemble_threads(golden):
    g = golden('g10_ensemble32')
    ens = orc.shestakov_ensemble(32, 512)
    E = c_oracle.Ensemble1F(ens, max_iter=100)
    for _ in range(100):
        E.step(4)
    assert np.all(E.status & 1)
    assert np.max(np.abs(E.iters - g['iters'])) <= 1
    assert relerr(E.u, g['finals']) < 1e-10
