import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]
    return get


def relerr(a, b):
    """max |a-b| / max |b| : the relative error norm used for the 1e-10 parity bar."""
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    den = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / (den if den > 0 else 1.0))


def relerr_elem(a, b):
    """max over elements of |a-b| / |b| (entries with |b| < 1e-300 left out): no slack for the small tail of a profile."""
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    ok = np.abs(b) > 1e-300
    return float(np.max(np.abs(a - b)[ok] / np.abs(b)[ok])) if ok.any() else 0.0


def relerr_tail(a, b):
    """relerr restricted to the last quarter of the grid (last axis), where a Shestakov profile is ~100 times smaller."""
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    q = b.shape[-1] - b.shape[-1] // 4
    return relerr(a[..., q:], b[..., q:])
