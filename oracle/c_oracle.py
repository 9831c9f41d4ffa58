"""ctypes access to the C restatement (oracle/tango_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, '_build', 'liboracle.so')


class Params(C.Structure):
    _fields_ = [(k, C.c_double) for k in ('dt', 'dx', 'a_turb', 'a_prof', 'Dmin', 'Dmax', 'thr', 'tol')]


def build():
    subprocess.check_call(['make', '-s', '-C', HERE, '_build/liboracle.so'])
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        _lib = C.CDLL(LIB)
        _lib.orc_picard_1f_step.restype = C.c_int
    return _lib


class Ensemble1F:
    """Host-resident one-field ensemble stepped by the C oracle."""

    def __init__(self, ens, dt=1e4, a_turb=0.3, a_prof=1.0, tol=1e-11, max_iter=1000, fixed_count=False,
                 Dmin=1e-5, Dmax=1e13, thr=10.0):
        self.u = np.ascontiguousarray(ens['u0'], dtype=np.float64).copy()
        self.nb, self.n = self.u.shape
        self.um1 = self.u.copy()
        self.uE, self.DE, self.cE = (np.zeros_like(self.u) for _ in range(3))
        self.status = np.zeros(self.nb, np.int32)
        self.iters = np.zeros(self.nb, np.int32)
        self.err = np.zeros((self.nb, max_iter))
        self.UL, self.S0, self.delta = (np.ascontiguousarray(ens[k], dtype=np.float64) for k in ('nL', 'S0', 'delta'))
        self.q = Params(dt, ens['dx'], a_turb, a_prof, Dmin, Dmax, thr, tol)
        self.fixed = int(fixed_count)

    def step(self, nthreads=0):
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        return lib().orc_picard_1f_step(C.c_int64(self.nb), C.c_int(self.n), p(self.u), p(self.um1), p(self.uE), p(self.DE),
                                        p(self.cE), p(self.status), p(self.iters), p(self.err), C.c_int64(self.err.shape[1]),
                                        p(self.UL), p(self.S0), p(self.delta), C.byref(self.q), C.c_int(self.fixed),
                                        C.c_int(nthreads))
