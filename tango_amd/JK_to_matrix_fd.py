"""Two pair-coupled fields U, W:

    J1 dU/dt - d/dx( J2 dU/dx + J3 U + J4 ) - J6 U - J7 - J8 W = 0
    K1 dW/dt - d/dx( K2 dW/dx + K3 W + K4 ) - K6 W - K7 - K8 U = 0

Same interface as tango/JK_to_matrix_fd.py:49-166.  Assembly is two tg_h_to_matrix launches (the coupling
column G = -H8 comes out of the same kernel); the solve is the 2x2 block-tridiagonal HIP kernel instead of
the reference's interleaved five-band LAPACK call."""
from collections import namedtuple

import numpy as np

from . import _ops

MatEqnCoeffs = namedtuple('MatEqnCoeffs', ('A', 'B', 'C', 'D', 'G'))


def JK_to_matrix(dt, dx, rightBC, psi_mminus1,
                 J1=None, J2=None, J3=None, J4=None, J6=None, J7=None, J8=None,
                 K1=None, K2=None, K3=None, K4=None, K6=None, K7=None, K8=None):
    U_mminus1, W_mminus1 = psi_mminus1[0], psi_mminus1[1]
    UL, WL = rightBC[0], rightBC[1]
    assert np.shape(W_mminus1)[-1] == np.shape(U_mminus1)[-1]
    assert J1 is not None, 'Must supply J1'
    assert K1 is not None, 'Must supply K1'
    AU, BU, CU, DU, GW = _ops.h_to_matrix(dt, dx, UL, U_mminus1, J1, J2, J3, J4, J6, J7, J8, want_G=True)
    AW, BW, CW, DW, GU = _ops.h_to_matrix(dt, dx, WL, W_mminus1, K1, K2, K3, K4, K6, K7, K8, want_G=True)
    return (MatEqnCoeffs(A=AU, B=BU, C=CU, D=DU, G=GW), MatEqnCoeffs(A=AW, B=BW, C=CW, D=DW, G=GU))


def solve(UEqnCoeffs, WEqnCoeffs):
    U = (UEqnCoeffs.A, UEqnCoeffs.B, UEqnCoeffs.C, UEqnCoeffs.D, UEqnCoeffs.G)
    W = (WEqnCoeffs.A, WEqnCoeffs.B, WEqnCoeffs.C, WEqnCoeffs.D, WEqnCoeffs.G)
    return _ops.pair_solve(U, W)
