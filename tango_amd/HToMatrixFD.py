"""Finite-difference assembly of the Picard update of

    H1 dU/dt - d/dx( H2 dU/dx + H3 U + H4 ) - H6 U - H7 = 0

into a tridiagonal system, no-flux left boundary, Dirichlet right boundary U(L) = UL, conservative
half-grid fluxes and adaptive upwinding of the H3 term.  Same functions and argument meaning as
tango/HToMatrixFD.py:53-124; arrays may be (N,) or (profiles, N) (coefficients of shape (N,) are
shared by all profiles; dt and UL may be per-profile), the work is done by the tg_h_to_matrix kernel."""
import numpy as np

from . import _ops, tridiag


def H_to_matrix(dt, dx, UL, U_mminus1, H1, H2=None, H3=None, H4=None, H6=None, H7=None):
    N = np.shape(U_mminus1)[-1]
    for name, Hk in (('H1', H1), ('H2', H2), ('H3', H3), ('H4', H4), ('H6', H6), ('H7', H7)):
        if Hk is not None:
            assert np.shape(Hk)[-1] == N, f'{name} must have the length of U_mminus1'
    return _ops.h_to_matrix(dt, dx, UL, U_mminus1, H1, H2, H3, H4, H6, H7)


def solve(A, B, C, D):
    return tridiag.solve(A, B, C, D)


def H_to_matrix_and_solve(dt, dx, UL, U_mminus1, H1, H2=None, H3=None, H4=None, H6=None, H7=None):
    (A, B, C, D) = H_to_matrix(dt, dx, UL, U_mminus1, H1, H2, H3, H4, H6, H7)
    return solve(A, B, C, D)
