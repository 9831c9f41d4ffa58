// Point-wise pieces of the LoDestro split and the finite-difference assembly, written once and
// used by both the standalone kernels and the fused Picard kernels.  Each function states the
// reference expression it evaluates; operations are kept in the reference's order.
#pragma once
#include <hip/hip_runtime.h>

namespace tg {

struct Theta {
    double Dmin, Dmax, thr;
};

// tango/lodestro_method.py:624-634 (DHat, cHat, D, c) with the default theta rule of :667-676.
// gxx / gradx are the geometric factors on the turbulence grid (1 in Cartesian geometry).
__device__ __forceinline__ void flux_split_point(double flux, double p, double dpdx, double gxx, double gradx,
                                                 const Theta &th, double &D, double &c, double &DHat,
                                                 double &cHat, double &theta) {
    DHat = -flux / (gxx * dpdx);
    if (dpdx == 0.0) DHat = 0.0;
    cHat = flux / (gradx * p);
    theta = 1.0;
    if (DHat < th.Dmin) theta = 0.0;
    const bool small = fabs(dpdx) < th.thr;
    if (small && DHat >= th.Dmin && DHat <= th.Dmax) theta = (1.0 + (th.Dmax - DHat) / (th.Dmax - th.Dmin)) / 2.0;
    if (small && DHat > th.Dmax) theta = 0.5;
    D = theta * DHat;
    c = (1.0 - theta) * cHat;
}

// tango/HToMatrixFD.py:277-297: adaptive upwinding weight at a half-integer grid point.
__device__ __forceinline__ double fbeta_point(double H2h, double H3h, double dx) {
    const double gamma = 0.9;
    const double c = -H3h;
    double beta = 0.0;
    if (c > 0.0) {
        double star = 1.0 - H2h / (c * dx);
        if (star <= 0.5) beta = 0.5;
        else if (star > 0.5) beta = gamma * star + (1.0 - gamma);
    } else if (c < 0.0) {
        double star = -H2h / (c * dx);
        if (star >= 0.5) beta = 0.5;
        else if (star < 0.5) beta = gamma * star;
    }
    return beta;
}

// tango/derivatives.py:20-24: centred difference with two-point one-sided ends.
// um, u0, up are the values at j-1, j, j+1 (um unused at j = 0, up unused at j = n-1).
__device__ __forceinline__ double ddx_edge1_point(double um, double u0, double up, int j, int n, double dx) {
    if (j == 0) return (up - u0) / dx;
    if (j == n - 1) return (u0 - um) / dx;
    return (up - um) / (2.0 * dx);
}

// One row of tango/HToMatrixFD.py:87-117 accumulated in the order H1, H2, H3, H4, H6, H7, boundary.
// h2m/h3m/bm: H2, H3, beta at j-1/2; h2p/h3p/bp at j+1/2.  g4 is the H4 term already formed
// (-(H4_{j+1}-H4_{j-1})/(2dx), or -(H4_0+H4_1)/(2dx) at j = 0; 0 when H4 is absent); h6, h7 are
// 0 when absent.  Returns A (super), B (diag), C (sub) and the right-hand side D = -g.
__device__ __forceinline__ void fd_row(int j, int n, double H1, double Um1, double dt, double dx, double UL,
                                       double h2m, double h3m, double bm, double h2p, double h3p, double bp,
                                       double g4, double h6, double h7, double &A, double &B, double &C,
                                       double &D) {
    if (j == n - 1) {  // Dirichlet row (:200-204)
        A = 0.0;
        C = 0.0;
        B = 1.0;
        D = UL;
        return;
    }
    const double dx2 = dx * dx;
    double g = -H1 * Um1 / dt;
    A = -h2p / dx2 + (-h3p * (1.0 - bp) / dx);
    if (j == 0) {
        B = H1 / dt + h2p / dx2 + (-h3p * bp / dx);
        C = 0.0;
    } else {
        B = H1 / dt + (h2m + h2p) / dx2 + (-h3p * bp + h3m * (1.0 - bm)) / dx;
        C = -h2m / dx2 + h3m * bm / dx;
    }
    B += -h6;
    g += g4;
    g += -h7;
    D = -g;
}

}  // namespace tg
