// Point-wise pieces of the LoDestro split and the finite-difference assembly, written once and
// used by both the standalone kernels and the fused Picard kernels.  Each function states the
// reference expression it evaluates; operations are kept in the reference's order.
//
// Divisions by quantities that are constant over a launch (dx, dx^2, 2dx, dt) can be done as
// multiplications by a reciprocal formed once (an fp64 divide costs ~15 fp64 issue slots on gfx950 and
// the assembly has nine of them per grid point).  That changes a coefficient by at most one ulp
// relative to the reference's true division.  The template flag EXACT selects correctly rounded quotients
// instead (quot of tg_comm.h): the standalone kernels and the large-grid (n > 512, condition number > 1e9) fused
// path use them, the wave-per-profile fused path uses reciprocals.  All branches are written as selects so that
// a lane's eight points form one straight-line block the scheduler can interleave.
#pragma once
#include <hip/hip_runtime.h>

#include "tg_comm.h"

namespace tg {

// x / d, either as the division itself (EXACT) or as x * (1/d) with the reciprocal formed once
// (The EXACT quotient is quot_by of tg_comm.h: the correctly rounded quotient for operands in the normal range from the
// reciprocal that is at hand anyway, at a quarter of the division's issue slots.)
template <bool EXACT>
__device__ __forceinline__ double cdiv(double x, double d, double inv) {
    if constexpr (EXACT) return quot_by(x, d, inv);
    else return x * inv;
}

// launch-constant divisors and their reciprocals
struct Recip {
    double dx, dx2, twodx, dt;
    double idx, idx2, i2dx, idt;
    __host__ __device__ Recip(double dx_, double dt_)
        : dx(dx_), dx2(dx_ * dx_), twodx(2.0 * dx_), dt(dt_), idx(1.0 / dx_), idx2(1.0 / (dx_ * dx_)),
          i2dx(1.0 / (2.0 * dx_)), idt(1.0 / dt_) {}
};

struct Theta {
    double Dmin, Dmax, thr, span, ispan;
    __host__ __device__ Theta(double Dmin_, double Dmax_, double thr_)
        : Dmin(Dmin_), Dmax(Dmax_), thr(thr_), span(Dmax_ - Dmin_), ispan(1.0 / (Dmax_ - Dmin_)) {}
};

// The default theta rule of tango/lodestro_method.py:667-676.  NaN compares false everywhere, which leaves theta = 1 exactly as the reference's masks do.
// c = (1-theta)*cHat cancels to ~1e-14 of theta's rounding, so theta must round exactly as the reference's division
// does.  quot_by qualifies: with the correctly rounded 1/(Dmax - Dmin), formed once by a true division, the product and
// one residual correction are the last two steps of the IEEE division sequence and give the correctly rounded quotient
// (that sequence's operand scaling and special-case fix-up are no-ops for Dmax - DHat and Dmax - Dmin).
template <int FAST = 0>
__device__ __forceinline__ double theta_point(double dh, double dpdx, const Theta &th) {
    const bool small = fabs(dpdx) < th.thr;
    double qm;
    if constexpr (FAST == kDivIeee) qm = (th.Dmax - dh) / th.span;
    else qm = quot_by(th.Dmax - dh, th.span, th.ispan);
    if constexpr (FAST == kDivIeee) {
        const double mid = (1.0 + qm) / 2.0;
        double tv = 1.0;
        tv = (dh < th.Dmin) ? 0.0 : tv;
        tv = (small && dh >= th.Dmin && dh <= th.Dmax) ? mid : tv;
        tv = (small && dh > th.Dmax) ? 0.5 : tv;
        return tv;
    } else {
        // The same values from two comparisons and two selects instead of five and three (the fused kernels evaluate this
        // at every point of every iteration): beyond Dmax the quotient is negative and (1 + max(q, 0)) / 2 is the 1/2
        // wanted there, at Dmax and below the maximum changes nothing.  A NaN DHat gives 1/2 here where the rule above
        // gives 1; theta itself is not an output of the fused kernels, D = theta DHat is NaN either way, and a profile
        // with a NaN coefficient is reported invalid whatever its c.
        const double mid = (1.0 + fmax(qm, 0.0)) / 2.0;
        const double in = small ? mid : 1.0;
        return (dh < th.Dmin) ? 0.0 : in;
    }
}

// tango/lodestro_method.py:624-634 (DHat, cHat, D, c).
// gxx / gradx are the geometric factors on the turbulence grid (1 in Cartesian geometry).  MODE: see quot (tg_comm.h).
template <int MODE = 0>
__device__ __forceinline__ void flux_split_point(double flux, double p, double dpdx, double gxx, double gradx,
                                                 const Theta &th, double &D, double &c, double &DHat,
                                                 double &cHat, double &theta) {
    double dh = quot<MODE>(-flux, gxx * dpdx);
    dh = (dpdx == 0.0) ? 0.0 : dh;
    DHat = dh;
    cHat = quot<MODE>(flux, gradx * p);
    const double tv = theta_point<MODE>(dh, dpdx, th);
    theta = tv;
    D = tv * dh;
    c = (1.0 - tv) * cHat;
}

// D = theta DHat, c = (1 - theta) cHat (tango/lodestro_method.py:631-634) from DHat and cHat formed by the caller
template <int FAST = 0>
__device__ __forceinline__ void split_from_hats(double dh, double cHat, double dpdx, const Theta &th, double &D, double &c) {
    const double tv = theta_point<FAST>(dh, dpdx, th);
    D = tv * dh;
    c = (1.0 - tv) * cHat;
}

// The flux -(dp/dx)^3 / p^2 of tango/extras/shestakov_nonlinear_diffusion.py:59-60 pushed through the split above in
// Cartesian geometry, from the single quotient r = (dp/dx)/p: DHat = -flux/(dp/dx) = r^2 (0 where dp/dx is 0),
// cHat = flux/p = -r^3.  One division instead of three; DHat and cHat differ from the three-division values by
// rounding only (a few ulp); the quotient itself is quot<true> of tg_comm.h.
__device__ __forceinline__ void shestakov_split_point(double p, double dpdx, const Theta &th, double &D, double &c) {
    const double r = quot<true>(dpdx, p);
    const double dh = r * r;
    split_from_hats<true>(dh, -(dh * r), dpdx, th, D, c);
}

// tango/HToMatrixFD.py:277-297: adaptive upwinding weight at a half-integer grid point.
// One division H2h/(c dx) serves both signs of c; the result is discarded where c == 0.
template <int FAST = 0>
__device__ __forceinline__ double fbeta_point(double H2h, double H3h, double dx) {
    const double gamma = 0.9;
    const double c = -H3h;
    const double q = quot<FAST>(H2h, c * dx);
    const double sp = 1.0 - q;   // betaStar for c > 0
    const double sn = -q;        // betaStar for c < 0
    // A NaN betaStar (only from NaN coefficients) gives NaN here and 0 in the reference; either way the row it
    // enters already holds NaN and the profile is reported invalid.
    const double bp = (sp <= 0.5) ? 0.5 : gamma * sp + (1.0 - gamma);
    const double bn = (sn >= 0.5) ? 0.5 : gamma * sn;
    return (c > 0.0) ? bp : ((c < 0.0) ? bn : 0.0);
}

// A face at which the upwinding weight need not be computed, 1/2 standing in for it with the same rows bit for bit:
//  - diffusion dominates, H2 >= |H3| dx / 2 with H3 != 0: betaStar = 1 - H2/(c dx) <= 1/2 for c > 0 and -H2/(c dx) >= 1/2
//    for c < 0, so the weight IS 1/2 whatever the quotient (the test is exact: halving is, and the rounded quotient is
//    monotone in its operands);
//  - H3 == 0 (no convective part at this face: theta = 1 everywhere, as with the example scripts' thetaParams): fbeta gives
//    0 there, but the weight only ever multiplies H3 (fd_row), and a zero times any finite non-negative weight is the same
//    zero, sign included.
// One comparison covers both: H2 is never negative (a diffusivity times positive geometry factors), so H3 == 0 passes
// H2 >= 0 as well; and |H3 (dx/2)| is |H3 dx| / 2 to the bit (a power of two scales exactly).  NaN operands compare false:
// such a face takes the quotient, which gives the same row (and so would a negative H2 at a face with H3 = 0).
__device__ __forceinline__ bool weight_is_moot(double h2, double h3, double dx) {
    return h2 >= fabs(h3 * (0.5 * dx));
}

// tango/derivatives.py:20-24: centred difference with two-point one-sided ends.
// um, u0, up are the values at j-1, j, j+1 (um unused at the first point, up at the last).
template <bool EXACT>
__device__ __forceinline__ double ddx_edge1_point(double um, double u0, double up, bool first, bool last,
                                                  const Recip &rc) {
    const double num = first ? (up - u0) : (last ? (u0 - um) : (up - um));
    if constexpr (EXACT) return quot_by(num, (first || last) ? rc.dx : rc.twodx, (first || last) ? rc.idx : rc.i2dx);
    else return num * ((first || last) ? rc.idx : rc.i2dx);
}

// One row of tango/HToMatrixFD.py:87-117 accumulated in the order H1, H2, H3, H4, H6, H7, boundary.
// h2m/h3m/bm: H2, H3, beta at j-1/2 (pass zeros for the first row: its left-boundary formulas
// :145-146,:165-166 are the interior ones with the j-1/2 terms absent); h2p/h3p/bp at j+1/2.
// g4 is the H4 term already formed (0 when H4 is absent); h6, h7 are 0 when absent.
// `last` selects the Dirichlet row (:200-204).  Returns A (super), B (diag), C (sub), D = -g.
template <bool EXACT>
__device__ __forceinline__ void fd_row(bool last, double H1, double Um1, const Recip &rc, double UL, double h2m,
                                       double h3m, double bm, double h2p, double h3p, double bp, double g4,
                                       double h6, double h7, double &A, double &B, double &C, double &D) {
    double g = cdiv<EXACT>(-H1 * Um1, rc.dt, rc.idt);
    double a = cdiv<EXACT>(-h2p, rc.dx2, rc.idx2) + cdiv<EXACT>(-h3p * (1.0 - bp), rc.dx, rc.idx);
    double b = cdiv<EXACT>(H1, rc.dt, rc.idt) + cdiv<EXACT>(h2m + h2p, rc.dx2, rc.idx2) +
               cdiv<EXACT>(-h3p * bp + h3m * (1.0 - bm), rc.dx, rc.idx);
    double c = cdiv<EXACT>(-h2m, rc.dx2, rc.idx2) + cdiv<EXACT>(h3m * bm, rc.dx, rc.idx);
    b += -h6;
    g += g4;
    g += -h7;
    A = last ? 0.0 : a;
    B = last ? 1.0 : b;
    C = last ? 0.0 : c;
    D = last ? UL : -g;
}

}  // namespace tg
