// Chunked tridiagonal / block-tridiagonal elimination shared by every profile solver.
//
// Lane t of a group owns rows i = 0..P-1 (grid points t*P+i).  With xL the last unknown of the
// lane to the left and y the last unknown of this lane:
//   down sweep : x_i + e_i x_{i+1} = f_i - l_i xL            (Thomas with one fill-in column)
//   up sweep   : x_i = F_i - L_i xL - E_i y      for i <= P-2
//   reduced    : one row per lane in y_t, unit diagonal, solved by parallel cyclic reduction
//   back subst.: x_i from F, L, E with xL = y_{t-1}
// No pivoting: the Picard matrices are strictly diagonally dominant M-matrices, for which this
// elimination is as accurate as LAPACK's pivoted dgtsv (checked against extended precision in
// tests/test_partition_model.py).
#pragma once
#include "tg_comm.h"

namespace tg {

// Scalar system.  c = sub-diagonal, b = diagonal, a = super-diagonal, d = right-hand side.
// On return d holds the solution.  a, b, c are destroyed.  `singular` is OR-ed with 1 on a zero pivot.
template <int P, class G>
__device__ __forceinline__ void chunk_solve(const G &g, double (&c)[P], double (&b)[P], double (&a)[P],
                                            double (&d)[P], int &singular) {
    // reuse the input registers: a -> e/E, d -> f/F, c -> l/L
    {
        singular |= (b[0] == 0.0);
        double inv = 1.0 / b[0];
        a[0] *= inv;
        d[0] *= inv;
        c[0] *= inv;
    }
#pragma unroll
    for (int i = 1; i < P; ++i) {
        double den = b[i] - c[i] * a[i - 1];
        singular |= (den == 0.0);
        double inv = 1.0 / den;
        a[i] *= inv;
        d[i] = (d[i] - c[i] * d[i - 1]) * inv;
        c[i] = -(c[i] * c[i - 1]) * inv;
    }
    double s, p, r;
    if constexpr (P > 1) {
        const double el = a[P - 1], ll = c[P - 1], fl = d[P - 1];
#pragma unroll
        for (int i = P - 3; i >= 0; --i) {
            d[i] = d[i] - a[i] * d[i + 1];
            c[i] = c[i] - a[i] * c[i + 1];
            a[i] = -a[i] * a[i + 1];
        }
        // first-row coefficients of the lane to the right
        double F0 = g.from_right(d[0], 1, 0.0);
        double L0 = g.from_right(c[0], 1, 0.0);
        double E0 = g.from_right(a[0], 1, 0.0);
        double gg = 1.0 - el * L0;
        singular |= (gg == 0.0);
        double inv = 1.0 / gg;
        s = ll * inv;
        p = -(el * E0) * inv;
        r = (fl - el * F0) * inv;
    } else {
        s = c[0];
        p = a[0];
        r = d[0];
    }
    double y = g.pcr(s, p, r, singular);
    if constexpr (P > 1) {
        double yl = g.from_left(y, 1, 0.0);
#pragma unroll
        for (int i = 0; i < P - 1; ++i) d[i] = d[i] - c[i] * yl - a[i] * y;
    }
    d[P - 1] = y;
}

}  // namespace tg
