// Chunked tridiagonal elimination shared by every scalar profile solver.
//
// Lane t of a group owns rows i = 0..P-1 (grid points t*P+i).  With xL the last unknown of the
// lane to the left and y the last unknown of this lane:
//   down sweep : x_i + e_i x_{i+1} = f_i - l_i xL            (Thomas with one fill-in column)
//   up sweep   : x_0 = F - L xL - E y                         (only the first row's form is needed)
//   reduced    : one row per lane in y_t, unit diagonal, solved by parallel cyclic reduction
//   back subst.: x_i = f_i - l_i y_{t-1} - e_i x_{i+1}, i = P-2..0
// No pivoting: the Picard matrices are strictly diagonally dominant M-matrices.  Their diagonal
// dominance is tiny (row sums ~1/dt against entries ~1/dx^2), and plain cyclic reduction loses that
// slack to cancellation at every level; here the slack of every row is carried explicitly through the
// sweeps and the reduction (see ChunkElim::push and the pcr members of the lane groups), which keeps
// the residual of the computed solution at the level of a sequential Thomas sweep / LAPACK dgtsv.
//
// Rows are pushed one at a time so that a caller can assemble row i and eliminate it before it
// forms row i+1: only e, f, l (3 doubles per row) stay live, not the four input diagonals.
#pragma once
#include "tg_comm.h"

namespace tg {

// FAST: the division mode of the pivots' reciprocals (tg_comm.h: kDivIeee, kDivFast, kDivRounded).
template <int P, int FAST = 0>
struct ChunkElim {
    double e[P], f[P], l[P], rho[P];
    int singular = 0;

    // c = sub-diagonal, b = diagonal, a = super-diagonal, d = right-hand side of local row i.
    // Call from a fully unrolled loop (i must fold to a constant so that the arrays stay in registers).
    //
    // Besides e, f, l the sweep carries rho_i = 1 + e_i + l_i, the slack of the eliminated row, from the
    // slack dlt = a + b + c of the original row: rho_i = (dlt - c rho_{i-1}) / den and
    // den = dlt - a - c rho_{i-1} + c l_{i-1} (= b - c e_{i-1}).  For an M-matrix every term of these sums
    // is non-negative, so the small slack (1/dt against 1/dx^2 entries) is not lost to cancellation; for any
    // other matrix the expressions are the same numbers written differently.
    __device__ __forceinline__ void push(int i, double c, double b, double a, double d) { push_slack(i, c, b, a, d, (b + a) + c); }

    // the same with the row's slack dlt = a + b + c supplied by the caller (rows whose slack is known more
    // accurately than their coefficients give it, e.g. the rows of a reduced system)
    __device__ __forceinline__ void push_slack(int i, double c, double b, double a, double d, double dlt) {
        const double fp = (i == 0) ? 0.0 : f[i > 0 ? i - 1 : 0];
        const double lp = (i == 0) ? -1.0 : l[i > 0 ? i - 1 : 0];
        const double rp = (i == 0) ? 0.0 : rho[i > 0 ? i - 1 : 0];
        const double t = dlt - c * rp;
        const double den = (t - a) + c * lp;
        singular |= (den == 0.0);
        const double inv = recip<FAST>(den);
        e[i] = a * inv;
        f[i] = (d - c * fp) * inv;
        l[i] = -(c * lp) * inv;
        rho[i] = t * inv;
    }

    // Solve; x[0..P-1] receives this lane's unknowns.
    template <class G>
    __device__ __forceinline__ void finish(const G &g, double (&x)[P]) {
        double s, p, dl, r;
        if constexpr (P > 1) {
            // up sweep, keeping only the running values: x_0 = F - L xL - E y, tau = 1 + L + E
            double F = f[P - 2], L = l[P - 2], E = e[P - 2], tau = rho[P - 2];
#pragma unroll
            for (int i = P - 3; i >= 0; --i) {
                F = f[i] - e[i] * F;
                L = l[i] - e[i] * L;
                tau = rho[i] - e[i] * tau;
                E = -e[i] * E;
            }
            (void)L;
            // first-row coefficients of the lane to the right (an identity row past the end)
            double F0, E0, T0;
            if constexpr (G::kBlock) {      // one exchange (one pair of barriers) instead of three
                const double v[3] = {F, E, tau};
                double lq[3], rq[3];
                g.template exchange<3>(v, 1, lq, rq);
                F0 = rq[0];
                E0 = rq[1];
                T0 = (g.t() + 1 < G::T) ? rq[2] : 1.0;
            } else {
                F0 = g.from_right(F, 1, 0.0);
                E0 = g.from_right(E, 1, 0.0);
                T0 = g.from_right(tau, 1, 1.0);
            }
            // reduced row in y: ll y_{t-1} + (1 - el L0) y_t - el E0 y_{t+1} = fl - el F0, slack = rl - el T0
            const double el = e[P - 1], ll = l[P - 1], fl = f[P - 1], rl = rho[P - 1];
            const double sig = rl - el * T0;
            const double s2 = ll;
            const double p2 = -(el * E0);
            const double gg = sig - s2 - p2;
            singular |= (gg == 0.0);
            const double inv = recip<FAST>(gg);
            s = s2 * inv;
            p = p2 * inv;
            dl = sig * inv;
            r = (fl - el * F0) * inv;
        } else {
            s = l[0];
            p = e[0];
            dl = rho[0];
            r = f[0];
        }
        const double y = g.template pcr<FAST>(s, p, dl, r, singular);
        x[P - 1] = y;
        if constexpr (P > 1) {
            // back substitution with the down-sweep rows, as a sequential Thomas sweep would do inside the
            // chunk: x_i = f_i - l_i xL - e_i x_{i+1}; every interior row is then satisfied to rounding
            const double yl = g.from_left(y, 1, 0.0);
#pragma unroll
            for (int i = P - 2; i >= 0; --i) x[i] = (f[i] - l[i] * yl) - e[i] * x[i + 1];
        }
    }
};

// Array form: on return d holds the solution.
template <int P, class G>
__device__ __forceinline__ void chunk_solve(const G &g, const double (&c)[P], const double (&b)[P],
                                            const double (&a)[P], double (&d)[P], int &singular) {
    ChunkElim<P> el;
#pragma unroll
    for (int i = 0; i < P; ++i) el.push(i, c[i], b[i], a[i], d[i]);
    el.finish(g, d);
    singular |= el.singular;
}

}  // namespace tg
