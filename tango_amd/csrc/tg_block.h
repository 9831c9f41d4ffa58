// Block-tridiagonal version of the chunked elimination: unknowns are K-vectors, coefficients K x K blocks held
// per lane in registers (K = 2: the pair-coupled system of two fields; K = 3: three fully coupled fields).
//
// Lane t owns block rows i = 0..P-1.  With xL the last unknown of the lane to the left and y this lane's last:
//   down sweep : x_i + E_i x_{i+1} = f_i - L_i xL,   Den = B_i - C_i E_{i-1},  E_i = Den^-1 A_i,
//                f_i = Den^-1 (d_i - C_i f_{i-1}),   L_i = -Den^-1 C_i L_{i-1}
//   up sweep   : x_0 = F - L xL - E y  (running values only)
//   reduced    : S_t y_{t-1} + y_t + P_t y_{t+1} = r_t, one block row per lane, block parallel cyclic reduction
//   back subst.: x_i = f_i - L_i y_{t-1} - E_i x_{i+1}
// As in the scalar solver the diagonal-dominance slack of every (block) row is carried explicitly: the slack
// vector of a row is its row sums, and the diagonal of each pivot block is rebuilt from it (diag_from_rowsum).
// The K x K inverses are closed-form (adjugate / determinant) in registers; no pivoting (the transport blocks
// are diagonally dominant).
#pragma once
#include "tg_comm.h"

namespace tg {

template <int K>
struct Mat {
    double m[K][K];
};
template <int K>
struct Vec {
    double v[K];
};

template <int K>
__device__ __forceinline__ Mat<K> mat_zero() {
    Mat<K> r;
#pragma unroll
    for (int a = 0; a < K; ++a)
#pragma unroll
        for (int b = 0; b < K; ++b) r.m[a][b] = 0.0;
    return r;
}
template <int K>
__device__ __forceinline__ Mat<K> mat_identity(double s) {
    Mat<K> r = mat_zero<K>();
#pragma unroll
    for (int a = 0; a < K; ++a) r.m[a][a] = s;
    return r;
}
template <int K>
__device__ __forceinline__ Mat<K> mul(const Mat<K> &x, const Mat<K> &y) {
    Mat<K> r;
#pragma unroll
    for (int a = 0; a < K; ++a)
#pragma unroll
        for (int b = 0; b < K; ++b) {
            double s = 0.0;
#pragma unroll
            for (int c = 0; c < K; ++c) s += x.m[a][c] * y.m[c][b];
            r.m[a][b] = s;
        }
    return r;
}
template <int K>
__device__ __forceinline__ Vec<K> mul(const Mat<K> &x, const Vec<K> &y) {
    Vec<K> r;
#pragma unroll
    for (int a = 0; a < K; ++a) {
        double s = 0.0;
#pragma unroll
        for (int c = 0; c < K; ++c) s += x.m[a][c] * y.v[c];
        r.v[a] = s;
    }
    return r;
}
template <int K>
__device__ __forceinline__ Mat<K> sub(const Mat<K> &x, const Mat<K> &y) {
    Mat<K> r;
#pragma unroll
    for (int a = 0; a < K; ++a)
#pragma unroll
        for (int b = 0; b < K; ++b) r.m[a][b] = x.m[a][b] - y.m[a][b];
    return r;
}
template <int K>
__device__ __forceinline__ Vec<K> sub(const Vec<K> &x, const Vec<K> &y) {
    Vec<K> r;
#pragma unroll
    for (int a = 0; a < K; ++a) r.v[a] = x.v[a] - y.v[a];
    return r;
}
template <int K>
__device__ __forceinline__ Mat<K> neg(const Mat<K> &x) {
    Mat<K> r;
#pragma unroll
    for (int a = 0; a < K; ++a)
#pragma unroll
        for (int b = 0; b < K; ++b) r.m[a][b] = -x.m[a][b];
    return r;
}

template <int K>
__device__ __forceinline__ Vec<K> rowsum(const Mat<K> &x) {
    Vec<K> r;
#pragma unroll
    for (int a = 0; a < K; ++a) {
        double s = 0.0;
#pragma unroll
        for (int b = 0; b < K; ++b) s += x.m[a][b];
        r.v[a] = s;
    }
    return r;
}
template <int K>
__device__ __forceinline__ Vec<K> add(const Vec<K> &x, const Vec<K> &y) {
    Vec<K> r;
#pragma unroll
    for (int a = 0; a < K; ++a) r.v[a] = x.v[a] + y.v[a];
    return r;
}
// Replace the diagonal of G by rs_k - sum of the off-diagonal entries of row k, rs being the wanted row sums.
// The pivot blocks of the transport systems are M-matrices with tiny row sums (the diagonal-dominance slack);
// their diagonal formed directly (I - ...) loses that slack to cancellation, rebuilt from carried row sums
// every term has one sign.
template <int K>
__device__ __forceinline__ void diag_from_rowsum(Mat<K> &G, const Vec<K> &rs) {
#pragma unroll
    for (int a = 0; a < K; ++a) {
        double off = 0.0;
#pragma unroll
        for (int b = 0; b < K; ++b) off += (b == a) ? 0.0 : G.m[a][b];
        G.m[a][a] = rs.v[a] - off;
    }
}

template <int FAST = 0>
__device__ __forceinline__ Mat<1> inverse(const Mat<1> &x, int &singular) {
    singular |= (x.m[0][0] == 0.0);
    Mat<1> r;
    r.m[0][0] = recip<FAST>(x.m[0][0]);
    return r;
}
template <int FAST = 0>
__device__ __forceinline__ Mat<2> inverse(const Mat<2> &x, int &singular) {
    const double det = x.m[0][0] * x.m[1][1] - x.m[0][1] * x.m[1][0];
    singular |= (det == 0.0);
    const double inv = recip<FAST>(det);
    Mat<2> r;
    r.m[0][0] = x.m[1][1] * inv;
    r.m[0][1] = -x.m[0][1] * inv;
    r.m[1][0] = -x.m[1][0] * inv;
    r.m[1][1] = x.m[0][0] * inv;
    return r;
}
template <int FAST = 0>
__device__ __forceinline__ Mat<3> inverse(const Mat<3> &x, int &singular) {
    const double c00 = x.m[1][1] * x.m[2][2] - x.m[1][2] * x.m[2][1];
    const double c01 = x.m[1][2] * x.m[2][0] - x.m[1][0] * x.m[2][2];
    const double c02 = x.m[1][0] * x.m[2][1] - x.m[1][1] * x.m[2][0];
    const double det = x.m[0][0] * c00 + x.m[0][1] * c01 + x.m[0][2] * c02;
    singular |= (det == 0.0);
    const double inv = recip<FAST>(det);
    Mat<3> r;
    r.m[0][0] = c00 * inv;
    r.m[1][0] = c01 * inv;
    r.m[2][0] = c02 * inv;
    r.m[0][1] = (x.m[0][2] * x.m[2][1] - x.m[0][1] * x.m[2][2]) * inv;
    r.m[1][1] = (x.m[0][0] * x.m[2][2] - x.m[0][2] * x.m[2][0]) * inv;
    r.m[2][1] = (x.m[0][1] * x.m[2][0] - x.m[0][0] * x.m[2][1]) * inv;
    r.m[0][2] = (x.m[0][1] * x.m[1][2] - x.m[0][2] * x.m[1][1]) * inv;
    r.m[1][2] = (x.m[0][2] * x.m[1][0] - x.m[0][0] * x.m[1][2]) * inv;
    r.m[2][2] = (x.m[0][0] * x.m[1][1] - x.m[0][1] * x.m[1][0]) * inv;
    return r;
}

// a reduced block row: S (K*K), Pm (K*K), slack dl (K), r (K) flattened for the lane exchange
template <int K>
struct RowPack {
    static constexpr int N = 2 * K * K + 2 * K;
    double q[N];
    __device__ __forceinline__ void set(const Mat<K> &S, const Mat<K> &Pm, const Vec<K> &dl, const Vec<K> &r) {
#pragma unroll
        for (int a = 0; a < K; ++a)
#pragma unroll
            for (int b = 0; b < K; ++b) {
                q[a * K + b] = S.m[a][b];
                q[K * K + a * K + b] = Pm.m[a][b];
            }
#pragma unroll
        for (int a = 0; a < K; ++a) {
            q[2 * K * K + a] = dl.v[a];
            q[2 * K * K + K + a] = r.v[a];
        }
    }
    __device__ __forceinline__ void get(Mat<K> &S, Mat<K> &Pm, Vec<K> &dl, Vec<K> &r) const {
#pragma unroll
        for (int a = 0; a < K; ++a)
#pragma unroll
            for (int b = 0; b < K; ++b) {
                S.m[a][b] = q[a * K + b];
                Pm.m[a][b] = q[K * K + a * K + b];
            }
#pragma unroll
        for (int a = 0; a < K; ++a) {
            dl.v[a] = q[2 * K * K + a];
            r.v[a] = q[2 * K * K + K + a];
        }
    }
};

// Block parallel cyclic reduction of S_t y_{t-h} + y_t + P_t y_{t+h} = r_t over the T lanes of group g.
// dl_t = (S_t + I + P_t) 1 is the slack vector of the row, carried as in the scalar reduction.
template <int K, int FAST = 0, class G>
__device__ __forceinline__ Vec<K> block_pcr(const G &g, Mat<K> S, Mat<K> Pm, Vec<K> dl, Vec<K> r, int &singular) {
    // No special cases at the ends: the first lane has no left coupling (S = 0, enforced here) and by induction S = 0
    // in every lane t < h at the level of distance h (S2 = -S Sm with Sm from a lane below h), likewise Pm = 0 in the
    // lanes t >= T - h; whatever finite rows those lanes receive instead of a neighbour's is multiplied by exact zeros.
    if (g.t() == 0) S = mat_zero<K>();
    if (g.t() == G::T - 1) Pm = mat_zero<K>();
#pragma unroll
    for (int h = 1; h < G::T; h <<= 1) {
        RowPack<K> me, lf, rt;
        me.set(S, Pm, dl, r);
        g.template exchange_nofill<RowPack<K>::N>(me.q, h, lf.q, rt.q);   // rows of lanes t-h and t+h
        Mat<K> Sm, Pmm, Sp, Pp;
        Vec<K> dm, dp, rm, rp;
        lf.get(Sm, Pmm, dm, rm);
        rt.get(Sp, Pp, dp, rp);
        // (-S Sm) y_{t-2h} + (I - S Pmm - Pm Sp) y_t + (-Pm Pp) y_{t+2h} = r - S rm - Pm rp
        const Mat<K> S2 = neg(mul(S, Sm));
        const Mat<K> P2 = neg(mul(Pm, Pp));
        const Vec<K> d2 = sub(sub(dl, mul(S, dm)), mul(Pm, dp));
        Mat<K> Gm = sub(sub(mat_identity<K>(1.0), mul(S, Pmm)), mul(Pm, Sp));
        diag_from_rowsum(Gm, sub(sub(d2, rowsum(S2)), rowsum(P2)));
        const Mat<K> Gi = inverse<FAST>(Gm, singular);
        r = mul(Gi, sub(sub(r, mul(S, rm)), mul(Pm, rp)));
        S = mul(Gi, S2);
        Pm = mul(Gi, P2);
        dl = mul(Gi, d2);
    }
    return r;
}

// FAST: the division mode of the determinants' reciprocals (tg_comm.h: kDivIeee, kDivFast, kDivRounded).
template <int K, int P, int FAST = 0>
struct BlockElim {
    Mat<K> E[P], L[P];
    Vec<K> f[P], rho[P];     // rho_i = (I + E_i + L_i) 1, the slack vector of the eliminated row
    int singular = 0;

    // block row i (call from a fully unrolled loop): C x_{i-1} + B x_i + A x_{i+1} = d
    __device__ __forceinline__ void push(int i, const Mat<K> &C, const Mat<K> &B, const Mat<K> &A, const Vec<K> &d) {
        const Vec<K> dlt = rowsum(sub(sub(B, neg(A)), neg(C)));   // ((B + A) + C) 1
        Mat<K> Den = B;
        Vec<K> rhs = d;
        Mat<K> CL = C;    // i = 0: -C L_{-1} with L_{-1} = -I
        Vec<K> tv = dlt;  // dlt - C rho_{i-1}
        Vec<K> cl1;       // C (L_{i-1} 1)
        if (i > 0) {
            const int k = i > 0 ? i - 1 : 0;
            Den = sub(B, mul(C, E[k]));
            rhs = sub(d, mul(C, f[k]));
            CL = neg(mul(C, L[k]));
            tv = sub(dlt, mul(C, rho[k]));
            cl1 = mul(C, rowsum(L[k]));
        } else {
            cl1 = rowsum(neg(C));
        }
        // Den 1 = dlt - A 1 - C rho_{i-1} + C L_{i-1} 1
        diag_from_rowsum(Den, add(sub(tv, rowsum(A)), cl1));
        const Mat<K> Di = inverse<FAST>(Den, singular);
        E[i] = mul(Di, A);
        f[i] = mul(Di, rhs);
        L[i] = mul(Di, CL);
        rho[i] = mul(Di, tv);
    }

    // The same for DIAGONAL off-diagonal blocks C = diag(c), A = diag(a), as the transport systems have them (fields couple
    // only through B: tango/JK_to_matrix_fd.py:80-82): the products with the structural zeros of C and A, which the compiler
    // may not drop (0 x is not 0 for every x), are not formed.  For finite operands the results are those of push, bit for
    // bit: the dropped terms are exact zeros added to a sum.
    __device__ __forceinline__ void push_diag(int i, const Vec<K> &c, const Mat<K> &B, const Vec<K> &a, const Vec<K> &d) {
        Vec<K> dlt;
#pragma unroll
        for (int r = 0; r < K; ++r) {
            double s = 0.0;
#pragma unroll
            for (int q = 0; q < K; ++q) s += (q == r) ? (B.m[r][q] + a.v[r]) + c.v[r] : B.m[r][q];
            dlt.v[r] = s;
        }
        Mat<K> Den = B, CL;
        Vec<K> rhs = d, tv = dlt, cl1;
        const int k = i > 0 ? i - 1 : 0;
#pragma unroll
        for (int r = 0; r < K; ++r) {
            if (i > 0) {
                double ls = 0.0;
#pragma unroll
                for (int q = 0; q < K; ++q) {
                    Den.m[r][q] = B.m[r][q] - c.v[r] * E[k].m[r][q];
                    CL.m[r][q] = -(c.v[r] * L[k].m[r][q]);
                    ls += L[k].m[r][q];
                }
                rhs.v[r] = d.v[r] - c.v[r] * f[k].v[r];
                tv.v[r] = dlt.v[r] - c.v[r] * rho[k].v[r];
                cl1.v[r] = c.v[r] * ls;
            } else {
#pragma unroll
                for (int q = 0; q < K; ++q) CL.m[r][q] = (q == r) ? c.v[r] : 0.0;      // -C L_{-1} with L_{-1} = -I
                cl1.v[r] = -c.v[r];
            }
        }
        Vec<K> rs;
#pragma unroll
        for (int r = 0; r < K; ++r) rs.v[r] = (tv.v[r] - a.v[r]) + cl1.v[r];
        diag_from_rowsum(Den, rs);
        const Mat<K> Di = inverse<FAST>(Den, singular);
#pragma unroll
        for (int r = 0; r < K; ++r)
#pragma unroll
            for (int q = 0; q < K; ++q) E[i].m[r][q] = Di.m[r][q] * a.v[q];
        f[i] = mul(Di, rhs);
        if (i > 0) {
            L[i] = mul(Di, CL);
        } else {
#pragma unroll
            for (int r = 0; r < K; ++r)
#pragma unroll
                for (int q = 0; q < K; ++q) L[i].m[r][q] = Di.m[r][q] * c.v[q];
        }
        rho[i] = mul(Di, tv);
    }

    template <class G>
    __device__ __forceinline__ void finish(const G &g, Vec<K> (&x)[P]) {
        Mat<K> S, Pm;
        Vec<K> dl, r;
        if constexpr (P > 1) {
            Vec<K> F = f[P - 2], tau = rho[P - 2];
            Mat<K> Lr = L[P - 2], Er = E[P - 2];
#pragma unroll
            for (int i = P - 3; i >= 0; --i) {
                F = sub(f[i], mul(E[i], F));
                Lr = sub(L[i], mul(E[i], Lr));
                tau = sub(rho[i], mul(E[i], tau));
                Er = neg(mul(E[i], Er));
            }
            // first-row form of the lane to the right (an identity row past the end: zeros, slack 1)
            RowPack<K> me, lf, rt;
            me.set(Lr, Er, tau, F);
            g.template exchange<RowPack<K>::N>(me.q, 1, lf.q, rt.q);
            Mat<K> L0, E0;
            Vec<K> F0, T0;
            rt.get(L0, E0, T0, F0);
            if (g.t() + 1 >= G::T) T0 = rowsum(mat_identity<K>(1.0));
            const Mat<K> &El = E[P - 1];
            const Vec<K> sig = sub(rho[P - 1], mul(El, T0));
            const Mat<K> S2 = L[P - 1];
            const Mat<K> P2 = neg(mul(El, E0));
            Mat<K> Gm = sub(mat_identity<K>(1.0), mul(El, L0));
            diag_from_rowsum(Gm, sub(sub(sig, rowsum(S2)), rowsum(P2)));
            const Mat<K> Gi = inverse<FAST>(Gm, singular);
            S = mul(Gi, S2);
            Pm = mul(Gi, P2);
            dl = mul(Gi, sig);
            r = mul(Gi, sub(f[P - 1], mul(El, F0)));
        } else {
            S = L[0];
            Pm = E[0];
            dl = rho[0];
            r = f[0];
        }
        const Vec<K> y = block_pcr<K, FAST>(g, S, Pm, dl, r, singular);
        x[P - 1] = y;
        if constexpr (P > 1) {
            double yq[K], ylq[K], yrq[K];
#pragma unroll
            for (int a = 0; a < K; ++a) yq[a] = y.v[a];
            g.template exchange<K>(yq, 1, ylq, yrq);
            Vec<K> yl;
#pragma unroll
            for (int a = 0; a < K; ++a) yl.v[a] = ylq[a];
#pragma unroll
            for (int i = P - 2; i >= 0; --i) x[i] = sub(sub(f[i], mul(L[i], yl)), mul(E[i], x[i + 1]));
        }
    }
};

}  // namespace tg
