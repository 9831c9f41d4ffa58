// libtango_hip.so -- batched fp64 kernels for Tango's implicit transport inner loop, written for
// MI355X (gfx950: 64-wide wavefronts, 256 CUs, HBM-bound fp64 stencil + tridiagonal work; no MFMA,
// nothing on this path is a dense contraction).  The C ABI is declared in include/tango_hip.h.
//
// Data layout.  Persistent state is point-fastest, X[b*n + j]: a profile is a contiguous row, so
// one wavefront streams one profile with full-line accesses and every array is read and written
// exactly once per Picard iteration.  A wavefront (n <= 512) or a workgroup (n <= 16384) owns a
// profile; each lane keeps P = 8 consecutive grid points in registers from the first load to the
// last store: the forward-sweep intermediates of the tridiagonal solve never touch memory.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>

#include "../../include/tango_hip.h"
#include "tg_assemble.h"
#include "tg_comm.h"
#include "tg_host.h"
#include "tg_io.h"
#include "tg_solve.h"
#include "tg_wave_io.h"

thread_local char tg_err_buf[512] = "";

namespace {

using namespace tg;

// ------------------------------------------------------------------------------------------
// Tridiagonal solve, point-fastest: one lane group per profile.
// ------------------------------------------------------------------------------------------
// rows can go through the coalesced path of tg_wave_io.h when a wavefront owns whole profiles, 512 points in all
template <int P, class G, bool FULL>
constexpr bool kWaveCoop = FULL && P == 8 && !(G::kBlock && G::T > 512);   // 1024 threads: no LDS left for tiles
// LDS tiles per wavefront for that path: four, or two in a workgroup-per-profile kernel, whose exchange area shares
// the LDS (four tiles there cost a resident workgroup per CU and measured slower)
template <class G>
constexpr int kCoopTiles = G::kBlock ? 2 : 4;

template <int P, class G, bool FULL>
__global__ void __launch_bounds__(G::kBlock ? G::T : 256)
tridiag_pf_kernel(const double *__restrict__ A, const double *__restrict__ B, const double *__restrict__ C,
                  const double *__restrict__ D, double *__restrict__ u, int64_t nb, int n,
                  int32_t *__restrict__ status) {
    extern __shared__ double lds[];
    G g(lds);
    const int64_t b = (int64_t)blockIdx.x * G::profiles_per_block(blockDim.x) + g.profile_in_block();
    // coalesced rows through the wavefront's LDS tiles when the wavefront owns 512 consecutive doubles of every array
    // (one profile per wavefront or workgroup, or several short profiles that all exist)
    constexpr bool COOP_T = kWaveCoop<P, G, FULL>;
    constexpr bool COOP_ALWAYS = COOP_T && (G::kBlock || G::T == 64);
    const bool exists = b < nb;
    const bool coop = COOP_ALWAYS || (COOP_T && __all(exists));
    if (!exists) return;
    const int j0 = g.t() * P;
    const int64_t off = b * (int64_t)n;
    const int64_t woff = G::kBlock ? off + (int64_t)wave_in_block() * 512 : off - (int64_t)((threadIdx.x & 63) / G::T) * n;
    using Tile = WaveTile<8>;
    constexpr int NT = 2;      // two tiles per wavefront keep four wavefronts per SIMD resident
    Tile wt(lds + (COOP_T ? (G::kBlock ? 8 * G::T : 0) + wave_in_block() * NT * Tile::kDoubles : 0));
    double a[P], bb[P], c[P], d[P];
    if (COOP_T && coop) {
        double2 r0[4], r1[4], r2[4], r3[4];
        wt.fetch(A + woff, r0);
        wt.fetch(B + woff, r1);
        wt.fetch(C + woff, r2);
        wt.fetch(D + woff, r3);
        wt.stage(0, r0);
        wt.stage(1, r1);
        if constexpr (NT >= 4) {
            wt.stage(2, r2);
            wt.stage(3, r3);
        }
        Tile::sync();
        wt.take(0, a);
        wt.take(1, bb);
        if constexpr (NT >= 4) {
            wt.take(2, c);
            wt.take(3, d);
        } else {
            Tile::sync();
            wt.stage(0, r2);
            wt.stage(1, r3);
            Tile::sync();
            wt.take(0, c);
            wt.take(1, d);
        }
        Tile::sync();
    } else {
        load_chunk<P, FULL>(A + off, n, j0, a, 0.0);
        load_chunk<P, FULL>(B + off, n, j0, bb, 1.0);
        load_chunk<P, FULL>(C + off, n, j0, c, 0.0);
        load_chunk<P, FULL>(D + off, n, j0, d, 0.0);
    }
#pragma unroll
    for (int i = 0; i < P; ++i) {
        c[i] = (j0 + i == 0) ? 0.0 : c[i];
        a[i] = (j0 + i == n - 1) ? 0.0 : a[i];
    }
    int sing = 0;
    chunk_solve<P>(g, c, bb, a, d, sing);
    if (COOP_T && coop) {
        wt.put(0, d);
        Tile::sync();
        wt.flush(0, u + woff);
    } else {
        store_chunk<P, FULL>(u + off, n, j0, d);
    }
    if (status) {
        sing = g.any2(sing);
        if (sing && g.t() == 0) atomicOr(&status[b], TG_ST_SINGULAR);
    }
}

// Tridiagonal solve, batch-fastest: one lane per profile, every access of a sweep coalesced across the batch.
//
// The sweeps are LAPACK dgtsv's (the routine scipy.linalg.solve_banded reaches from tango/tridiag.py:94-96),
// operation by operation, partial pivoting included, with floating-point contraction switched off: for finite data
// the result is the reference's bit for bit.  Row i of the forward sweep compares |d_i| with the sub-diagonal entry
// below it; without an interchange it eliminates with fact = dl_i / d_i, with one it swaps rows i and i+1, which
// puts an entry on the second super-diagonal (du2).  The pivots d go to the work space and the modified right-hand
// side to u.  The modified super-diagonal and du2 are written only from the first row on at which some profile of
// the wavefront needed an interchange (`sw`, uniform over the wavefront); before that row du is the caller's A and
// du2 is zero, which is every row for the diagonally dominant matrices of the transport problem.
// U rows are fetched ahead of the arithmetic: the sweeps are one long dependent chain per lane and a batch of
// 65 536 profiles is one wavefront per SIMD, so nothing else hides the memory latency.
template <int U>
__global__ void __launch_bounds__(256)
tridiag_bf_kernel(const double *__restrict__ A, const double *__restrict__ B, const double *__restrict__ C,
                  const double *D, double *u, int64_t nb, int n, double *__restrict__ wd,
                  double *__restrict__ wdu, double *__restrict__ wdu2, int32_t *__restrict__ status) {
    // (u may alias D: every row of D is fetched before the same row of u is written, in program order)
#pragma clang fp contract(off)
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    int info = 0;
    int sw = n;                         // first row whose du / du2 live in the work space
    double d = B[b], rhs = D[b], du = (n > 1) ? A[b] : 0.0;
    // rows i+1 .. i+U of C, B, D, A, fetched one block ahead
    double cq[U], bq[U], dq[U], aq[U];
    auto fetch = [&](int i0, double (&c_)[U], double (&b_)[U], double (&d_)[U], double (&a_)[U]) {
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int r = i0 + 1 + q;
            const bool in = r < n;
            const int64_t k = (int64_t)(in ? r : n - 1) * nb + b;
            c_[q] = C[k];
            b_[q] = B[k];
            d_[q] = D[k];
            a_[q] = (r < n - 1) ? A[k] : 0.0;
        }
    };
    fetch(0, cq, bq, dq, aq);
    for (int i0 = 0; i0 < n - 1; i0 += U) {
        double cn[U], bn_[U], dn_[U], an[U];
        if (i0 + U < n - 1) fetch(i0 + U, cn, bn_, dn_, an);
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int i = i0 + q;
            if (i < n - 1) {
                const double dl = cq[q];
                double dnext = bq[q], bnext = dq[q], dunext = aq[q], du2 = 0.0;
                const bool piv = !(fabs(d) >= fabs(dl));
                if (sw == n && __any(piv)) sw = i;
                if (!piv) {
                    info |= (d == 0.0);
                    const double fact = dl / d;
                    dnext = dnext - fact * du;
                    bnext = bnext - fact * rhs;
                } else {
                    const double fact = d / dl;
                    d = dl;
                    double temp = dnext;
                    dnext = du - fact * temp;
                    if (i < n - 2) {
                        du2 = dunext;
                        dunext = -fact * du2;
                    }
                    du = temp;
                    temp = rhs;
                    rhs = bnext;
                    bnext = temp - fact * bnext;
                }
                const int64_t k = (int64_t)i * nb + b;
                wd[k] = d;
                u[k] = rhs;
                if (i >= sw) {
                    wdu[k] = du;
                    wdu2[k] = du2;
                }
                d = dnext;
                rhs = bnext;
                du = dunext;
            }
        }
#pragma unroll
        for (int q = 0; q < U; ++q) {
            cq[q] = cn[q];
            bq[q] = bn_[q];
            dq[q] = dn_[q];
            aq[q] = an[q];
        }
    }
    info |= (d == 0.0);
    // back substitution: x_i = (b_i - du_i x_{i+1} - du2_i x_{i+2}) / d_i (no du2 term in the last two rows)
    double x1 = rhs / d, x2 = 0.0;
    u[(int64_t)(n - 1) * nb + b] = x1;
    double pd[U], pb[U], pu[U], p2[U];
    auto fetch_back = [&](int ihi, double (&d_)[U], double (&b_)[U], double (&u_)[U], double (&v_)[U]) {
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int r = ihi - q;
            const int rc = r >= 0 ? r : 0;
            const int64_t k = (int64_t)rc * nb + b;
            d_[q] = wd[k];
            b_[q] = u[k];
            if (rc >= sw) {
                u_[q] = wdu[k];
                v_[q] = wdu2[k];
            } else {
                u_[q] = A[k];
                v_[q] = 0.0;
            }
        }
    };
    if (n > 1) fetch_back(n - 2, pd, pb, pu, p2);
    for (int ihi = n - 2; ihi >= 0; ihi -= U) {
        double nd[U], nbv[U], nu[U], n2[U];
        if (ihi - U >= 0) fetch_back(ihi - U, nd, nbv, nu, n2);
#pragma unroll
        for (int q = 0; q < U; ++q) {
            const int i = ihi - q;
            if (i >= 0) {
                double x;
                if (i == n - 2) x = (pb[q] - pu[q] * x1) / pd[q];
                else x = (pb[q] - pu[q] * x1 - p2[q] * x2) / pd[q];
                u[(int64_t)i * nb + b] = x;
                x2 = x1;
                x1 = x;
            }
        }
#pragma unroll
        for (int q = 0; q < U; ++q) {
            pd[q] = nd[q];
            pb[q] = nbv[q];
            pu[q] = nu[q];
            p2[q] = n2[q];
        }
    }
    if (status && info) atomicOr(&status[b], TG_ST_SINGULAR);
}

// Tridiagonal solve, batch-fastest arrays, solved by the point-fastest algorithm (n <= 512).
//
// A workgroup of W = 16 wavefronts takes 16 neighbouring profiles: in the interleaved layout their entries of one row
// are a whole 128-byte line, and a wavefront instruction moves four such rows.  Thread 64 w + l of pass r holds element
// (row 64 r + 4 w + l/16, profile l%16).  The rows go through 16 LDS tiles, one per profile, in the layout of
// tg_wave_io.h (tile bases 16 bytes apart modulo the bank period); wavefront w then takes profile w out of its tile,
// eight consecutive points per lane, and solves it exactly as tridiag_pf_kernel does: the results are that kernel's
// bit for bit.  40 bytes per point, against 80 for the dgtsv sweeps above, which have to park the pivots and the
// modified right-hand side in memory between the two sweeps.  What bounds it is the access shape, not the solve: a
// workgroup reads a strip one line wide and 512 rows tall, rows a whole batch apart (the same kernel without the
// solve takes as long), and the two tile buffers of 16 profiles leave room for one workgroup per CU.  Measured at
// 65 536 x 512: 0.355 ms; with W = 8 (64-byte pieces of a row, two workgroups per CU) 0.645 ms, 0.443 ms when the two
// workgroups that share a line are sent to the same XCD at the same time.  It is the faster way to solve arrays that
// ARE interleaved; arrays that can be laid out either way belong in the point-fastest layout.
constexpr int kTileStride = 512 + 2;     // doubles between the tiles of neighbouring profiles
template <int W, bool FULL>      // W = 8 or 16 profiles per workgroup, a wavefront each; W = 16 is the one launched
__global__ void __launch_bounds__(64 * W, 4)
tridiag_bt_kernel(const double *__restrict__ A, const double *__restrict__ B, const double *__restrict__ C,
                  const double *D, double *u, int64_t nb, int n, int32_t *__restrict__ status) {
    extern __shared__ double lds[];
    using G = WaveGroup<64>;
    using Tile = WaveTile<8>;
    G g(lds);
    const int lane = threadIdx.x & 63, w = wave_in_block();
    const int64_t b0 = (int64_t)blockIdx.x * W;
    // the element this thread moves in pass r: row 64 r + jrow, profile b0 + p
    const int p = lane & (W - 1), jrow = (64 / W) * w + lane / W;
    const bool pin = FULL || b0 + p < nb;
    // ... and its place in tile p: piece (64 r + jrow) / 2 = 64 (r/2) + 4 a + k with a = 8 (r%2) + jrow/8, k = (jrow/2)%4
    const int k = (jrow >> 1) & 3, xk = Tile::X(k), a0 = jrow >> 3;
    const int place0 = p * kTileStride + 2 * (16 * k + (a0 ^ xk)) + (jrow & 1);
    const int place1 = p * kTileStride + 2 * (16 * k + ((8 + a0) ^ xk)) + (jrow & 1);
    auto place = [&](int r) { return ((r & 1) ? place1 : place0) + 128 * (r >> 1); };
    auto fetch = [&](const double *X, double (&v)[8], double fill) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int j = 64 * r + jrow;
            v[r] = (FULL || (pin && j < n)) ? __builtin_nontemporal_load(&X[(int64_t)j * nb + b0 + p]) : fill;
        }
    };
    auto stage = [&](int buf, const double (&v)[8]) {
#pragma unroll
        for (int r = 0; r < 8; ++r) lds[buf * W * kTileStride + place(r)] = v[r];
    };
    double r0[8], r1[8], r2[8], r3[8];
    fetch(A, r0, 0.0);
    fetch(B, r1, 1.0);
    fetch(C, r2, 0.0);
    fetch(D, r3, 0.0);
    Tile t0(lds + w * kTileStride), t1(lds + (W + w) * kTileStride);
    double a[8], bb[8], c[8], d[8];
    stage(0, r0);
    stage(1, r1);
    __syncthreads();
    t0.take(0, a);
    t1.take(0, bb);
    __syncthreads();
    stage(0, r2);
    stage(1, r3);
    __syncthreads();
    t0.take(0, c);
    t1.take(0, d);
    const int j0 = 8 * lane;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        c[i] = (j0 + i == 0) ? 0.0 : c[i];
        a[i] = (j0 + i == n - 1) ? 0.0 : a[i];
    }
    int sing = 0;
    chunk_solve<8>(g, c, bb, a, d, sing);
    Tile::sync();          // the wavefront's own reads of tile w are done: the tile takes the solution
    t0.put(0, d);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int j = 64 * r + jrow;
        const double x = lds[place(r)];
        if (FULL || (pin && j < n)) __builtin_nontemporal_store(x, &u[(int64_t)j * nb + b0 + p]);
    }
    if (status) {
        sing = g.any2(sing);
        if (sing && lane == 0 && (FULL || b0 + w < nb)) atomicOr(&status[b0 + w], TG_ST_SINGULAR);
    }
}

// ------------------------------------------------------------------------------------------
// Element-wise / stencil kernels on (nb, n) point-fastest arrays: one thread per point.
// ------------------------------------------------------------------------------------------
__global__ void ewma_kernel(const double *__restrict__ y, double *__restrict__ ye, double a, int seeded, int64_t count) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < count; i += stride) {
        const double v = y[i];
        ye[i] = seeded ? a * v + (1.0 - a) * ye[i] : a * v + (1.0 - a) * v;
    }
}

__global__ void ewma_profiles_kernel(const double *__restrict__ y, double *__restrict__ ye, double a, const int32_t *__restrict__ status,
                                     int64_t nb, int n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = nb * n, stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int st = status[i / n];
        if (st & (TG_ST_CONVERGED | TG_ST_INVALID)) continue;
        const double v = y[i];
        ye[i] = a * v + (1.0 - a) * ((st & TG_ST_SEEDED) ? ye[i] : v);
    }
}

__global__ void ddx_kernel(const double *__restrict__ u, double dx, int order, double *__restrict__ out, int64_t nb, int n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = nb * n, stride = (int64_t)gridDim.x * blockDim.x;
    const Recip rc(dx, 1.0);
    for (; i < total; i += stride) {
        const int j = (int)(i % n);
        const double *r = u + (i - j);
        double v;
        if (order == 2 && j == 0) v = (-3.0 / 2.0 * r[0] + 2.0 * r[1] - 1.0 / 2.0 * r[2]) / dx;
        else if (order == 2 && j == n - 1) v = (3.0 / 2.0 * r[n - 1] - 2.0 * r[n - 2] + 1.0 / 2.0 * r[n - 3]) / dx;
        else v = ddx_edge1_point<true>(j > 0 ? r[j - 1] : 0.0, r[j], j < n - 1 ? r[j + 1] : 0.0, j == 0, j == n - 1, rc);
        out[i] = v;
    }
}

__global__ void shestakov_flux_kernel(const double *__restrict__ p, double dx, double *__restrict__ out, int64_t nb, int n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = nb * n, stride = (int64_t)gridDim.x * blockDim.x;
    const Recip rc(dx, 1.0);
    for (; i < total; i += stride) {
        const int j = (int)(i % n);
        const double *r = p + (i - j);
        const double d = ddx_edge1_point<true>(j > 0 ? r[j - 1] : 0.0, r[j], j < n - 1 ? r[j + 1] : 0.0, j == 0, j == n - 1, rc);
        out[i] = -(d * d * d) / (r[j] * r[j]);
    }
}

__global__ void flux_split_kernel(const double *__restrict__ flux, const double *__restrict__ p, double dx, Theta th,
                                  const double *__restrict__ gxx, const double *__restrict__ gradx,
                                  double *__restrict__ D, double *__restrict__ c, double *__restrict__ DHat,
                                  double *__restrict__ cHat, double *__restrict__ theta, int64_t nb, int n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = nb * n, stride = (int64_t)gridDim.x * blockDim.x;
    const Recip rc(dx, 1.0);
    for (; i < total; i += stride) {
        const int j = (int)(i % n);
        const double *r = p + (i - j);
        const double dpdx = ddx_edge1_point<true>(j > 0 ? r[j - 1] : 0.0, r[j], j < n - 1 ? r[j + 1] : 0.0, j == 0, j == n - 1, rc);
        double Dv, cv, DH, cH, thv;
        flux_split_point(flux[i], r[j], dpdx, gxx ? gxx[j] : 1.0, gradx ? gradx[j] : 1.0, th, Dv, cv, DH, cH, thv);
        D[i] = Dv;
        c[i] = cv;
        if (DHat) DHat[i] = DH;
        if (cHat) cHat[i] = cH;
        if (theta) theta[i] = thv;
    }
}

// Centred moving average (tango/smoother.py:13-48): full window w in the interior, the symmetric
// window 2j+1 on the (w-1)/2 points at either end.  Summed directly, left to right.
__global__ void moving_average_kernel(const double *__restrict__ x, int w, double *__restrict__ out, int64_t nb, int n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = nb * n, stride = (int64_t)gridDim.x * blockDim.x;
    const int h = (w - 1) / 2;
    for (; i < total; i += stride) {
        const int j = (int)(i % n);
        const double *r = x + (i - j);
        int hh = h;
        if (j < hh) hh = j;
        if (n - 1 - j < hh) hh = n - 1 - j;
        double s = 0.0;
        for (int k = j - hh; k <= j + hh; ++k) s += r[k];
        out[i] = s / (double)(2 * hh + 1);
    }
}

// Part of a row as a fixed affine map of a stretch of another row: out[b, out_lo + r] = bias[r] + sum_z M[r, z] in[b, in_lo + z]
// (bias may be null).
// The polynomial extrapolation of transport coefficients beyond the trusted part of a turbulence grid is such a map
// (tango/interfacegrids_gene.py:288-292: a least-squares fit to the points of a zone, evaluated on the points beyond it,
// is linear in the data and its matrix depends on the two grids only).  Summed left to right.
__global__ void zone_apply_kernel(const double *__restrict__ in, int64_t nin, int64_t in_lo, int zone, const double *__restrict__ M,
                                  const double *__restrict__ bias, double *__restrict__ out, int64_t nout, int64_t out_lo, int rows,
                                  int nonneg, int64_t nb) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = nb * rows, stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int64_t b = i / rows;
        const int r = (int)(i - b * rows);
        const double *x = in + b * nin + in_lo;
        const double *m = M + (int64_t)r * zone;
        double s = bias ? bias[r] : 0.0;
        for (int z = 0; z < zone; ++z) s += m[z] * x[z];
        out[b * nout + out_lo + r] = (nonneg && s < 0.0) ? 0.0 : s;
    }
}

struct HC {
    tg_hcoeffs h;
};

__device__ __forceinline__ double hval(const double *H, int64_t stride, int64_t b, int j) {
    return H[b * stride + j];
}

__global__ void h_to_matrix_kernel(double dt, const double *__restrict__ dt_b, double dx, const double *__restrict__ UL,
                                   const double *__restrict__ Um1, HC hc, double *__restrict__ A, double *__restrict__ B,
                                   double *__restrict__ C, double *__restrict__ D, double *__restrict__ G, int64_t nb, int n) {
    const tg_hcoeffs &H = hc.h;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = nb * n, stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int j = (int)(i % n);
        const int64_t b = i / n;
        const Recip rc(dx, dt_b ? dt_b[b] : dt);
        double h2m = 0, h3m = 0, bm = 0, h2p = 0, h3p = 0, bp = 0, g4 = 0, h6 = 0, h7 = 0;
        if (j < n - 1) {
            if (H.H2) {
                h2p = (hval(H.H2, H.s2, b, j) + hval(H.H2, H.s2, b, j + 1)) / 2.0;
                if (j > 0) h2m = (hval(H.H2, H.s2, b, j - 1) + hval(H.H2, H.s2, b, j)) / 2.0;
            }
            if (H.H3) {
                h3p = (hval(H.H3, H.s3, b, j) + hval(H.H3, H.s3, b, j + 1)) / 2.0;
                bp = fbeta_point(h2p, h3p, dx);
                if (j > 0) {
                    h3m = (hval(H.H3, H.s3, b, j - 1) + hval(H.H3, H.s3, b, j)) / 2.0;
                    bm = fbeta_point(h2m, h3m, dx);
                }
            }
            if (H.H4) {
                if (j == 0) g4 = -(hval(H.H4, H.s4, b, 0) + hval(H.H4, H.s4, b, 1)) / (2.0 * dx);
                else g4 = -(hval(H.H4, H.s4, b, j + 1) - hval(H.H4, H.s4, b, j - 1)) / (2.0 * dx);
            }
            if (H.H6) h6 = hval(H.H6, H.s6, b, j);
            if (H.H7) h7 = hval(H.H7, H.s7, b, j);
        }
        double a_, b_, c_, d_;
        fd_row<true>(j == n - 1, hval(H.H1, H.s1, b, j), Um1[i], rc, UL[b], h2m, h3m, bm, h2p, h3p, bp, g4, h6, h7, a_, b_, c_, d_);
        A[i] = a_;
        B[i] = b_;
        C[i] = c_;
        D[i] = d_;
        if (G) G[i] = (H.H8 && j < n - 1) ? -hval(H.H8, H.s8, b, j) : 0.0;
    }
}

// Residual: one lane group of 64 lanes per profile (any n), strided over the row.
__global__ void __launch_bounds__(256)
residual_kernel(const double *__restrict__ A, const double *__restrict__ B, const double *__restrict__ C,
                const double *__restrict__ f, const double *__restrict__ u, const double *__restrict__ G,
                const double *__restrict__ w, double *__restrict__ resid, double *__restrict__ sumsq, int64_t nb, int n) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave_in_block();
    if (b >= nb) return;
    const int64_t off = b * (int64_t)n;
    double mx = 0.0;
    for (int j = lane; j < n; j += 64) mx = fmax(mx, fabs(f[off + j]));
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) mx = fmax(mx, __shfl_xor(mx, m, 64));
    double ss = 0.0;
    for (int j = lane; j < n; j += 64) {
        const double up = (j < n - 1) ? u[off + j + 1] : 0.0;
        const double um = (j > 0) ? u[off + j - 1] : 0.0;
        double r = A[off + j] * up + B[off + j] * u[off + j] + C[off + j] * um;
        if (G) r = r + G[off + j] * w[off + j];
        r = r - f[off + j];
        r = r / mx;
        if (resid) resid[off + j] = r;
        ss += r * r;
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) ss += __shfl_xor(ss, m, 64);
    if (sumsq && lane == 0) sumsq[b] = ss;
}

// Streaming probe with the fused step's memory shape: five input streams, four output streams,
// 16 bytes per lane, fully coalesced, no dependent arithmetic.  Its rate is the achievable ceiling
// for the step's traffic on the device at hand.
__global__ void __launch_bounds__(256)
stream_probe_kernel(const double2 *__restrict__ a, const double2 *__restrict__ b, const double2 *__restrict__ c,
                       const double2 *__restrict__ d, const double2 *__restrict__ e, double2 *__restrict__ w,
                       double2 *__restrict__ x, double2 *__restrict__ y, double2 *__restrict__ z, int64_t n2) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n2; i += stride) {
        double2 va, vb, vc, vd, ve;
        va.x = __builtin_nontemporal_load(&a[i].x); va.y = __builtin_nontemporal_load(&a[i].y);
        vb.x = __builtin_nontemporal_load(&b[i].x); vb.y = __builtin_nontemporal_load(&b[i].y);
        vc.x = __builtin_nontemporal_load(&c[i].x); vc.y = __builtin_nontemporal_load(&c[i].y);
        vd.x = __builtin_nontemporal_load(&d[i].x); vd.y = __builtin_nontemporal_load(&d[i].y);
        ve.x = __builtin_nontemporal_load(&e[i].x); ve.y = __builtin_nontemporal_load(&e[i].y);
        __builtin_nontemporal_store(va.x + vb.x, &w[i].x); __builtin_nontemporal_store(va.y + vb.y, &w[i].y);
        __builtin_nontemporal_store(vb.x + vc.x, &x[i].x); __builtin_nontemporal_store(vb.y + vc.y, &x[i].y);
        __builtin_nontemporal_store(vc.x + vd.x, &y[i].x); __builtin_nontemporal_store(vc.y + vd.y, &y[i].y);
        __builtin_nontemporal_store(vd.x + ve.x, &z[i].x); __builtin_nontemporal_store(vd.y + ve.y, &z[i].y);
    }
}


// ------------------------------------------------------------------------------------------
// Cubic interpolating spline with not-a-knot ends, the curve scipy.interpolate.InterpolatedUnivariateSpline
// (FITPACK, k = 3) passes through the same points, in the second-derivative ("moment") form: the interior moments
// M_1..M_{n-2} solve a tridiagonal system (the end moments follow from the continuity of the third derivative at
// x_1 and x_{n-2} and are eliminated from rows 1 and n-2).  Used to move profiles and transport coefficients
// between the transport grid and a turbulence grid (tango/interfacegrids_gene.py:51-57, :446-484).
// ------------------------------------------------------------------------------------------
__global__ void spline_rows_kernel(const double *__restrict__ x, const double *__restrict__ y, int64_t nb, int n,
                                   double *__restrict__ A, double *__restrict__ B, double *__restrict__ C,
                                   double *__restrict__ D) {
    const int m = n - 2;
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; k < nb * m; k += stride) {
        const int64_t b = k / m;
        const int i = (int)(k - b * m) + 1;          // interior point 1 .. n-2
        const double hm = x[i] - x[i - 1], hp = x[i + 1] - x[i];
        const double *yr = y + b * n;
        const double rhs = 6.0 * ((yr[i + 1] - yr[i]) / hp - (yr[i] - yr[i - 1]) / hm);
        double c = hm, d = 2.0 * (hm + hp), a = hp;
        if (i == 1) {              // M_0 = ((h_0 + h_1) M_1 - h_0 M_2) / h_1
            d += hm * (hm + hp) / hp;
            a -= hm * hm / hp;
            c = 0.0;
        }
        if (i == n - 2) {          // M_{n-1} = ((h_{n-2} + h_{n-3}) M_{n-2} - h_{n-2} M_{n-3}) / h_{n-3}
            d += hp * (hp + hm) / hm;
            c -= hp * hp / hm;
            a = 0.0;
        }
        A[k] = a;
        B[k] = d;
        C[k] = c;
        D[k] = rhs;
    }
}

__global__ void spline_eval_kernel(const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ Mi,
                                   int64_t nb, int n, const double *__restrict__ xo, int no, int lo, int hi, int nonneg,
                                   double *__restrict__ out) {
    const int m = n - 2;
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; k < nb * no; k += stride) {
        const int64_t b = k / no;
        const int j = (int)(k - b * no);
        double v = 0.0;
        if (j >= lo && j < hi) {
            const double xv = xo[j];
            // interval i with x_i <= xv < x_{i+1}, end intervals beyond the ends (the end cubics continue)
            int a = 0, z = n - 1;
            while (z - a > 1) {
                const int mid = (a + z) >> 1;
                if (x[mid] <= xv) a = mid;
                else z = mid;
            }
            const int i = a;
            const double *Mr = Mi + b * m;
            const double *yr = y + b * n;
            auto moment = [&](int q) -> double {
                if (q == 0) {
                    const double h0 = x[1] - x[0], h1 = x[2] - x[1];
                    return ((h0 + h1) * Mr[0] - h0 * Mr[1]) / h1;
                }
                if (q == n - 1) {
                    const double hz = x[n - 1] - x[n - 2], hy = x[n - 2] - x[n - 3];
                    return ((hz + hy) * Mr[m - 1] - hz * Mr[m - 2]) / hy;
                }
                return Mr[q - 1];
            };
            const double h = x[i + 1] - x[i];
            const double M0 = moment(i), M1 = moment(i + 1);
            const double dl = xv - x[i], dr = x[i + 1] - xv;
            v = (M0 * dr * dr * dr + M1 * dl * dl * dl) / (6.0 * h) + (yr[i] / h - M0 * h / 6.0) * dr + (yr[i + 1] / h - M1 * h / 6.0) * dl;
            if (nonneg && v < 0.0) v = 0.0;
        }
        out[k] = v;
    }
}

// Tukey window (tango/extras/bufferzone.py:51 -> scipy.signal.tukey): cosine lobes over the first and last
// floor(alpha (n-1) / 2) + 1 points, one in between.
__global__ void tukey_window_kernel(double *__restrict__ w, int n, double alpha) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    double v = 1.0;
    if (n > 1 && alpha > 0.0) {
        const double a = alpha < 1.0 ? alpha : 1.0;
        const int width = (int)floor(a * (n - 1) / 2.0);
        if (k <= width) v = 0.5 * (1.0 + cos(M_PI * (-1.0 + 2.0 * k / a / (n - 1))));
        if (k >= n - width - 1) v = 0.5 * (1.0 + cos(M_PI * (-2.0 / a + 1.0 + 2.0 * k / a / (n - 1))));
    }
    w[k] = v;
}

// Philox 4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3"): a keyed bijection of a 128-bit counter.
struct Philox {
    uint32_t k0, k1;
    __device__ Philox(uint64_t seed) : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)) {}
    __device__ void operator()(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t (&out)[4]) const {
        uint32_t a = k0, b = k1;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
            const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ a, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ b, n3 = (uint32_t)p0;
            c0 = n0; c1 = n1; c2 = n2; c3 = n3;
            a += 0x9E3779B9u;
            b += 0xBB67AE85u;
        }
        out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
    }
};

// two independent standard normal numbers for (sequence, profile, pair index): 53-bit uniforms, Box-Muller
__device__ __forceinline__ void normal_pair(const Philox &rng, uint64_t sequence, uint64_t profile, uint32_t pair, double &z0, double &z1) {
    uint32_t r[4];
    rng((uint32_t)sequence ^ (uint32_t)(profile >> 32) * 0x85EBCA6Bu, (uint32_t)(sequence >> 32), (uint32_t)profile, pair, r);
    const double u0 = ((double)(((uint64_t)r[0] << 21) ^ (r[1] >> 11)) + 0.5) * (1.0 / 9007199254740992.0);     // (0, 1)
    const double u1 = ((double)(((uint64_t)r[2] << 21) ^ (r[3] >> 11)) + 0.5) * (1.0 / 9007199254740992.0);
    const double rad = sqrt(-2.0 * log(u0));
    z0 = rad * cos(2.0 * M_PI * u1);
    z1 = rad * sin(2.0 * M_PI * u1);
}

// AR(1) flux noise: one wavefront per profile, 512 points per pass (8 per lane).  Every lane runs the recursion
// y_j = lambda y_{j-1} + a_j over its own points from zero; what the points before it contribute follows from a scan
// over the lanes of the affine maps c -> lambda^8 c + (lane's last value), and the pass hands its last value on.
__global__ void __launch_bounds__(256)
ar1_noise_kernel(uint64_t seed, uint64_t sequence, double amplitude, double lambda, int damp, const double *__restrict__ taper,
                 double *__restrict__ out, int64_t nb, int n) {
    const int lane = threadIdx.x & 63;
    const int64_t b = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave_in_block();
    if (b >= nb) return;
    const Philox rng(seed);
    const double sd = sqrt(1.0 - lambda * lambda);
    double lp[9];                       // lambda^0 .. lambda^8
    lp[0] = 1.0;
#pragma unroll
    for (int i = 1; i <= 8; ++i) lp[i] = lp[i - 1] * lambda;
    // stationary start: the value "before the first point" has unit variance
    double carry, spare;
    normal_pair(rng, sequence, (uint64_t)b, 0xFFFFFFFFu, carry, spare);
    for (int base = 0; base < n; base += 512) {
        const int j0 = base + lane * 8;
        double y[8];
        double acc = 0.0;
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
            double z0, z1;
            normal_pair(rng, sequence, (uint64_t)b, (uint32_t)((j0 + i) >> 1), z0, z1);
            acc = lambda * acc + sd * z0;
            y[i] = acc;
            acc = lambda * acc + sd * z1;
            y[i + 1] = acc;
        }
        // inclusive scan over lanes of (A, B): value after this lane's points = A * (value before lane 0) + B
        double A = lp[8], B = acc;
#pragma unroll
        for (int h = 1; h < 64; h <<= 1) {
            const double Al = __shfl_up(A, (unsigned)h, 64), Bl = __shfl_up(B, (unsigned)h, 64);
            if (lane >= h) {
                B = A * Bl + B;
                A = A * Al;
            }
        }
        // value before this lane's first point
        const double Ap = __shfl_up(A, 1u, 64), Bp = __shfl_up(B, 1u, 64);
        const double before = (lane == 0) ? carry : Ap * carry + Bp;
        const double lastA = __shfl(A, 63, 64), lastB = __shfl(B, 63, 64);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int j = j0 + i;
            if (j < n) {
                const double v = y[i] + lp[i + 1] * before;
                // dampen_sides (tango/extras/noisyflux.py:141-162): np.linspace(0, 1, damp) over the first and last damp points
                double d = 1.0;
                if (damp > 1) {
                    if (j < damp) d = (double)j / (double)(damp - 1);
                    if (j >= n - damp) d = (double)(n - 1 - j) / (double)(damp - 1);
                } else if (damp == 1) {
                    if (j == 0) d = 0.0;
                    if (j == n - 1) d = 1.0;
                }
                const double m = 1.0 + d * (amplitude * v);
                out[b * (int64_t)n + j] = taper ? m * taper[j] : m;
            }
        }
        carry = lastA * carry + lastB;
    }
}

__global__ void status_or_kernel(int32_t *__restrict__ status, int64_t nb, int bits) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nb) status[i] |= bits;
}

__global__ void status_reduce_kernel(const int32_t *__restrict__ status, int64_t nb, unsigned long long *counts) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned long long act = 0, conv = 0, inv = 0, sing = 0;
    for (; i < nb; i += stride) {
        const int s = status[i];
        conv += (s & TG_ST_CONVERGED) != 0;
        inv += (s & TG_ST_INVALID) != 0;
        sing += (s & TG_ST_SINGULAR) != 0;
        act += (s & (TG_ST_CONVERGED | TG_ST_INVALID)) == 0;
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        act += __shfl_xor(act, m, 64);
        conv += __shfl_xor(conv, m, 64);
        inv += __shfl_xor(inv, m, 64);
        sing += __shfl_xor(sing, m, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        if (act) atomicAdd(&counts[0], act);
        if (conv) atomicAdd(&counts[1], conv);
        if (inv) atomicAdd(&counts[2], inv);
        if (sing) atomicAdd(&counts[3], sing);
    }
}

// Residual summary of an ensemble (tango/solver.py:464-475 taken over every profile that still iterates): from each
// profile's latest rms residual, out[0] = sum over those profiles of sum_j r_j^2 (= n rms^2), out[1] = the number of
// residual entries behind it (n per profile), out[2] = profiles still iterating, out[3] = the largest latest rms of
// any profile, out[4] = 1 if any profile is invalid.  out[0..2] add up over shards, out[3..4] take the maximum.
__global__ void residual_reduce_kernel(const int32_t *__restrict__ status, const int32_t *__restrict__ iters,
                                       const double *__restrict__ err, int64_t err_stride, int64_t nb, int n, double *out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double ss = 0.0, cnt = 0.0, act = 0.0, mx = 0.0, inv = 0.0;
    for (; i < nb; i += stride) {
        const int s = status[i];
        const int it = iters[i];
        const bool has = it > 0 && it <= err_stride;
        const double rms = has ? err[i * err_stride + it - 1] : 0.0;
        const bool active = (s & (TG_ST_CONVERGED | TG_ST_INVALID)) == 0;
        if (active && has) {
            ss += rms * rms * (double)n;
            cnt += (double)n;
        }
        act += active ? 1.0 : 0.0;
        mx = fmax(mx, rms);          // fmax drops a NaN residual; such a profile is reported through `invalid`
        inv = fmax(inv, (s & TG_ST_INVALID) ? 1.0 : 0.0);
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        ss += __shfl_xor(ss, m, 64);
        cnt += __shfl_xor(cnt, m, 64);
        act += __shfl_xor(act, m, 64);
        mx = fmax(mx, __shfl_xor(mx, m, 64));
        inv = fmax(inv, __shfl_xor(inv, m, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        if (cnt != 0.0) {
            atomicAdd(&out[0], ss);
            atomicAdd(&out[1], cnt);
        }
        if (act != 0.0) atomicAdd(&out[2], act);
        // non-negative doubles order like their bit patterns
        if (mx > 0.0) atomicMax(reinterpret_cast<unsigned long long *>(&out[3]), (unsigned long long)__double_as_longlong(mx));
        if (inv > 0.0) atomicMax(reinterpret_cast<unsigned long long *>(&out[4]), (unsigned long long)__double_as_longlong(inv));
    }
}

// Centred moving average of the flux along a profile inside the fused step (tango/smoother.py:13-48 at its call site
// tango/lodestro_method.py:359-362): full window 2h+1 in the interior, the symmetric window 2j+1 on the h points at
// either end, each window summed left to right and divided by its length, as moving_average_kernel does.  The P
// values per lane go through a row in LDS (`row`: the region of this lane group's profile); element e sits in slot
// e + e/8, one double of padding per lane chunk, so that lanes reading the same window position (a stride of 8
// elements) fall on different banks.
template <int P, class G>
__device__ __forceinline__ void smooth_flux(double *row, int n, int j0, int h, double (&fl)[P]) {
    auto barrier = [] {
        if constexpr (G::kBlock) {
            __syncthreads();
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    };
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int e = j0 + i;
        row[e + (e >> 3)] = fl[i];
    }
    barrier();
    // The lane's P windows advance together, one window position per pass: P independent LDS reads and P additions per
    // pass instead of a chain of dependent reads per point.  A point whose (shrunken) window does not reach a position
    // adds an exact zero there, so every window is still summed left to right over its own elements.
    int hh[P];
    double sum[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int j = j0 + i;
        int w = h;
        w = j < w ? j : w;
        w = (n - 1 - j) < w ? (n - 1 - j) : w;
        hh[i] = (j < n) ? w : -1;
        sum[i] = 0.0;
    }
    for (int d = -h; d <= h; ++d) {
        const int ad = d < 0 ? -d : d;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            int k = j0 + i + d;
            k = k < 0 ? 0 : (k > n - 1 ? n - 1 : k);
            const double v = row[k + (k >> 3)];
            sum[i] += (ad <= hh[i]) ? v : 0.0;
        }
    }
    // the quotient by the window's length: the short division sequence gives the correctly rounded quotient for these
    // operands, i.e. what the division itself gives (tg_comm.h)
#pragma unroll
    for (int i = 0; i < P; ++i) {
        if (hh[i] >= 0) fl[i] = quot<kDivRounded>(sum[i], (double)(2 * hh[i] + 1));
    }
    barrier();
}

// ------------------------------------------------------------------------------------------
// Fused Picard iteration, one field.
// ------------------------------------------------------------------------------------------
// One launch advances every active profile by up to Q.iters_per_launch iterations.  The profile stays in
// registers between those iterations (u, the three EWMA memories and u_{m-1}: 42 doubles per lane), so the
// state crosses HBM once per launch instead of once per iteration; the convergence test, the freeze of a
// converged profile and the residual history are per iteration, exactly as with one iteration per launch.
// GEN = 0 is the Cartesian case with the built-in flux (no geometry arrays, no external flux): H2 and H3 are then the
// EWMA memories themselves and need no registers of their own.  GEN = 1 adds what a run coupled to a turbulence code
// needs (external flux array, geometry arrays, H1/H7 arrays, flux smoothing, flux multiplier); GEN = 2 adds pseudo-timestep
// control and the power-law flux, whose loop-carried values and long code would otherwise cost the coupled run its
// registers.
// MULTI = false compiles the single-iteration form (no loop, no selects on `done`).

template <int P, class G, bool FULL, int GEN, bool MULTI>
__global__ void __launch_bounds__(G::kBlock ? G::T : 256, (G::kBlock && G::T > 512) ? 4 : 2)   // two wavefronts per SIMD: <= 256 registers
picard_1f_kernel(tg_state_1f S_, tg_params_1f Q) {
    constexpr bool GENERAL = GEN != 0;
    constexpr bool EXTRAS = GEN == 2;      // pseudo-timestep control, power-law flux
    extern __shared__ double lds[];
    G g(lds);
    const int64_t b = (int64_t)blockIdx.x * G::profiles_per_block(blockDim.x) + g.profile_in_block();
    // Rows move coalesced and are transposed through the wavefront's LDS tiles (tg_wave_io.h) when the wavefront
    // owns 512 consecutive doubles of every array: always with one profile per wavefront (COOP_ALWAYS), and with
    // several short profiles per wavefront as long as all of them exist (not the tail of the ensemble).  Profiles
    // of such a wavefront that need no iteration stay in it as `done` from the start and take part in the I/O.
    constexpr bool COOP_T = kWaveCoop<P, G, FULL>;
    constexpr bool COOP_ALWAYS = COOP_T && (G::kBlock || G::T == 64);
    const bool exists = b < S_.nb;
    const bool coop = COOP_ALWAYS || (COOP_T && __all(exists));
    if (!exists) return;
    int st = S_.status[b];
    const bool active = Q.fixed_count || !(st & (TG_ST_CONVERGED | TG_ST_INVALID));
    if ((COOP_T && !COOP_ALWAYS && coop) ? !__any(active) : !active) return;
    const int n = (int)S_.n;
    const int t = g.t();
    const int j0 = t * P;
    const int64_t off = b * S_.n;
    const double dt_full = Q.dt_b ? Q.dt_b[b] : Q.dt;
    Recip rc(Q.dx, dt_full);
    const double UL = Q.UL[b];
    const bool lane_first = (t == 0);
    // grids beyond one wavefront (n > 512) have condition numbers above 1e9: keep the reference's divisions
    constexpr bool EXACT = G::kBlock;
    constexpr int DIV = EXACT ? kDivRounded : kDivFast;      // pivots: correctly rounded reciprocals on those grids
    const bool external = GENERAL && (Q.flux_model == TG_FLUX_EXTERNAL);
    const bool geom = GENERAL && (Q.Vprime != nullptr);
    const int iters = MULTI ? (Q.iters_per_launch > 1 ? Q.iters_per_launch : 1) : 1;
    const double S0 = Q.H7 ? 0.0 : Q.S0[b];
    const int src_cut = Q.H7 ? 0 : Q.src_cut[b];
    const Theta th(Q.Dmin, Q.Dmax, Q.dpdx_thr);
    const double omp = 1.0 - Q.a_prof, omt = 1.0 - Q.a_turb;
    int it0 = S_.iters[b];

    // the whole state of the profile, loaded once
    double u[P + 2], ue[P], um1[P], DE[P], cE[P];
    double fl[GENERAL ? P : 1];
    // first element of this wavefront's 512 doubles: its first profile, or its segment of the workgroup's profile
    const int64_t woff = G::kBlock ? off + (int64_t)wave_in_block() * 512 : off - (int64_t)((threadIdx.x & 63) / G::T) * S_.n;
    using Tile = WaveTile<8>;
    constexpr int NT = kCoopTiles<G>;
    Tile wt(lds + (COOP_T ? (G::kBlock ? 8 * G::T : 0) + wave_in_block() * NT * Tile::kDoubles : 0));
    // The old EWMA memory of the profile has weight 1 - a_prof: with a_prof = 1 (and whenever the turbulence step is
    // skipped) it is not read at all; zeros stand in, so that a_prof u + 0 * memory is u exactly.
    const bool need_ue = !Q.frozen_turb && (external || Q.a_prof != 1.0);
    auto load_state_coop = [&]() {
        double2 r0[4], r1[4], r2[4], r3[4], r4[4];
        wt.fetch(S_.u + woff, r0);
        if (need_ue) wt.fetch(S_.uE + woff, r1);
        wt.fetch(S_.DE + woff, r2);
        wt.fetch(S_.cE + woff, r3);
        wt.fetch(S_.um1 + woff, r4);
        double2 r5[GENERAL ? 4 : 1];
        if constexpr (GENERAL) {
            if (external) wt.fetch(Q.flux + woff, r5);
        }
        if (!need_ue) {
#pragma unroll
            for (int i = 0; i < P; ++i) ue[i] = 0.0;
        }
        if constexpr (NT >= 4) {
            wt.stage(0, r0);
            if (need_ue) wt.stage(1, r1);
            wt.stage(2, r2);
            wt.stage(3, r3);
            Tile::sync();
            wt.take(0, u + 1);
            if (need_ue) wt.take(1, ue);
            wt.take(2, DE);
            wt.take(3, cE);
            Tile::sync();
        } else {
            wt.stage(0, r0);
            if (need_ue) wt.stage(1, r1);
            Tile::sync();
            wt.take(0, u + 1);
            if (need_ue) wt.take(1, ue);
            Tile::sync();
            wt.stage(0, r2);
            wt.stage(1, r3);
            Tile::sync();
            wt.take(0, DE);
            wt.take(1, cE);
            Tile::sync();
        }
        wt.stage(0, r4);
        Tile::sync();
        wt.take(0, um1);
        Tile::sync();
        if constexpr (GENERAL) {
            if (external) {
                wt.stage(0, r5);
                Tile::sync();
                wt.take(0, fl);
                Tile::sync();
            }
        }
    };
    auto load_state_lane = [&]() {
        load_chunk<P, FULL>(S_.u + off, n, j0, u + 1, 0.0);
        if (need_ue) {
            load_chunk<P, FULL>(S_.uE + off, n, j0, ue, 1.0);
        } else {
#pragma unroll
            for (int i = 0; i < P; ++i) ue[i] = 0.0;
        }
        load_chunk<P, FULL>(S_.DE + off, n, j0, DE, 0.0);
        load_chunk<P, FULL>(S_.cE + off, n, j0, cE, 0.0);
        load_chunk<P, FULL>(S_.um1 + off, n, j0, um1, 0.0);
        if constexpr (GENERAL) {
            if (external) load_chunk<P, FULL>(Q.flux + off, n, j0, fl, 0.0);
        }
    };
    // P points per lane of one or two arrays that all profiles share (n doubles each; a1 may be null).  Plain
    // per-lane loads: these arrays stay in the caches, and a detour through the LDS tiles measured slower.
    auto load_shared = [&](const double *a0, double *d0, const double *a1, double *d1) {
        load_chunk<P, FULL>(a0, n, j0, d0, 1.0);
        if (a1) load_chunk<P, FULL>(a1, n, j0, d1, 1.0);
    };
    if constexpr (COOP_ALWAYS) {
        load_state_coop();
    } else if constexpr (COOP_T) {
        if (coop) load_state_coop();
        else load_state_lane();
    } else {
        load_state_lane();
    }

    // The new profile, and the three memories when the turbulence step ran, go back the way they came.
    auto store_coop = [&](auto pred, bool prof, bool mem) {
        const uint64_t amask = decltype(pred)::value ? __ballot(active) : ~0ull;   // lanes whose profile was stepped
        constexpr bool PR = decltype(pred)::value;
        if constexpr (NT >= 4) {
            if (prof) wt.put(0, u + 1);
            if (mem) {
                wt.put(1, ue);
                wt.put(2, DE);
                wt.put(3, cE);
            }
            Tile::sync();
            if (prof) wt.template flush<PR>(0, S_.u + woff, amask);
            if (mem) {
                if (!external) wt.template flush<PR>(1, S_.uE + woff, amask);
                wt.template flush<PR>(2, S_.DE + woff, amask);
                wt.template flush<PR>(3, S_.cE + woff, amask);
            }
            Tile::sync();
        } else {
            if (mem) {
                wt.put(0, DE);
                wt.put(1, cE);
                Tile::sync();
                wt.template flush<PR>(0, S_.DE + woff, amask);
                wt.template flush<PR>(1, S_.cE + woff, amask);
                Tile::sync();
            }
            if (prof) wt.put(0, u + 1);
            if (mem) wt.put(1, ue);
            Tile::sync();
            if (prof) wt.template flush<PR>(0, S_.u + woff, amask);
            if (mem && !external) wt.template flush<PR>(1, S_.uE + woff, amask);
            Tile::sync();
        }
    };
    auto store_lane = [&](bool prof, bool mem) {
        if (prof) store_chunk<P, FULL>(S_.u + off, n, j0, u + 1);
        if (mem) {
            if (!external) store_chunk<P, FULL>(S_.uE + off, n, j0, ue);
            store_chunk<P, FULL>(S_.DE + off, n, j0, DE);
            store_chunk<P, FULL>(S_.cE + off, n, j0, cE);
        }
    };
    auto store_state = [&](bool prof, bool mem) {
        if constexpr (COOP_ALWAYS) {
            store_coop(std::false_type{}, prof, mem);
        } else if constexpr (COOP_T) {
            if (coop) store_coop(std::true_type{}, prof, mem);
            else store_lane(prof, mem);
        } else {
            store_lane(prof, mem);
        }
    };
    // a single-iteration launch stores the memories as soon as they are final, which frees their registers for the
    // assembly and the solve (one wavefront per profile: 203 -> 149 registers; 1024- and 2048-point profiles then run
    // three wavefronts per SIMD, 0.42 -> 0.38 / 0.39 ms).  Wavefronts shared by several short profiles keep the late
    // stores of the plain form: their predicated stores measured 4 % slower when issued early.
    constexpr bool EARLY = !MULTI && (GENERAL || G::kBlock || G::T == 64);

    // this profile needs no further iteration (uniform over its lane group)
    constexpr bool PACKED = COOP_T && !COOP_ALWAYS;    // wavefronts that may carry profiles that were never active
    bool done = PACKED ? !active : false;
    const double inv_n = 1.0 / (double)n;      // (a division: formed once, not in every iteration of the launch)
    // the box source of this lane's rows is the same in every iteration of the launch: formed once and kept (`landed` keeps
    // the compiler from forming it again in the loop)
    constexpr bool SRC_KEPT = MULTI && !GENERAL && !G::kBlock && G::T == 64;
    double src_row[SRC_KEPT ? P : 1];
    if constexpr (SRC_KEPT) {
#pragma unroll
        for (int i = 0; i < P; ++i) {
            src_row[i] = (j0 + i < src_cut) ? S0 : 0.0;
            landed(src_row[i]);
        }
    }
    // A profile's first iterate seeds the EWMA memory of the profile (tango/lodestro_method.py:706-708): where a wavefront
    // owns the profile that is a copy made here, once, instead of a select in every iteration of the launch
    constexpr bool PRESEED = !G::kBlock && G::T == 64;
    if constexpr (PRESEED) {
        if (!(st & TG_ST_SEEDED) && !Q.frozen_turb && !external) {
#pragma unroll
            for (int i = 0; i < P; ++i) ue[i] = u[i + 1];
        }
    }
    double face0 = lane_first ? 0.0 : 0.5;
    landed(face0);
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if constexpr (!MULTI && !PACKED) done = false;
        // A group that is done keeps executing (other groups of the wavefront may not be) but every update of
        // its state below is a select on `done`, so nothing of it changes any more.
        const bool seeded = (st & TG_ST_SEEDED) != 0;
        // Only a wavefront shared by several profiles needs the selects: elsewhere `done` and `seeded` are uniform
        // over the wavefront (workgroup), the loop is left as soon as `done` is set, and `seeded` is a branch.
        constexpr bool kShared = !G::kBlock && G::T < 64;
        const bool hold = kShared && done;
        if constexpr (EXTRAS) {
            if (Q.pseudo_time) {
                // the previous iterate takes the u_{m-1} slot (tango/solver.py:198-201); treadInitially's short
                // pseudo-timestep for a profile's first iterations (:210-215)
#pragma unroll
                for (int i = 0; i < P; ++i) um1[i] = hold ? um1[i] : u[i + 1];
                rc = Recip(Q.dx, (it0 < Q.tread_initial_steps) ? Q.tread_initial_dtau : dt_full);
            }
        }
        if (!Q.frozen_turb) {
            // EWMA of the profile (tango/lodestro_method.py:345); with an external flux the caller has advanced it
            if (!external) {
                if constexpr (kShared || G::kBlock) {
#pragma unroll
                    for (int i = 0; i < P; ++i) {
                        const double v = Q.a_prof * u[i + 1] + omp * (seeded ? ue[i] : u[i + 1]);
                        ue[i] = hold ? ue[i] : v;
                    }
                } else if (PRESEED || seeded) {
#pragma unroll
                    for (int i = 0; i < P; ++i) ue[i] = Q.a_prof * u[i + 1] + omp * ue[i];
                } else {     // the first iterate seeds the memory (tango/lodestro_method.py:706-708)
#pragma unroll
                    for (int i = 0; i < P; ++i) ue[i] = Q.a_prof * u[i + 1] + omp * u[i + 1];
                }
            }
            constexpr bool INPLACE = G::kBlock || GENERAL;    // short of registers: update the memories point by point
            double Dn[INPLACE ? 1 : P], cn[INPLACE ? 1 : P];
            // <g^xx> and <|grad x|> on the turbulence grid (tango/lodestro_method.py:352-353), one chunk per lane
            double gxt[GENERAL ? P : 1], grt[GENERAL ? P : 1];
            if constexpr (GENERAL) {
                if (Q.gxx_turb) load_shared(Q.gxx_turb, gxt, Q.gradx_turb, grt);
                else if (Q.gradx_turb) load_shared(Q.gradx_turb, grt, nullptr, nullptr);
            }
            // (a workgroup exchanges through LDS: one exchange of both values costs one pair of barriers instead of two)
            double wl, wr;
            if constexpr (G::kBlock) {
                const double v[2] = {ue[P - 1], ue[0]};
                double lq[2], rq[2];
                g.template exchange<2>(v, 1, lq, rq);
                wl = lq[0];
                wr = rq[1];
            } else {
                wl = g.from_left(ue[P - 1], 1, 0.0);
                wr = g.from_right(ue[0], 1, 0.0);
            }
            // Flux smoothing (tango/lodestro_method.py:359-362): the flux of the whole profile has to exist before any point
            // is split, so the built-in model is evaluated into fl first; from there on it is treated like an outside flux.
            bool have_flux = external;
            if constexpr (GENERAL) {
                const bool power = EXTRAS && Q.flux_model == TG_FLUX_POWER;
                if (Q.smooth_window > 1 || Q.flux_mult || power) {
                    if (!external) {
#pragma unroll
                        for (int i = 0; i < P; ++i) {
                            const int j = j0 + i;
                            const bool first = (i == 0) && lane_first;
                            const bool last = FULL ? (i == P - 1 && t == G::T - 1) : (j == n - 1);
                            const double pm = (i == 0) ? wl : ue[i > 0 ? i - 1 : 0];
                            const double pp = (i == P - 1) ? wr : ue[i < P - 1 ? i + 1 : 0];
                            const double dpdx = ddx_edge1_point<EXACT>(pm, ue[i], pp, first, last, rc);
                            // examples/stripped_down_tango_example.py:60-64: Deff = |(dn/dx / n)^q|, Gamma = -Deff dn/dx
                            if constexpr (EXTRAS) fl[i] = power ? -(pow(fabs(dpdx / ue[i]), Q.flux_power) * dpdx) : quot<DIV>(-(dpdx * dpdx * dpdx), ue[i] * ue[i]);
                            else fl[i] = quot<DIV>(-(dpdx * dpdx * dpdx), ue[i] * ue[i]);
                        }
                    }
                    if (Q.flux_mult) {      // flux-model decorators (taper, noise) act on the flux the model returns
#pragma unroll
                        for (int i = 0; i < P; ++i) {
                            const int j = j0 + i;
                            const int jc = (FULL || j < n) ? j : n - 1;
                            fl[i] = fl[i] * Q.flux_mult[b * Q.flux_mult_stride + jc];
                        }
                    }
                    if (Q.smooth_window > 1) {
                        // the padded row of this profile: in the wavefront's own tiles where it has some (they are idle while
                        // the flux is smoothed, and a row area of its own would cost a resident workgroup per CU), else in the
                        // area the launcher reserved
                        double *row;
                        if constexpr (COOP_T && !G::kBlock) {
                            row = lds + wave_in_block() * NT * Tile::kDoubles + ((threadIdx.x & 63) / G::T) * (G::T * 9);
                        } else {
                            row = lds + Q.reserved_;
                            if constexpr (!G::kBlock) row += wave_in_block() * 576 + ((threadIdx.x & 63) / G::T) * (G::T * 9);
                        }
                        smooth_flux<P, G>(row, n, j0, (Q.smooth_window - 1) / 2, fl);
                    }
                    have_flux = true;
                }
            }
#pragma unroll
            for (int i = 0; i < P; ++i) {
                const int j = j0 + i;
                const bool first = (i == 0) && lane_first;
                const bool last = FULL ? (i == P - 1 && t == G::T - 1) : (j == n - 1);
                const double pm = (i == 0) ? wl : ue[i > 0 ? i - 1 : 0];
                const double pp = (i == P - 1) ? wr : ue[i < P - 1 ? i + 1 : 0];
                const double p = ue[i];
                const double dpdx = ddx_edge1_point<EXACT>(pm, p, pp, first, last, rc);
                double D, c;
                if constexpr (GENERAL) {
                    // flux model (tango/extras/shestakov_nonlinear_diffusion.py:59-60)
                    double flux = quot<DIV>(-(dpdx * dpdx * dpdx), p * p);
                    double gx = 1.0, gr = 1.0;
                    if constexpr (GENERAL) {
                        flux = have_flux ? fl[i] : flux;
                        gx = Q.gxx_turb ? gxt[i] : 1.0;
                        gr = Q.gradx_turb ? grt[i] : 1.0;
                    }
                    double DH, cH, thv;
                    flux_split_point<DIV>(flux, p, dpdx, gx, gr, th, D, c, DH, cH, thv);
                } else {
                    // built-in flux, Cartesian: one quotient.  Also on the grids of more than 512 points: DHat and cHat are
                    // within a few ulp of the three-division values and nothing downstream amplifies that (the pivots and
                    // the rows, where those grids' condition numbers bite, keep their correctly rounded quotients)
                    shestakov_split_point(p, dpdx, th, D, c);
                }
                if constexpr (INPLACE) {
                    const double nD = Q.a_turb * D + omt * (seeded ? DE[i] : D);
                    const double nc = Q.a_turb * c + omt * (seeded ? cE[i] : c);
                    DE[i] = hold ? DE[i] : nD;
                    cE[i] = hold ? cE[i] : nc;
                } else {
                    Dn[i] = D;
                    cn[i] = c;
                }
            }
            // EWMA of D and c (tango/lodestro_method.py:368-369)
            if constexpr (INPLACE) {
            } else if constexpr (kShared) {
#pragma unroll
                for (int i = 0; i < P; ++i) {
                    const double nD = Q.a_turb * Dn[i] + omt * (seeded ? DE[i] : Dn[i]);
                    const double nc = Q.a_turb * cn[i] + omt * (seeded ? cE[i] : cn[i]);
                    DE[i] = hold ? DE[i] : nD;
                    cE[i] = hold ? cE[i] : nc;
                }
            } else if (seeded) {
#pragma unroll
                for (int i = 0; i < P; ++i) {
                    DE[i] = Q.a_turb * Dn[i] + omt * DE[i];
                    cE[i] = Q.a_turb * cn[i] + omt * cE[i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < P; ++i) {
                    DE[i] = Q.a_turb * Dn[i] + omt * Dn[i];
                    cE[i] = Q.a_turb * cn[i] + omt * cn[i];
                }
            }
        }

        // H2 = V' D <g^xx>, H3 = -V' c <|grad x|> (tango/lodestro_method.py:400-405); without geometry arrays they
        // are the EWMA memories themselves
        double H2v[GENERAL ? P : 1], H3v[GENERAL ? P : 1];
        if constexpr (GENERAL) {
            if (geom) {
                double vp[P], gq[P];
                load_shared(Q.Vprime, vp, Q.gxx, gq);
                if (Q.gxx) {
#pragma unroll
                    for (int i = 0; i < P; ++i) H2v[i] = vp[i] * DE[i] * gq[i];
                } else {
#pragma unroll
                    for (int i = 0; i < P; ++i) H2v[i] = vp[i] * DE[i] * 1.0;
                }
                if (Q.gradx) {
                    load_shared(Q.gradx, gq, nullptr, nullptr);
#pragma unroll
                    for (int i = 0; i < P; ++i) H3v[i] = -vp[i] * cE[i] * gq[i];
                } else {
#pragma unroll
                    for (int i = 0; i < P; ++i) H3v[i] = -vp[i] * cE[i] * 1.0;
                }
            } else {
#pragma unroll
                for (int i = 0; i < P; ++i) {
                    H2v[i] = DE[i];
                    H3v[i] = -cE[i];
                }
            }
        }
        // The coupled-run form has the registers to fetch H1 and H7 of its rows before its early stores (a wait for a load
        // that follows stores also waits for those stores to reach memory); the form with the extras keeps them in the loop.
        constexpr bool HOIST = (GEN == 1) && EARLY;
        double H1v[HOIST ? P : 1], h7v[HOIST ? P : 1];
        if constexpr (HOIST) {
#pragma unroll
            for (int i = 0; i < P; ++i) {
                const int j = j0 + i;
                const int jc = (FULL || j < n) ? j : n - 1;
                H1v[i] = Q.H1 ? Q.H1[jc] : 1.0;
                h7v[i] = Q.H7 ? Q.H7[b * Q.h7_stride + jc] : ((j < src_cut) ? S0 : 0.0);
            }
#pragma unroll
            for (int i = 0; i < P; ++i) {
                landed(H1v[i]);
                landed(h7v[i]);
            }
        }
        if constexpr (EARLY) {
            if (!Q.frozen_turb) store_state(false, true);
        }
        auto H2at = [&](int k) -> double {
            if constexpr (GENERAL) return H2v[k];
            else return DE[k];
        };
        auto H3at = [&](int k) -> double {
            if constexpr (GENERAL) return H3v[k];
            else return -cE[k];
        };
        const double h2first = H2at(0), h3first = H3at(0), h2last = H2at(P - 1), h3last = H3at(P - 1);
        double h2l, h3l, h2r, h3r;
        if constexpr (G::kBlock) {      // six halo values in one exchange: one pair of barriers instead of six
            const double v[6] = {h2last, h3last, u[P], h2first, h3first, u[1]};
            double lq[6], rq[6];
            g.template exchange<6>(v, 1, lq, rq);
            h2l = lq[0];
            h3l = lq[1];
            u[0] = lq[2];
            h2r = rq[3];
            h3r = rq[4];
            u[P + 1] = rq[5];
        } else {
            h2l = g.from_left(h2last, 1, 0.0), h3l = g.from_left(h3last, 1, 0.0);
            h2r = g.from_right(h2first, 1, 0.0), h3r = g.from_right(h3first, 1, 0.0);
            u[0] = g.from_left(u[P], 1, 0.0);
            u[P + 1] = g.from_right(u[1], 1, 0.0);
        }

        // Assemble row by row and eliminate at once.  Half-grid slot "m" lies between local points i-1 and i,
        // slot "p" between i and i+1; the p values of row i are the m values of row i+1.
        // With tread_lightly the assembly and solve are repeated with halved pseudo-timesteps until the step is
        // acceptable (tango/solver.py:361-371); `settled` marks the profiles that have their new iterate.
        double x[P];
        double rms = 0.0;
        int singular = 0;
        bool settled = done;
        bool tread_limit = false;
        double dtau = dt_full;
        // Upwinding weights (tango/HToMatrixFD.py:277-297); faces with H3 = 0 need none either (weight_is_moot, tg_assemble.h).  Where diffusion dominates, H2 >= |H3| dx / 2 with
        // H3 != 0, betaStar = 1 - H2/(c dx) <= 1/2 for c > 0 and -H2/(c dx) >= 1/2 for c < 0, so the weight is 1/2
        // whatever the quotient (the test is exact: halving is, and the rounded quotient is monotone in its
        // operands).  When that holds at every half point the wavefront uses, the quotients are skipped; the weights
        // and everything computed from them are the same numbers either way.  The half point left of the first grid
        // point carries H2 = H3 = 0 and its weight multiplies zeros; rows past the end of a ragged grid and the
        // Dirichlet row do not use theirs.
        bool half = true;
        {
            // (plain & and |: evaluated for every face, no branch per face)
            auto dominated = [&](double h2, double h3) { return weight_is_moot(h2, h3, rc.dx); };
            if (!lane_first && (FULL || j0 < n - 1)) half = dominated((h2l + h2first) / 2.0, (h3l + h3first) / 2.0);
            double a2 = h2first, a3 = h3first;
#pragma unroll
            for (int i = 0; i < P; ++i) {
                const int j = j0 + i;
                const bool used = FULL ? !(i == P - 1 && t == G::T - 1) : (j < n - 1);
                const double n2 = (i == P - 1) ? h2r : H2at(i < P - 1 ? i + 1 : 0);
                const double n3 = (i == P - 1) ? h3r : H3at(i < P - 1 ? i + 1 : 0);
                half = half & (!used | dominated((a2 + n2) / 2.0, (a3 + n3) / 2.0));
                a2 = n2;
                a3 = n3;
            }
        }
        const bool half_everywhere = __all(half);
#pragma unroll 1
        for (int attempt = 0;; ++attempt) {
        ChunkElim<P, DIV> el;
        double mx = 0.0, ss = 0.0;
        // (the face left of the first point does not exist: its values are zeros, here through the factor, 1/2 or 0, that
        // the lane keeps; halving by a product is exact)
        double hm2 = (h2l + h2first) * face0, hm3 = (h3l + h3first) * face0;
        double bm = 0.5;
        if (!half_everywhere) bm = fbeta_point<DIV>(hm2, hm3, rc.dx);     // a branch the whole wavefront takes or skips
        double h2c = h2first, h3c = h3first;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int j = j0 + i;
            const bool valid = FULL || (j < n);
            const bool last = FULL ? (i == P - 1 && t == G::T - 1) : (j == n - 1);
            const double h2n = (i == P - 1) ? h2r : H2at(i < P - 1 ? i + 1 : 0);
            const double h3n = (i == P - 1) ? h3r : H3at(i < P - 1 ? i + 1 : 0);
            const double hp2 = (h2c + h2n) / 2.0;
            const double hp3 = (h3c + h3n) / 2.0;
            double bp = 0.5;
            if (!half_everywhere) bp = fbeta_point<DIV>(hp2, hp3, rc.dx);
            // H1 and H7 come as arrays in the general forms only (the launcher sends such calls there), so that the built-in
            // form has no load after its first store (a wait for such a load would also wait for the stores before it to
            // reach memory); the coupled-run form has fetched them ahead of its early stores (HOIST).
            double H1 = 1.0, h7;
            if constexpr (SRC_KEPT) h7 = src_row[i];
            else h7 = (j < src_cut) ? S0 : 0.0;
            if constexpr (HOIST) {
                H1 = H1v[i];
                h7 = h7v[i];
            } else if constexpr (GENERAL) {
                const int jc = valid ? j : n - 1;
                H1 = Q.H1 ? Q.H1[jc] : 1.0;
                h7 = Q.H7 ? Q.H7[b * Q.h7_stride + jc] : h7;
            }
            double A, B, C, f;
            fd_row<EXACT>(last, H1, um1[i], rc, UL, hm2, hm3, bm, hp2, hp3, bp, 0.0, 0.0, h7, A, B, C, f);
            if (!FULL) {
                A = valid ? A : 0.0;
                B = valid ? B : 1.0;
                C = valid ? C : 0.0;
                f = valid ? f : 0.0;
            }
            // residual of the current profile with the new matrix (tango/fieldgroups.py:139)
            const double r = A * u[i + 2] + B * u[i + 1] + C * u[i] - f;
            ss += valid ? r * r : 0.0;
            mx = fmax(mx, valid ? fabs(f) : 0.0);
            el.push(i, C, B, A, f);
            hm2 = hp2;
            hm3 = hp3;
            bm = bp;
            h2c = h2n;
            h3c = h3n;
        }
        // rms of r / max|f| (tango/fieldgroups.py:140, tango/solver.py:464-470): the first solve's matrix decides
        if (attempt == 0) {
            double mxv[1] = {mx}, ssv[1] = {ss};
            g.template max_and_sum<1>(mxv, ssv);
            rms = sqrt(inv_n * quot<DIV>(ssv[0], mxv[0] * mxv[0]));
        }

        double xt[P];
        el.finish(g, xt);
        if constexpr (!EXTRAS) {
#pragma unroll
            for (int i = 0; i < P; ++i) x[i] = xt[i];
            singular = el.singular;
            break;
        } else {
#pragma unroll
            for (int i = 0; i < P; ++i) x[i] = settled ? x[i] : xt[i];
            singular |= settled ? 0 : el.singular;
            if (!Q.tread_lightly) break;
            // tango/solver.py:564-595: a change of maxFractionalChange or more anywhere in the inner part
            int far = 0;
#pragma unroll
            for (int i = 0; i < P; ++i) {
                const double change = (xt[i] - u[i + 1]) / u[i + 1];
                far |= (j0 + i < Q.tread_inner_count) && (fabs(change) >= Q.tread_max_change);
            }
            far = g.any2(far);
            if (!settled && far && attempt >= TG_TREAD_MAX_HALVINGS) tread_limit = true;
            settled = settled || !far || attempt >= TG_TREAD_MAX_HALVINGS;
            if (G::kBlock ? settled : !__any(!settled)) break;
            dtau = settled ? dtau : dtau / 2.0;
            rc = Recip(Q.dx, dtau);
        }
        }   // attempts
        int bad = 0;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            bad |= (FULL || j0 + i < n) ? !(isfinite(x[i]) && x[i] >= 0.0) : 0;
            u[i + 1] = hold ? u[i + 1] : x[i];
        }
        bad = g.any2(bad | (singular << 1));

        if (!done) {
            if (t == 0 && S_.err && it0 < S_.err_stride) S_.err[b * S_.err_stride + it0] = rms;
            // history of the iterates (profilesAllIterations, tango/solver.py:225-227)
            if (Q.iterates && it0 < Q.iterates_rows) store_chunk<P, FULL>(Q.iterates + ((int64_t)it0 * S_.nb + b) * S_.n, n, j0, u + 1);
            ++it0;
            if (!Q.frozen_turb) st |= TG_ST_SEEDED;
            if (!Q.fixed_count && rms < Q.tol) st |= TG_ST_CONVERGED;
            if (bad & 1) st |= TG_ST_INVALID;
            if (bad & 2) st |= TG_ST_SINGULAR;
            if (tread_limit) st |= TG_ST_TREAD_LIMIT;
            done = !Q.fixed_count && (st & (TG_ST_CONVERGED | TG_ST_INVALID));
        }
        // leave when every profile handled by this wavefront (workgroup) is done
        if constexpr (!MULTI) break;
        if (G::kBlock ? done : !__any(!done)) break;
    }

    store_state(true, !EARLY && !Q.frozen_turb);
    if (t == 0 && active) {
        S_.iters[b] = it0;
        S_.status[b] = st;
    }
}

// ------------------------------------------------------------------------------------------
// Launch geometry: lanes per profile for a grid of n points at P points per lane.
// ------------------------------------------------------------------------------------------
int grid_1d(int64_t total, int block) {
    int64_t g = (total + block - 1) / block;
    const int64_t cap = 256 * 32;  // CUs x resident blocks; grid-stride beyond that
    return (int)(g < cap ? (g > 0 ? g : 1) : cap);
}

// HEAVY: kernels whose per-lane state needs more than the 128 registers a 1024-thread workgroup leaves per lane;
// they take 4096 < n <= 8192 as 512 lanes x 16 points.
template <template <int, class> class Launcher, bool HEAVY = false, class... Args>
int dispatch_group(int64_t nb, int n, hipStream_t s, Args... args) {
    constexpr int P = 8;
    const int lanes = pow2_at_least((n + P - 1) / P);
    if (lanes <= 8) return Launcher<P, WaveGroup<8>>::go(nb, n, s, args...);
    if (lanes == 16) return Launcher<P, WaveGroup<16>>::go(nb, n, s, args...);
    if (lanes == 32) return Launcher<P, WaveGroup<32>>::go(nb, n, s, args...);
    if (lanes == 64) return Launcher<P, WaveGroup<64>>::go(nb, n, s, args...);
    if (lanes == 128) return Launcher<P, BlockGroup<128>>::go(nb, n, s, args...);
    if (lanes == 256) return Launcher<P, BlockGroup<256>>::go(nb, n, s, args...);
    if (lanes == 512) return Launcher<P, BlockGroup<512>>::go(nb, n, s, args...);
    if (lanes == 1024) {
        if constexpr (HEAVY) return Launcher<16, BlockGroup<512>>::go(nb, n, s, args...);
        else return Launcher<P, BlockGroup<1024>>::go(nb, n, s, args...);
    }
    if (n <= 16384) return Launcher<16, BlockGroup<1024>>::go(nb, n, s, args...);
    return fail(TG_ERR_UNSUPPORTED, "point-fastest solvers support n <= 16384%s");
}

template <int P, class G>
struct TridiagLauncher {
    static int go(int64_t nb, int n, hipStream_t s, const double *A, const double *B, const double *C, const double *D,
                  double *u, int32_t *status) {
        int grid, block;
        size_t lds;
        geometry<G>(nb, grid, block, lds);
        const bool full = (n == P * G::T) && ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) |
                                                 reinterpret_cast<uintptr_t>(C) | reinterpret_cast<uintptr_t>(D) |
                                                 reinterpret_cast<uintptr_t>(u)) & 15) == 0;
        if (kWaveCoop<P, G, true> && full) lds += (size_t)(block / 64) * 2 * WaveTile<8>::kDoubles * sizeof(double);
        if (full) hipLaunchKernelGGL((tridiag_pf_kernel<P, G, true>), dim3(grid), dim3(block), lds, s, A, B, C, D, u, nb, n, status);
        else hipLaunchKernelGGL((tridiag_pf_kernel<P, G, false>), dim3(grid), dim3(block), lds, s, A, B, C, D, u, nb, n, status);
        return launch_check();
    }
};

template <int P, class G>
struct Picard1fLauncher {
    static int go(int64_t nb, int n, hipStream_t s, const tg_state_1f *st, const tg_params_1f *q) {
        int grid, block;
        size_t lds;
        geometry<G>(nb, grid, block, lds);
        if (!G::kBlock && q->iters_per_launch > 1) {
            // A wavefront that keeps a profile for many iterations lives long; alone in its workgroup it hands its slot and
            // its LDS tiles to the next one when IT is done, not when the slowest of four is (+1 % at 20 iterations per
            // launch; the single-iteration form is 2-3 % slower that way and keeps four wavefronts per workgroup)
            block = 64;
            grid = (int)((nb + 64 / G::T - 1) / (64 / G::T));
        }
        uintptr_t al = reinterpret_cast<uintptr_t>(st->u) | reinterpret_cast<uintptr_t>(st->um1) |
                       reinterpret_cast<uintptr_t>(st->uE) | reinterpret_cast<uintptr_t>(st->DE) |
                       reinterpret_cast<uintptr_t>(st->cE) | reinterpret_cast<uintptr_t>(q->flux) |
                       reinterpret_cast<uintptr_t>(q->Vprime) | reinterpret_cast<uintptr_t>(q->gxx) |
                       reinterpret_cast<uintptr_t>(q->gradx) | reinterpret_cast<uintptr_t>(q->gxx_turb) |
                       reinterpret_cast<uintptr_t>(q->gradx_turb) | reinterpret_cast<uintptr_t>(q->iterates);
        const bool full = (n == P * G::T) && (al & 15) == 0;
        const bool extras = q->flux_model == TG_FLUX_POWER || q->pseudo_time || q->tread_lightly;
        const bool general = extras || q->flux_model != TG_FLUX_SHESTAKOV || q->Vprime || q->gxx_turb || q->gradx_turb || q->H1 || q->H7 ||
                             q->smooth_window > 1 || q->flux_mult;
        const bool multi = q->iters_per_launch > 1;
        if (kWaveCoop<P, G, true> && full) lds += (size_t)(block / 64) * kCoopTiles<G> * WaveTile<8>::kDoubles * sizeof(double);
        tg_params_1f qs = *q;
        qs.reserved_ = 0;
        if (q->smooth_window > 1) {
            // one padded row per profile for the flux smoothing: behind the exchange area and the tiles, except where a
            // wavefront owns whole profiles and has tiles: there the row lies in them
            if (n > 4096) return fail(TG_ERR_UNSUPPORTED, "flux smoothing inside the fused step needs n <= 4096%s");
            const bool in_tiles = !G::kBlock && kWaveCoop<P, G, true> && full;
            if (!in_tiles) {
                qs.reserved_ = (int32_t)(lds / sizeof(double));
                lds += sizeof(double) * (G::kBlock ? (size_t)G::T * P * 9 / 8 + 8 : (size_t)(block / 64) * 576);
            }
            if (lds > max_lds_bytes()) return fail(TG_ERR_UNSUPPORTED, "flux smoothing: not enough LDS for this grid size%s");
        }
        q = &qs;
#define TG_LAUNCH_1F(F, GEN, M) hipLaunchKernelGGL((picard_1f_kernel<P, G, F, GEN, M>), dim3(grid), dim3(block), lds, s, *st, *q)
        if (extras && multi) {
            if (full) TG_LAUNCH_1F(true, 2, true);
            else TG_LAUNCH_1F(false, 2, true);
        } else if (extras) {
            if (full) TG_LAUNCH_1F(true, 2, false);
            else TG_LAUNCH_1F(false, 2, false);
        } else if (general && multi) {
            if (full) TG_LAUNCH_1F(true, 1, true);
            else TG_LAUNCH_1F(false, 1, true);
        } else if (general) {
            if (full) TG_LAUNCH_1F(true, 1, false);
            else TG_LAUNCH_1F(false, 1, false);
        } else if (multi) {
            if (full) TG_LAUNCH_1F(true, 0, true);
            else TG_LAUNCH_1F(false, 0, true);
        } else {
            if (full) TG_LAUNCH_1F(true, 0, false);
            else TG_LAUNCH_1F(false, 0, false);
        }
#undef TG_LAUNCH_1F
        return launch_check();
    }
};

}  // namespace

// argument checks shared by tg_picard_1f_step and tg_picard_1f_step_static (C++ linkage: not part of the C ABI)
int tg_picard_1f_validate(const tg_state_1f *st, const tg_params_1f *q) {
    if (!st || !q) return fail(TG_ERR_ARG, "NULL state or params%s");
    if (int e = check_dims(st->nb, st->n, 3)) return e;
    if (!st->u || !st->um1 || !st->uE || !st->DE || !st->cE || !st->status || !st->iters) return fail(TG_ERR_ARG, "NULL state array%s");
    if (!q->UL) return fail(TG_ERR_ARG, "UL is required%s");
    if (!q->H7 && (!q->S0 || !q->src_cut)) return fail(TG_ERR_ARG, "either H7 or (S0, src_cut) is required%s");
    if (q->flux_model == TG_FLUX_EXTERNAL && !q->flux) return fail(TG_ERR_ARG, "external flux model needs a flux array%s");
    if (q->flux_model != TG_FLUX_EXTERNAL && q->flux_model != TG_FLUX_SHESTAKOV && q->flux_model != TG_FLUX_POWER)
        return fail(TG_ERR_ARG, "unknown flux model%s");
    if (q->flux_model == TG_FLUX_POWER && !(q->flux_power >= 0.0)) return fail(TG_ERR_ARG, "TG_FLUX_POWER needs flux_power >= 0%s");
    if (q->flux_mult && q->flux_mult_stride != 0 && q->flux_mult_stride != st->n)
        return fail(TG_ERR_ARG, "flux_mult_stride must be 0 (one array for all profiles) or n%s");
    if ((q->flux_mult || q->flux_model == TG_FLUX_POWER) && st->n > 4096)
        return fail(TG_ERR_UNSUPPORTED, "a flux multiplier or the power-law flux inside the fused step needs n <= 4096%s");
    if (q->flux_model == TG_FLUX_EXTERNAL && q->iters_per_launch > 1) return fail(TG_ERR_ARG, "an external flux is valid for one iteration only%s");
    if (q->tread_lightly && !q->pseudo_time) return fail(TG_ERR_ARG, "tread_lightly needs pseudo_time (the previous iterate in the u_{m-1} slot)%s");
    if (q->tread_lightly && (q->tread_inner_count < 0 || q->tread_inner_count > st->n || !(q->tread_max_change > 0.0)))
        return fail(TG_ERR_ARG, "tread_lightly needs 0 <= tread_inner_count <= n and tread_max_change > 0%s");
    if (q->tread_initial_steps < 0 || (q->tread_initial_steps > 0 && (!q->pseudo_time || !(q->tread_initial_dtau > 0.0))))
        return fail(TG_ERR_ARG, "tread_initial_steps needs pseudo_time and tread_initial_dtau > 0%s");
    if (q->smooth_window > 1 && (q->smooth_window % 2 == 0 || q->smooth_window > st->n))
        return fail(TG_ERR_ARG, "smooth_window must be odd and no larger than the grid (callers add one to an even window, tango/smoother.py:50-57)%s");
    if (st->err && st->err_stride <= 0) return fail(TG_ERR_ARG, "err_stride must be positive%s");
    if (q->iterates && q->iterates_rows < 0) return fail(TG_ERR_ARG, "iterates_rows must not be negative%s");
    return TG_OK;
}

int tg_picard_1f_dispatch(const tg_state_1f *st, const tg_params_1f *q, hipStream_t s) {
    return dispatch_group<Picard1fLauncher, true>(st->nb, (int)st->n, s, st, q);
}

// ==========================================================================================
// C ABI
// ==========================================================================================
extern "C" {

int tg_version(void) { return 100; }
const char *tg_last_error(void) { return tg_err_buf; }

int tg_device_count(int *count) {
    if (!count) return fail(TG_ERR_ARG, "count is NULL%s");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *count = 0;
        return fail(TG_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = c;
    return TG_OK;
}
int tg_set_device(int device) { TG_HIP(hipSetDevice(device)); return TG_OK; }
int tg_malloc(void **ptr, size_t bytes) {
    if (!ptr) return fail(TG_ERR_ARG, "ptr is NULL%s");
    TG_HIP(hipMalloc(ptr, bytes ? bytes : 8));
    return TG_OK;
}
int tg_free(void *ptr) { TG_HIP(hipFree(ptr)); return TG_OK; }
int tg_host_alloc(void **ptr, size_t bytes) {
    if (!ptr) return fail(TG_ERR_ARG, "ptr is NULL%s");
    TG_HIP(hipHostMalloc(ptr, bytes ? bytes : 8, hipHostMallocDefault));
    return TG_OK;
}
int tg_host_free(void *ptr) { TG_HIP(hipHostFree(ptr)); return TG_OK; }
int tg_memset(void *ptr, int value, size_t bytes, tg_stream stream) { TG_HIP(hipMemsetAsync(ptr, value, bytes, S(stream))); return TG_OK; }
int tg_copy_h2d(void *dst, const void *src, size_t bytes, tg_stream stream) { TG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, S(stream))); return TG_OK; }
int tg_copy_d2h(void *dst, const void *src, size_t bytes, tg_stream stream) { TG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, S(stream))); return TG_OK; }
int tg_copy_d2d(void *dst, const void *src, size_t bytes, tg_stream stream) { TG_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, S(stream))); return TG_OK; }
int tg_copy_d2d_2d(void *dst, size_t dst_pitch, const void *src, size_t src_pitch, size_t width_bytes, size_t rows, tg_stream stream) {
    if (width_bytes > dst_pitch || width_bytes > src_pitch) return fail(TG_ERR_ARG, "row width exceeds a pitch%s");
    if (width_bytes == 0 || rows == 0) return TG_OK;
    TG_HIP(hipMemcpy2DAsync(dst, dst_pitch, src, src_pitch, width_bytes, rows, hipMemcpyDeviceToDevice, S(stream)));
    return TG_OK;
}
int tg_stream_create(tg_stream *stream) {
    if (!stream) return fail(TG_ERR_ARG, "stream is NULL%s");
    hipStream_t s;
    TG_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = s;
    return TG_OK;
}
int tg_stream_destroy(tg_stream stream) { TG_HIP(hipStreamDestroy(S(stream))); return TG_OK; }
int tg_stream_sync(tg_stream stream) { TG_HIP(hipStreamSynchronize(S(stream))); return TG_OK; }

size_t tg_tridiag_work_bytes(int64_t nb, int64_t n, int layout) {
    return layout == TG_BATCH_FASTEST ? 3 * sizeof(double) * (size_t)nb * (size_t)n : 0;     // pivots, du, du2
}

int tg_tridiag_solve(const double *A, const double *B, const double *C, const double *D, double *u, int64_t nb, int64_t n,
                     int layout, int32_t *status, void *work, size_t work_bytes, tg_stream stream) {
    if (int e = check_dims(nb, n, 1)) return e;
    if (!A || !B || !C || !D || !u) return fail(TG_ERR_ARG, "NULL array%s");
    if (nb == 0) return TG_OK;
    if (layout == TG_BATCH_FASTEST) {
        if (!work || work_bytes < tg_tridiag_work_bytes(nb, n, layout)) return fail(TG_ERR_ARG, "work space too small%s");
        if (u == A || u == B || u == C) return fail(TG_ERR_ARG, "batch-fastest solve: u may alias D only%s");
        double *wd = (double *)work, *wdu = wd + nb * n, *wdu2 = wdu + nb * n;
        hipLaunchKernelGGL(tridiag_bf_kernel<8>, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, S(stream), A, B, C, D, u, nb,
                           (int)n, wd, wdu, wdu2, status);
        return launch_check();
    }
    if (layout == TG_BATCH_FASTEST_TILED) {
        if (n > 512) return fail(TG_ERR_UNSUPPORTED, "TG_BATCH_FASTEST_TILED takes grids of up to 512 points (TG_BATCH_FASTEST has no limit)%s");
        constexpr int W = 16;
        const size_t lds = sizeof(double) * 2 * W * kTileStride;      // beyond 64 KiB: has to be asked for
        if (lds > max_lds_bytes()) return fail(TG_ERR_UNSUPPORTED, "not enough LDS on this device%s");
        const unsigned grid = (unsigned)((nb + W - 1) / W);
        if (n == 512 && nb % W == 0) {
            TG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&tridiag_bt_kernel<W, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((tridiag_bt_kernel<W, true>), dim3(grid), dim3(64 * W), lds, S(stream), A, B, C, D, u, nb, (int)n, status);
        } else {
            TG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&tridiag_bt_kernel<W, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL((tridiag_bt_kernel<W, false>), dim3(grid), dim3(64 * W), lds, S(stream), A, B, C, D, u, nb, (int)n, status);
        }
        return launch_check();
    }
    if (layout != TG_POINT_FASTEST) return fail(TG_ERR_ARG, "unknown layout%s");
    return dispatch_group<TridiagLauncher>(nb, (int)n, S(stream), A, B, C, D, u, status);
}

int tg_h_to_matrix(double dt, const double *dt_b, double dx, const double *UL, const double *U_mminus1, const tg_hcoeffs *H,
                   double *A, double *B, double *C, double *D, double *G, int64_t nb, int64_t n, tg_stream stream) {
    if (int e = check_dims(nb, n, 2)) return e;
    if (!H || !H->H1) return fail(TG_ERR_ARG, "H1 is required%s");
    if (!UL || !U_mminus1 || !A || !B || !C || !D) return fail(TG_ERR_ARG, "NULL array%s");
    if (nb == 0) return TG_OK;
    HC hc{*H};
    hipLaunchKernelGGL(h_to_matrix_kernel, dim3(grid_1d(nb * n, 256)), dim3(256), 0, S(stream), dt, dt_b, dx, UL, U_mminus1, hc,
                       A, B, C, D, G, nb, (int)n);
    return launch_check();
}

int tg_residual(const double *A, const double *B, const double *C, const double *f, const double *u, const double *G,
                const double *w, double *resid, double *sumsq, int64_t nb, int64_t n, tg_stream stream) {
    if (int e = check_dims(nb, n, 1)) return e;
    if (!A || !B || !C || !f || !u) return fail(TG_ERR_ARG, "NULL array%s");
    if ((G != nullptr) != (w != nullptr)) return fail(TG_ERR_ARG, "G and w go together%s");
    if (nb == 0) return TG_OK;
    hipLaunchKernelGGL(residual_kernel, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0, S(stream), A, B, C, f, u, G, w, resid,
                       sumsq, nb, (int)n);
    return launch_check();
}

int tg_ewma(const double *y, double *y_ewma_inout, double param, int seeded, int64_t count, tg_stream stream) {
    if (count < 0 || !y || !y_ewma_inout) return fail(TG_ERR_ARG, "bad argument%s");
    if (count == 0) return TG_OK;
    hipLaunchKernelGGL(ewma_kernel, dim3(grid_1d(count, 256)), dim3(256), 0, S(stream), y, y_ewma_inout, param, seeded, count);
    return launch_check();
}

int tg_ewma_profiles(const double *y, double *y_ewma_inout, double param, const int32_t *status, int64_t nb, int64_t n, tg_stream stream) {
    if (int e = check_dims(nb, n, 1)) return e;
    if (!y || !y_ewma_inout || !status) return fail(TG_ERR_ARG, "NULL array%s");
    if (nb == 0) return TG_OK;
    hipLaunchKernelGGL(ewma_profiles_kernel, dim3(grid_1d(nb * n, 256)), dim3(256), 0, S(stream), y, y_ewma_inout, param, status, nb, (int)n);
    return launch_check();
}

int tg_flux_split(const double *flux, const double *p, double dx, double Dmin, double Dmax, double dpdx_thr, const double *gxx,
                  const double *gradx, double *D, double *c, double *DHat, double *cHat, double *theta, int64_t nb, int64_t n,
                  tg_stream stream) {
    if (int e = check_dims(nb, n, 2)) return e;
    if (!flux || !p || !D || !c) return fail(TG_ERR_ARG, "NULL array%s");
    if (nb == 0) return TG_OK;
    hipLaunchKernelGGL(flux_split_kernel, dim3(grid_1d(nb * n, 256)), dim3(256), 0, S(stream), flux, p, dx,
                       Theta(Dmin, Dmax, dpdx_thr), gxx, gradx, D, c, DHat, cHat, theta, nb, (int)n);
    return launch_check();
}

int tg_ddx(const double *u, double dx, int edge_order, double *out, int64_t nb, int64_t n, tg_stream stream) {
    if (int e = check_dims(nb, n, edge_order == 2 ? 3 : 2)) return e;
    if (!u || !out || (edge_order != 1 && edge_order != 2)) return fail(TG_ERR_ARG, "bad argument%s");
    if (nb == 0) return TG_OK;
    hipLaunchKernelGGL(ddx_kernel, dim3(grid_1d(nb * n, 256)), dim3(256), 0, S(stream), u, dx, edge_order, out, nb, (int)n);
    return launch_check();
}

int tg_moving_average(const double *x, int window, double *out, int64_t nb, int64_t n, tg_stream stream) {
    if (int e = check_dims(nb, n, 1)) return e;
    if (!x || !out || window < 1 || window % 2 == 0) return fail(TG_ERR_ARG, "window must be a positive odd integer%s");
    if (window > n) return fail(TG_ERR_ARG, "window larger than the grid%s");
    if (nb == 0) return TG_OK;
    hipLaunchKernelGGL(moving_average_kernel, dim3(grid_1d(nb * n, 256)), dim3(256), 0, S(stream), x, window, out, nb, (int)n);
    return launch_check();
}

int tg_zone_apply(const double *in, int64_t nb, int64_t nin, int64_t in_lo, int64_t zone, const double *M, const double *bias, double *out,
                  int64_t nout, int64_t out_lo, int64_t rows, int nonneg, tg_stream stream) {
    if (!in || !M || !out) return fail(TG_ERR_ARG, "NULL array%s");
    if (nb < 0 || nin < 1 || nout < 1 || zone < 1 || zone > INT32_MAX || rows < 0 || rows > INT32_MAX || in_lo < 0 || in_lo + zone > nin || out_lo < 0 ||
        out_lo + rows > nout)
        return fail(TG_ERR_ARG, "zone or output range outside its row%s");
    if (nb == 0 || rows == 0) return TG_OK;
    hipLaunchKernelGGL(zone_apply_kernel, dim3(grid_1d(nb * rows, 256)), dim3(256), 0, S(stream), in, nin, in_lo, (int)zone, M, bias, out, nout,
                       out_lo, (int)rows, nonneg, nb);
    return launch_check();
}

size_t tg_spline_work_bytes(int64_t nb, int64_t nin) {
    if (nb <= 0 || nin < 4) return 0;
    return ((size_t)5 * (size_t)nb * (size_t)(nin - 2) * sizeof(double) + 255) & ~(size_t)255;
}

int tg_spline_resample(const double *xin, const double *yin, int64_t nb, int64_t nin, const double *xout, int64_t nout,
                       int64_t out_lo, int64_t out_hi, int nonneg, double *yout, void *work, size_t work_bytes, tg_stream stream) {
    if (nb < 0 || nin < 4 || nin - 2 > 16384 || nout < 1 || nout > INT32_MAX) return fail(TG_ERR_ARG, "spline: need 4 <= nin <= 16386 and nout >= 1%s");
    if (!xin || !yin || !xout || !yout) return fail(TG_ERR_ARG, "NULL array%s");
    if (out_lo < 0 || out_hi > nout || out_lo > out_hi) return fail(TG_ERR_ARG, "spline: bad output range%s");
    if (!work || work_bytes < tg_spline_work_bytes(nb, nin)) return fail(TG_ERR_ARG, "spline: work space too small (tg_spline_work_bytes)%s");
    if (nb == 0) return TG_OK;
    const int64_t m = nin - 2, plane = nb * m;
    double *A = static_cast<double *>(work), *B = A + plane, *C = B + plane, *D = C + plane, *M = D + plane;
    hipLaunchKernelGGL(spline_rows_kernel, dim3(grid_1d(plane, 256)), dim3(256), 0, S(stream), xin, yin, nb, (int)nin, A, B, C, D);
    if (int e = launch_check()) return e;
    if (int e = dispatch_group<TridiagLauncher>(nb, (int)m, S(stream), (const double *)A, (const double *)B, (const double *)C,
                                                (const double *)D, M, (int32_t *)nullptr))
        return e;
    hipLaunchKernelGGL(spline_eval_kernel, dim3(grid_1d(nb * nout, 256)), dim3(256), 0, S(stream), xin, yin, (const double *)M, nb,
                       (int)nin, xout, (int)nout, (int)out_lo, (int)out_hi, nonneg, yout);
    return launch_check();
}

int tg_tukey_window(double *w, int64_t n, double alpha, tg_stream stream) {
    if (!w || n < 0 || n > INT32_MAX) return fail(TG_ERR_ARG, "bad argument%s");
    if (n == 0) return TG_OK;
    hipLaunchKernelGGL(tukey_window_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, S(stream), w, (int)n, alpha);
    return launch_check();
}

int tg_ar1_noise(uint64_t seed, uint64_t sequence, double amplitude, double tac, int32_t damp_points, const double *taper, double *out,
                 int64_t nb, int64_t n, tg_stream stream) {
    if (int e = check_dims(nb, n, 1)) return e;
    if (!out) return fail(TG_ERR_ARG, "NULL array%s");
    if (!(tac > 1.0)) return fail(TG_ERR_ARG, "the correlation length must exceed one grid point%s");
    if (damp_points < 0 || 2 * (int64_t)damp_points >= n) return fail(TG_ERR_ARG, "damp_points is too large (tango/extras/noisyflux.py:156)%s");
    if (nb == 0) return TG_OK;
    const double lambda = (tac - 1.0) / (tac + 1.0);
    hipLaunchKernelGGL(ar1_noise_kernel, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0, S(stream), seed, sequence, amplitude, lambda,
                       (int)damp_points, taper, out, nb, (int)n);
    return launch_check();
}

int tg_shestakov_flux(const double *nprof, double dx, double *flux, int64_t nb, int64_t n, tg_stream stream) {
    if (int e = check_dims(nb, n, 2)) return e;
    if (!nprof || !flux) return fail(TG_ERR_ARG, "NULL array%s");
    if (nb == 0) return TG_OK;
    hipLaunchKernelGGL(shestakov_flux_kernel, dim3(grid_1d(nb * n, 256)), dim3(256), 0, S(stream), nprof, dx, flux, nb, (int)n);
    return launch_check();
}

size_t tg_picard_1f_work_bytes(int64_t nb, int64_t n) { return (nb > 0 && n > 0) ? tg_seg_work_bytes(nb, n) : 0; }
size_t tg_picard_1f_work_bytes_tread(int64_t nb, int64_t n) { return (nb > 0 && n > 0) ? tg_seg_work_bytes_tread(nb, n) : 0; }

int tg_picard_1f_step(const tg_state_1f *st, const tg_params_1f *q, tg_stream stream) {
    if (int e = tg_picard_1f_validate(st, q)) return e;
    if (st->nb == 0) return TG_OK;
    if (tg_seg_applies(st, q)) return tg_seg_step(st, q, S(stream));
    return tg_picard_1f_dispatch(st, q, S(stream));
}

int tg_stream_probe(const double *const in[5], double *const out[4], int64_t count, tg_stream stream) {
    if (!in || !out || count < 0 || (count & 1)) return fail(TG_ERR_ARG, "count must be even and non-negative%s");
    for (int k = 0; k < 5; ++k)
        if (!in[k] || (reinterpret_cast<uintptr_t>(in[k]) & 15)) return fail(TG_ERR_ARG, "inputs must be 16-byte aligned%s");
    for (int k = 0; k < 4; ++k)
        if (!out[k] || (reinterpret_cast<uintptr_t>(out[k]) & 15)) return fail(TG_ERR_ARG, "outputs must be 16-byte aligned%s");
    if (count == 0) return TG_OK;
    const int64_t n2 = count / 2;
    int64_t blocks = (n2 + 255) / 256;      // one element per thread measures fastest; grid-stride beyond 2^22 blocks
    if (blocks > (1 << 22)) blocks = 1 << 22;
    hipLaunchKernelGGL(stream_probe_kernel, dim3((unsigned)blocks), dim3(256), 0, S(stream), (const double2 *)in[0],
                       (const double2 *)in[1], (const double2 *)in[2], (const double2 *)in[3], (const double2 *)in[4],
                       (double2 *)out[0], (double2 *)out[1], (double2 *)out[2], (double2 *)out[3], n2);
    return launch_check();
}

int tg_status_reduce(const int32_t *status, int64_t nb, int64_t *counts, tg_stream stream) {
    if (!status || !counts || nb < 0) return fail(TG_ERR_ARG, "bad argument%s");
    TG_HIP(hipMemsetAsync(counts, 0, 4 * sizeof(int64_t), S(stream)));
    if (nb == 0) return TG_OK;
    hipLaunchKernelGGL(status_reduce_kernel, dim3(grid_1d(nb, 256)), dim3(256), 0, S(stream), status, nb,
                       reinterpret_cast<unsigned long long *>(counts));
    return launch_check();
}

int tg_status_or(int32_t *status, int64_t nb, int32_t bits, tg_stream stream) {
    if (!status || nb < 0) return fail(TG_ERR_ARG, "bad argument%s");
    if (nb == 0) return TG_OK;
    hipLaunchKernelGGL(status_or_kernel, dim3((unsigned)((nb + 255) / 256)), dim3(256), 0, S(stream), status, nb, (int)bits);
    return launch_check();
}

int tg_residual_reduce(const int32_t *status, const int32_t *iters, const double *err, int64_t err_stride, int64_t nb, int64_t n,
                       double *out, tg_stream stream) {
    if (!status || !iters || !err || !out || nb < 0 || n < 1 || err_stride < 1) return fail(TG_ERR_ARG, "bad argument%s");
    TG_HIP(hipMemsetAsync(out, 0, 5 * sizeof(double), S(stream)));
    if (nb == 0) return TG_OK;
    hipLaunchKernelGGL(residual_reduce_kernel, dim3(grid_1d(nb, 256)), dim3(256), 0, S(stream), status, iters, err, err_stride, nb,
                       (int)n, out);
    return launch_check();
}

// Entry points of the coupled-field path are defined in tango_block.hip.

}  // extern "C"
