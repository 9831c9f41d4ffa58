// Segmented form of the fused one-field Picard iteration for long profiles (n = S x 512, S = 8, 16, 32, 64).
//
// A workgroup per profile runs out of registers beyond 4096 points (one resident workgroup per CU, its load,
// compute and store phases serialised).  Here every 512-point segment is one wavefront, exactly as a whole profile
// of up to 512 points is in picard_1f_kernel, and the coupling between segments is resolved at the level the
// chunked elimination already works on: every lane eliminates its 8 rows down to one row in the lanes' last
// unknowns y; those n/8 rows form one tridiagonal system per profile, whatever segment a lane belongs to.
//
//   halo    (one thread per segment boundary): the EWMA memories next to the boundary are copied aside
//   phase 0 (one wavefront per segment): EWMA -> flux -> split -> EWMA (stored), assembly, residual partial sums,
//            elimination of the lane's rows; the lane's last-row and first-row forms go to the work space
//   reduce  (one workgroup per profile): the reduced rows are formed and solved (same chunked elimination + PCR,
//            slack carried through), y goes to the work space; residual norm, convergence decision
//   phase 1 (one wavefront per segment): assembly and elimination again (three arrays are read instead of keeping
//            24 bytes per point of multipliers), back substitution from y, validity check, new profile stored
//
// Tango reference: tango/solver.py:188-272 on tango/lodestro_method.py:327-408, tango/HToMatrixFD.py:53-117,
// tango/tridiag.py:12-31, as for picard_1f_kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <type_traits>

#include "../../include/tango_hip.h"
#include "tg_assemble.h"
#include "tg_comm.h"
#include "tg_host.h"
#include "tg_io.h"
#include "tg_solve.h"
#include "tg_wave_io.h"

namespace {

using namespace tg;

constexpr int kSeg = 512;        // points per segment = 64 lanes x 8 points
constexpr int kRowQ = 4;         // doubles a lane leaves for the reduced system
constexpr int kHaloQ = 12;       // doubles kept per segment boundary
constexpr int kEdgeQ = 24;       // doubles a segment's two edge lanes leave for the wavefronts of the neighbouring segments
// what a launch of the segment kernels is: an ordinary iteration; the first attempt of an iteration under pseudo-timestep
// control (the new iterate is tentative: no verdict on its validity yet); a further attempt with half the pseudo-timestep for
// the profiles marked in SegWork::redo (same D and c, no bookkeeping: iteration count, residual and convergence were the
// first attempt's business)
constexpr int kPlain = 0, kFirstAttempt = 1, kRetry = 2;

// work space layout (all double unless noted), L = nb n / 8 lanes, S = n / 512 segments per profile
struct SegWork {
    double *row[kRowQ];   // each (L,): the lane's row of the reduced system (sub-diagonal, super-diagonal, right-hand side,
                          // slack); lane 63 of a segment, whose right neighbour belongs to another wavefront, leaves its
                          // last-row form (e, l, f, rho) instead
    double *first;        // (nb S, 3) first-row form (F, E, tau) of lane 0 of every segment, for lane 63 of the segment before
    double *y;            // (L,) lane-last unknowns
    double *ss, *mx;      // (nb S,) residual partial sums and max |f| per segment
    double *halo;         // (nb (S-1), 12) EWMA memories at the four points around every segment boundary, copied before a
                          // first half overwrites them: uE(j-2), uE(j-1), uE(j), uE(j+1), DE(j-1), DE(j), cE(j-1), cE(j),
                          // j = 512 (k+1); then, between two iterations of one call, the new profile at j-2 .. j+1
    double *edge;         // (nb S, 24): the eliminated rows 0..6 of lane 0 of every segment (f, l, e: 21 values) and row 6 of
                          // its lane 63 (f, l, e), left by a first half; from them and the lane unknowns the reduce kernel
                          // forms the new profile next to the segment boundaries between two iterations of one call
    int32_t *weigh;       // (nb S,) the segment's first half found a face that needs an upwinding quotient: its second half need not look again
    int32_t *flag;        // (nb,) 0: not stepped in this iteration, 1: stepped, 2: stepped and converged
    int32_t *sing;        // (nb,) zero pivot met somewhere
    // pseudo-timestep control
    double *dtau;         // (nb,) pseudo-timestep of the profile's current attempt
    int32_t *redo;        // (nb,) the last attempt moved the profile too far: another one with half the pseudo-timestep
    int32_t *attempts;    // (nb,) halvings so far in this iteration
    int32_t *nredo;       // (1,) profiles marked in redo by the last check
    double *uprev;        // (nb n) the previous iterate, only in a work space of tg_picard_1f_work_bytes_tread
};

size_t seg_work_bytes(int64_t nb, int64_t n);
size_t seg_work_bytes_tread(int64_t nb, int64_t n) { return seg_work_bytes(nb, n) + (((size_t)nb * (size_t)n * sizeof(double) + 255) & ~(size_t)255); }

size_t seg_work_bytes(int64_t nb, int64_t n) {
    const int64_t L = nb * n / 8, S = n / kSeg;
    size_t b = sizeof(double) * (size_t)((kRowQ + 1) * L + (5 + kEdgeQ) * nb * S + kHaloQ * nb * (S - 1)) + sizeof(double) * (size_t)nb +
               sizeof(int32_t) * (size_t)(4 * nb + 4 + nb * S);
    return (b + 255) & ~(size_t)255;
}

SegWork seg_carve(void *work, int64_t nb, int64_t n) {
    const int64_t L = nb * n / 8, S = n / kSeg;
    SegWork W;
    double *p = static_cast<double *>(work);
    for (int k = 0; k < kRowQ; ++k, p += L) W.row[k] = p;
    W.y = p; p += L;
    W.ss = p; p += nb * S;
    W.mx = p; p += nb * S;
    W.halo = p; p += kHaloQ * nb * (S - 1);
    W.first = p; p += 3 * nb * S;
    W.edge = p; p += kEdgeQ * nb * S;
    W.dtau = p; p += nb;
    W.flag = reinterpret_cast<int32_t *>(p);
    W.sing = W.flag + nb;
    W.redo = W.sing + nb;
    W.attempts = W.redo + nb;
    W.nredo = W.attempts + nb;
    W.weigh = W.nredo + 4;
    W.uprev = reinterpret_cast<double *>(static_cast<char *>(work) + seg_work_bytes(nb, n));
    return W;
}

// Position of lane r of a profile (r = 0 .. n/8-1) in the per-lane work arrays: transposed so that the reduce kernel,
// whose thread t owns lanes 8t .. 8t+7, reads and writes them coalesced (lane 8t+i at i n/64 + t).
__device__ __forceinline__ int64_t lane_slot(int64_t profile, int n, int r) {
    const int T = n / 64;
    return profile * (n / 8) + (int64_t)(r & 7) * T + (r >> 3);
}

// ------------------------------------------------------------------------------------------
// One row of a lane's sweep: face values of H2, H3 towards the next point, the row of tango/HToMatrixFD.py:87-117, the
// carried face values moved on.  Shared by the segment kernels and the boundary kernel, which must round alike.
// ------------------------------------------------------------------------------------------
struct FaceCarry {
    double hm2, hm3, bm, h2c, h3c;
};

// `weigh` = false: the caller has found diffusion dominant at every face it uses (faces_all_half), the upwinding weight
// is 1/2 there whatever the quotient and the quotient is skipped; the numbers are the same either way.
__device__ __forceinline__ void seg_row(FaceCarry &c, double h2n, double h3n, bool last, double H1, double um1, const Recip &rc, double UL,
                                        double h7, bool weigh, double &A, double &B, double &C, double &f) {
    const double hp2 = (c.h2c + h2n) / 2.0;
    const double hp3 = (c.h3c + h3n) / 2.0;
    double bp = 0.5;
    if (weigh) bp = fbeta_point<kDivRounded>(hp2, hp3, rc.dx);
    fd_row<true>(last, H1, um1, rc, UL, c.hm2, c.hm3, c.bm, hp2, hp3, bp, 0.0, 0.0, h7, A, B, C, f);
    c.hm2 = hp2;
    c.hm3 = hp3;
    c.bm = bp;
    c.h2c = h2n;
    c.h3c = h3n;
}

// a y + (1 - a) memory (tango/lodestro_method.py:706-711) with its rounding spelled out: one product rounded, the other
// fused into the sum.  An edge lane computes the memories of its halo point as the neighbouring wavefront computes them for
// itself, in another place of the code or in another kernel of the chain, and the two must agree to the bit: left as
// a * y + om * mem the compiler fuses whichever product it likes, not the same one everywhere.
__device__ __forceinline__ double ewma(double a, double y, double om, double mem) { return __builtin_fma(a, y, om * mem); }

// One step of the back substitution inside a lane, x_i = (f_i - l_i y_left) - e_i x_{i+1}.
__device__ __forceinline__ double back_step(double f, double l, double e, double y_left, double x_next) {
    return fma(-e, x_next, fma(-l, y_left, f));
}

// ------------------------------------------------------------------------------------------
// A first half replaces uE, DE and cE in place while neighbouring segments still need the old values next to their
// boundaries: those are copied aside first.  Boundary k of profile b lies before point jb = 512 (k + 1).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void boundary_values(const tg_state_1f &S_, int64_t b, int k, double (&v)[8]) {
    const int64_t j = b * S_.n + (int64_t)(k + 1) * kSeg;
    v[0] = S_.uE[j - 2];
    v[1] = S_.uE[j - 1];
    v[2] = S_.uE[j];
    v[3] = S_.uE[j + 1];
    v[4] = S_.DE[j - 1];
    v[5] = S_.DE[j];
    v[6] = S_.cE[j - 1];
    v[7] = S_.cE[j];
}

__device__ __forceinline__ void boundary_snapshot(const tg_state_1f &S_, const SegWork &W, int64_t b, int k) {
    const int S = (int)(S_.n / kSeg);
    double v[8];
    boundary_values(S_, b, k, v);
    double *h = W.halo + kHaloQ * (b * (S - 1) + k);
#pragma unroll
    for (int q = 0; q < 8; ++q) h[q] = v[q];
}

__global__ void seg_halo_kernel(tg_state_1f S_, SegWork W) {
    const int S = (int)(S_.n / kSeg);
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= S_.nb * (S - 1)) return;
    const int64_t b = k / (S - 1);
    boundary_snapshot(S_, W, b, (int)(k - b * (S - 1)));
}

// Between two iterations of one call, when the second half of the iteration and the first half of the next run in
// one kernel, an edge lane also needs the neighbouring segment's NEW profile at the two points next to the boundary
// before the neighbour has stored it.  One thread per side of a boundary (in the reduce kernel, which has the lane
// unknowns y of the whole profile) does the back substitution of the lane on its side, x_7 = y,
// x_i = (f_i - l_i y_left) - e_i x_{i+1}, from the eliminated rows that lane left in SegWork::edge: the same operations on
// the same numbers as the owner of those rows will perform.
// side 0: last lane of the segment on the left (row 6 is enough), 1: first lane of the segment on the right (rows 6..0).
// rec: the 21 doubles this thread works from: side 1 the rows 0..6 (f, l, e) of the first lane of the segment on the right,
// side 0 the last 21 doubles of the left segment's record, whose last three are row 6 of its lane 63; y_of_lane: the lane
// unknowns of the profile in lane order.
__device__ __forceinline__ void boundary_finish(const tg_state_1f &S_, const SegWork &W, int64_t b, int k, int side,
                                                const double *rec, const double *y_of_lane) {
    constexpr int P = 8;
    const int S = (int)(S_.n / kSeg);
    const int jb = (k + 1) * kSeg;
    double *h = W.halo + kHaloQ * (b * (S - 1) + k);
    const int lane_r = (side ? jb : jb - P) / P;
    const double y = y_of_lane[lane_r];
    const double yl = y_of_lane[lane_r - 1];
    if (side == 0) {
        h[9] = y;                                                  // u(jb-1)
        h[8] = back_step(rec[18], rec[19], rec[20], yl, y);        // u(jb-2)
    } else {
        double xx = y, x1 = 0.0;
#pragma unroll
        for (int i = P - 2; i >= 0; --i) {
            xx = back_step(rec[i], rec[7 + i], rec[14 + i], yl, xx);
            if (i == 1) x1 = xx;
        }
        h[10] = xx;                                                // u(jb)
        h[11] = x1;                                                // u(jb+1)
    }
}

// ------------------------------------------------------------------------------------------
// the two halves of an iteration: one wavefront per segment
// ------------------------------------------------------------------------------------------
// PHASE 0: first half of an iteration (turbulence update, assembly, elimination down to the lanes' last unknowns).
// PHASE 1: second half (assembly and elimination again, back substitution, new profile stored).
// PHASE 2: the second half of one iteration and, for profiles that go on, the first half of the next in the same
//          wavefront: D, c, u_{m-1} and the new profile are in registers already, so the next first half reads only the
//          profile's EWMA memory (80 instead of 115 bytes per point and iteration).  The two points of the new profile
//          that an edge lane needs from the neighbouring segment come from the reduce kernel (boundary_finish).
// wavefronts per SIMD: three (<= 168 registers) and four (<= 128) for the two halves; the combined kernel needs 216
// registers, and with them at two per SIMD it measured 17 % faster than with 168 and 48 of them spilled
// ARR: H1 or H7 come as arrays (otherwise H1 = 1 and H7 is the box source of the profile: no loads).
// No load follows a store in these kernels (a wait for such a load would also wait for the stores before it to reach
// memory): whatever the later parts need from memory is fetched at the top.
template <int PHASE, bool ARR>
__global__ void __launch_bounds__(256, PHASE == 0 ? (ARR ? 2 : 3) : (PHASE == 1 ? (ARR ? 3 : 4) : (ARR ? 2 : 3)))
picard_seg_kernel(tg_state_1f S_, tg_params_1f Q, SegWork W, int mode) {
    extern __shared__ double lds[];
    using G = WaveGroup<64>;
    constexpr int P = 8;
    constexpr bool EXACT = true;      // these grids have condition numbers above 1e9: the reference's quotients
    G g(lds);
    const int n = (int)S_.n;
    const int S = n / kSeg;
    const int wave = wave_in_block();      // the same in every lane: say so
    const int64_t w = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;   // segment index over the ensemble
    if (w >= S_.nb * S) return;
    const int64_t b = w / S;
    const int s = (int)(w - b * S);
    const int st = S_.status[b];
    int flag = 1;
    if (PHASE == 0) {
        if (mode == kRetry ? !W.redo[b] : (!Q.fixed_count && (st & (TG_ST_CONVERGED | TG_ST_INVALID)))) return;
    } else {
        flag = W.flag[b];
        if (flag == 0) return;
    }
    const int t = g.t();
    const int j0 = s * kSeg + t * P;                 // first point of this lane within the profile
    const int64_t off = b * S_.n;
    const int64_t woff = off + (int64_t)s * kSeg;
    const int lane_r = s * 64 + t;                                    // this lane within its profile
    // its slot in the per-lane work arrays (lane_slot): the profile's part is the same for the whole wavefront and stays
    // in scalar registers, the lane's own part is a 32-bit offset
    const int64_t lane_base = b * (n / 8);
    const int lane_off = (lane_r & 7) * (n / 64) + (lane_r >> 3);
    const bool first_lane = (s == 0 && t == 0);      // holds the first point of the profile
    const bool last_lane = (s == S - 1 && t == 63);  // holds the last point of the profile (Dirichlet row)
    const Recip rc(Q.dx, Q.dt_b ? Q.dt_b[b] : Q.dt);
    const double UL = Q.UL[b];
    const double S0 = Q.H7 ? 0.0 : Q.S0[b];
    const int src_cut = Q.H7 ? 0 : Q.src_cut[b];
    const Theta th(Q.Dmin, Q.Dmax, Q.dpdx_thr);
    const double omp = 1.0 - Q.a_prof, omt = 1.0 - Q.a_turb;
    const bool turb = !Q.frozen_turb;
    // the second half of the first iteration seeds the memories: a first half that follows it in this launch finds them seeded
    const bool seeded = (st & TG_ST_SEEDED) != 0 || (PHASE == 2 && turb);

    using Tile = WaveTile<8>;
    constexpr int NT = (PHASE == 1) ? 2 : 3;      // LDS tiles per wavefront: few enough for three wavefronts per SIMD
    Tile wt(lds + wave * NT * Tile::kDoubles);

    double u[P + 2], um1[P], DE[P], cE[P];
    double H1v[ARR ? P : 1], h7v[ARR ? P : 1];
    if constexpr (ARR) {
#pragma unroll
        for (int i = 0; i < P; ++i) {
            H1v[i] = Q.H1 ? Q.H1[j0 + i] : 1.0;
            h7v[i] = Q.H7 ? Q.H7[b * Q.h7_stride + j0 + i] : ((j0 + i < src_cut) ? S0 : 0.0);
        }
    }
    // values at the point just outside this wavefront's segment, on the side of this lane if it is lane 0 or 63
    // (nothing there for the two ends of the profile): new EWMA profile, H2, H3, current profile
    const bool edge_l = (t == 0 && s > 0), edge_r = (t == 63 && s < S - 1);
    const int jh = edge_l ? j0 - 1 : j0 + P;         // the halo point of an edge lane
    double h_ue = 0.0, h_D = 0.0, h_c = 0.0, h_u = 0.0;
    // the neighbouring segment's two nearest points, for the edge lanes of a first half: its new iterates are recomputed
    // here (index 0: the halo point, 1: one further out); memories as they were before this launch's first half
    double n_u0 = 0.0, n_u1 = 0.0, n_e0 = 0.0, n_e1 = 0.0, n_D = 0.0, n_c = 0.0;
    auto snapshot = [&]() {
        const double *h = W.halo + kHaloQ * (b * (S - 1) + (edge_l ? s - 1 : s));
        n_e0 = edge_l ? h[1] : h[2];
        n_e1 = edge_l ? h[0] : h[3];
        n_D = edge_l ? h[4] : h[5];
        n_c = edge_l ? h[6] : h[7];
    };

    // assembly and elimination of this lane's rows (tango/HToMatrixFD.py:87-117 row by row, as in picard_1f_kernel);
    // RESID: the residual of the current profile with the new matrix (tango/fieldgroups.py:139), zero-padded shifts at
    // the two ends of the profile
    // `known`: < 0: look at the faces; 0 / 1: the first half of this iteration has (same memories, same faces) and left its answer
    auto assemble = [&](auto resid, ChunkElim<P, kDivRounded> &el, double &ss, double &mx, int known) -> bool {
        constexpr bool RESID = decltype(resid)::value;
        // H2 = D, H3 = -c of the neighbouring lanes' nearest points; across a segment boundary the halo point's
        // (every lane takes part in the shuffles; the edge lanes then substitute their halo values)
        const double s2l = g.from_left(DE[P - 1], 1, 0.0), s3l = g.from_left(-cE[P - 1], 1, 0.0);
        const double s2r = g.from_right(DE[0], 1, 0.0), s3r = g.from_right(-cE[0], 1, 0.0);
        const double h2l = edge_l ? h_D : s2l;
        const double h3l = edge_l ? -h_c : s3l;
        const double h2r = edge_r ? h_D : s2r;
        const double h3r = edge_r ? -h_c : s3r;
        if constexpr (RESID) {
            const double ul = g.from_left(u[P], 1, 0.0), ur = g.from_right(u[1], 1, 0.0);
            u[0] = edge_l ? h_u : ul;
            u[P + 1] = edge_r ? h_u : ur;
        }
        // Upwinding weights (tango/HToMatrixFD.py:277-297); faces with H3 = 0 need none either (weight_is_moot, tg_assemble.h).  Where diffusion dominates, H2 >= |H3| dx / 2 with H3 != 0, the
        // weight is 1/2 whatever the quotient H2 / (c dx) (the test is exact: halving is, and the rounded quotient is monotone
        // in its operands).  When that holds at every face this wavefront uses, the quotients are skipped (a branch the whole
        // wavefront takes or not); weights and rows are the same numbers either way.  The face left of the profile's first
        // point carries H2 = H3 = 0 and its weight multiplies zeros; the Dirichlet row does not use its right face.
        bool weigh = known != 0;
        if (known < 0) {       // (uniform over the wavefront)
            bool half = true;
            // (plain & and |: evaluated for every face, no branch per face)
            auto dominated = [&](double h2, double h3) { return weight_is_moot(h2, h3, rc.dx); };
            if (!first_lane) half = dominated((h2l + DE[0]) / 2.0, (h3l - cE[0]) / 2.0);
            double a2 = DE[0], a3 = -cE[0];
#pragma unroll
            for (int i = 0; i < P; ++i) {
                const bool used = !((i == P - 1) && last_lane);
                const double n2 = (i == P - 1) ? h2r : DE[i < P - 1 ? i + 1 : 0];
                const double n3 = (i == P - 1) ? h3r : -cE[i < P - 1 ? i + 1 : 0];
                half = half & (!used | dominated((a2 + n2) / 2.0, (a3 + n3) / 2.0));
                a2 = n2;
                a3 = n3;
            }
            weigh = !__all(half);
        }
        FaceCarry c;
        c.hm2 = first_lane ? 0.0 : (h2l + DE[0]) / 2.0;
        c.hm3 = first_lane ? 0.0 : (h3l - cE[0]) / 2.0;
        c.bm = 0.5;
        if (weigh) c.bm = fbeta_point<kDivRounded>(c.hm2, c.hm3, rc.dx);
        c.h2c = DE[0];
        c.h3c = -cE[0];
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int j = j0 + i;
            const bool last = (i == P - 1) && last_lane;
            const double h2n = (i == P - 1) ? h2r : DE[i < P - 1 ? i + 1 : 0];
            const double h3n = (i == P - 1) ? h3r : -cE[i < P - 1 ? i + 1 : 0];
            double H1 = 1.0, h7 = (j < src_cut) ? S0 : 0.0;
            if constexpr (ARR) {
                H1 = H1v[i];
                h7 = h7v[i];
            }
            double A, B, C, f;
            seg_row(c, h2n, h3n, last, H1, um1[i], rc, UL, h7, weigh, A, B, C, f);
            if constexpr (RESID) {
                const double r = A * u[i + 2] + B * u[i + 1] + C * u[i] - f;
                ss += r * r;
                mx = fmax(mx, fabs(f));
            }
            el.push(i, C, B, A, f);
        }
        return weigh;
    };

    if (PHASE != 0) {
        // ---- second half of an iteration ----
        double2 r1[4], r2[4], r3[4], r4[4];
        wt.fetch(S_.DE + woff, r2);
        wt.fetch(S_.cE + woff, r3);
        wt.fetch(S_.um1 + woff, r4);
        if (PHASE == 2) wt.fetch(S_.uE + woff, r1);
        // this lane's last unknown and its left neighbour's (any segment), for the back substitution
        const double y = (W.y + lane_base)[lane_off];
        const double yl = first_lane ? 0.0 : (W.y + lane_base)[((lane_r - 1) & 7) * (n / 64) + ((lane_r - 1) >> 3)];
        const int weigh_known = __builtin_amdgcn_readfirstlane(W.weigh[b * S + s]);      // left by this iteration's first half
        int sing0 = 0, it_old = 0;
        if (s == 0) {
            sing0 = __builtin_amdgcn_readfirstlane(W.sing[b]);
            it_old = __builtin_amdgcn_readfirstlane(S_.iters[b]);
        }
        if (edge_l || edge_r) {
            if (PHASE == 2) {
                // the neighbours may already be replacing their memories: the copies taken before this launch
                snapshot();
                h_D = n_D;
                h_c = n_c;
                // the neighbouring segment's new profile next to the boundary, for the first half that follows
                const double *h = W.halo + kHaloQ * (b * (S - 1) + (edge_l ? s - 1 : s));
                n_u0 = edge_l ? h[9] : h[10];
                n_u1 = edge_l ? h[8] : h[11];
            } else {
                h_D = S_.DE[off + jh];
                h_c = S_.cE[off + jh];
            }
        }
        wt.stage(0, r2);
        wt.stage(1, r3);
        Tile::sync();
        wt.take(0, DE);
        wt.take(1, cE);
        Tile::sync();
        wt.stage(0, r4);
        if (PHASE == 2) wt.stage(2, r1);      // the profile's EWMA memory waits in the third tile for the next first half
        Tile::sync();
        wt.take(0, um1);
        Tile::sync();
        // What only the two edge lanes will need in the first half that follows waits in LDS meanwhile (six values per
        // edge lane, in registers they would be six doubles of every lane held through the whole second half): 179 -> 168
        // registers, three wavefronts per SIMD.
        double *park = lds + 4 * NT * Tile::kDoubles + wave * 16 + (edge_r ? 8 : 0);
        if (PHASE == 2 && (edge_l || edge_r)) {
            park[0] = n_u0;
            park[1] = n_u1;
            park[2] = n_e0;
            park[3] = n_e1;
            park[4] = n_D;
            park[5] = n_c;
        }

        ChunkElim<P, kDivRounded> el;
        double mx = 0.0, ss = 0.0;
        assemble(std::false_type{}, el, ss, mx, weigh_known);
        // back substitution
        double x[P];
        x[P - 1] = y;
#pragma unroll
        for (int i = P - 2; i >= 0; --i) x[i] = back_step(el.f[i], el.l[i], el.e[i], yl, x[i + 1]);
        int bad = 0;
#pragma unroll
        for (int i = 0; i < P; ++i) bad |= !(isfinite(x[i]) && x[i] >= 0.0);
        bad = g.any2(bad);
        wt.put(0, x);
        Tile::sync();
        wt.flush(0, S_.u + woff);
        if (t == 0) {
            // under pseudo-timestep control the iterate is tentative: seg_tread_check_kernel judges the one that stays
            int bits = (bad && mode == kPlain) ? TG_ST_INVALID : 0;
            if (s == 0) {
                if (mode != kRetry) {
                    S_.iters[b] = it_old + 1;
                    if (turb) bits |= TG_ST_SEEDED;
                    if (flag == 2) bits |= TG_ST_CONVERGED;
                }
                if (sing0) bits |= TG_ST_SINGULAR;
            }
            if (bits) atomicOr(&S_.status[b], bits);
        }
        if (PHASE == 1) return;
        // a converged profile stops here.  (A profile that this iteration left invalid in ANOTHER segment is not known to
        // be so here: its memories advance once more, then the reduce kernel drops it.)
        if (flag == 2 || bad) return;
        Tile::sync();
#pragma unroll
        for (int i = 0; i < P; ++i) u[i + 1] = x[i];
        asm volatile("" ::: "memory");      // the parked values are to be read back from LDS, not kept in registers
        if (edge_l || edge_r) {
            n_u0 = park[0];
            n_u1 = park[1];
            n_e0 = park[2];
            n_e1 = park[3];
            n_D = park[4];
            n_c = park[5];
        }
    }

    // ---- first half of an iteration ----
    {
        double ue[P];
        if (PHASE == 0) {
            double2 r0[4], r1[4], r2[4], r3[4], r4[4];
            wt.fetch(S_.u + woff, r0);
            wt.fetch(S_.uE + woff, r1);
            wt.fetch(S_.DE + woff, r2);
            wt.fetch(S_.cE + woff, r3);
            wt.fetch(S_.um1 + woff, r4);
            if (edge_l || edge_r) {
                const int j1 = edge_l ? jh - 1 : jh + 1;
                n_u0 = S_.u[off + jh];       // u changes in the second half only
                n_u1 = S_.u[off + j1];
                snapshot();
            }
            wt.stage(0, r0);
            wt.stage(1, r1);
            wt.stage(2, r2);
            Tile::sync();
            wt.take(0, u + 1);
            wt.take(1, ue);
            wt.take(2, DE);
            Tile::sync();
            wt.stage(0, r3);
            wt.stage(1, r4);
            Tile::sync();
            wt.take(0, cE);
            wt.take(1, um1);
            Tile::sync();
        } else {
            wt.take(2, ue);
            Tile::sync();
        }

        // flux model and split (tango/extras/shestakov_nonlinear_diffusion.py:59-60, tango/lodestro_method.py:624-634) through
        // the one quotient (dp/dx)/p, as in the wave-per-profile kernel: DHat and cHat are within a few ulp of the three-division
        // values, and nothing downstream amplifies that (the pivots and the rows, where these grids' condition numbers bite,
        // keep their correctly rounded quotients)
        auto split = [&](double pm, double p, double pp, bool first, bool last, double &D, double &c) {
            const double dpdx = ddx_edge1_point<EXACT>(pm, p, pp, first, last, rc);
            shestakov_split_point(p, dpdx, th, D, c);
        };
        if (turb) {
#pragma unroll
            for (int i = 0; i < P; ++i) ue[i] = ewma(Q.a_prof, u[i + 1], omp, seeded ? ue[i] : u[i + 1]);
            const double e0 = ewma(Q.a_prof, n_u0, omp, seeded ? n_e0 : n_u0);
            const double e1 = ewma(Q.a_prof, n_u1, omp, seeded ? n_e1 : n_u1);
            h_ue = e0;
            // D and c at the halo point, from the new EWMA profile on both of its sides
            {
                double D, c;
                split(edge_l ? e1 : ue[P - 1], e0, edge_l ? ue[0] : e1, false, false, D, c);
                h_D = ewma(Q.a_turb, D, omt, seeded ? n_D : D);
                h_c = ewma(Q.a_turb, c, omt, seeded ? n_c : c);
            }
            double wl = g.from_left(ue[P - 1], 1, 0.0);
            double wr = g.from_right(ue[0], 1, 0.0);
            wl = edge_l ? h_ue : wl;
            wr = edge_r ? h_ue : wr;
#pragma unroll
            for (int i = 0; i < P; ++i) {
                const bool first = (i == 0) && first_lane;
                const bool last = (i == P - 1) && last_lane;
                const double pm = (i == 0) ? wl : ue[i > 0 ? i - 1 : 0];
                const double pp = (i == P - 1) ? wr : ue[i < P - 1 ? i + 1 : 0];
                double D, c;
                split(pm, ue[i], pp, first, last, D, c);
                DE[i] = ewma(Q.a_turb, D, omt, seeded ? DE[i] : D);
                cE[i] = ewma(Q.a_turb, c, omt, seeded ? cE[i] : c);
            }
            wt.put(0, ue);
            wt.put(1, DE);
            wt.put(2, cE);
            Tile::sync();
            // (the offset taken through an empty asm: the store addresses are then formed here, not at the top of the kernel
            // to be carried, or spilled, all the way)
            int64_t wo = woff;
            asm volatile("" : "+s"(wo));
            wt.flush(0, S_.uE + wo);
            wt.flush(1, S_.DE + wo);
            wt.flush(2, S_.cE + wo);
        } else {
            h_D = n_D;
            h_c = n_c;
        }
        h_u = n_u0;

        ChunkElim<P, kDivRounded> el;
        double mx = 0.0, ss = 0.0;
        const bool weighed = assemble(std::true_type{}, el, ss, mx, -1);

        // The eliminated rows the neighbouring segments' wavefronts will want between this iteration and the next one of the
        // same call (see SegWork::edge): rows 0..6 of lane 0 and row 6 of lane 63.  They go through a 24-double record in LDS
        // so that ONE store instruction writes them: every lane takes part in the LDS writes, all but the edge lane aiming
        // at a spare slot, so there is no branch and the row loop above and the sweep below stay one block for the scheduler
        // (predicated stores behind a branch made the compiler spill 20 registers here).
        double *erec = lds + 4 * NT * Tile::kDoubles + 64 + wave * 32;
        {
            double *p0 = erec + (t == 0 ? 0 : 24), *p63 = erec + (t == 63 ? 21 : 25);
#pragma unroll
            for (int i = 0; i < P - 1; ++i) {
                p0[i] = el.f[i];
                p0[(t == 0 ? 7 : 0) + i] = el.l[i];
                p0[(t == 0 ? 14 : 0) + i] = el.e[i];
            }
            p63[0] = el.f[P - 2];
            p63[t == 63 ? 1 : 0] = el.l[P - 2];
            p63[t == 63 ? 2 : 0] = el.e[P - 2];
        }

        // what the reduced system needs from this lane: its last row x_last + e x_next = f - l y_left with slack rho,
        // and its first unknown x_0 = F - L y_left - E y with slack tau (ChunkElim::finish forms the same values)
        double F = el.f[P - 2], L = el.l[P - 2], E = el.e[P - 2], tau = el.rho[P - 2];
#pragma unroll
        for (int i = P - 3; i >= 0; --i) {
            F = el.f[i] - el.e[i] * F;
            L = el.l[i] - el.e[i] * L;
            tau = el.rho[i] - el.e[i] * tau;
            E = -el.e[i] * E;
        }
        (void)L;
        // the lane's row of the reduced system, l y_{r-1} + (1 - e L0) y_r - e E0 y_{r+1} = f - e F0 with slack rho - e T0,
        // from the first-row form (F0, E0, T0) of the lane to the right (ChunkElim::finish forms the same values); lane 63
        // leaves its own last-row form and the reduce kernel combines it with what lane 0 of the next segment left
        const double F0 = g.from_right(F, 1, 0.0), E0 = g.from_right(E, 1, 0.0), T0 = g.from_right(tau, 1, 1.0);
        const double e7 = el.e[P - 1], l7 = el.l[P - 1], f7 = el.f[P - 1], r7 = el.rho[P - 1];
        const bool own = (t == 63);
        (W.row[0] + lane_base)[lane_off] = own ? e7 : l7;
        (W.row[1] + lane_base)[lane_off] = own ? l7 : -(e7 * E0);
        (W.row[2] + lane_base)[lane_off] = own ? f7 : f7 - e7 * F0;
        (W.row[3] + lane_base)[lane_off] = own ? r7 : r7 - e7 * T0;
        Tile::sync();
        {
            // the record, one double per lane (a buffer store: the offsets of the lanes beyond it lie outside the buffer,
            // where the hardware drops the store)
            double *rbase = W.edge + (int64_t)kEdgeQ * (b * S + s);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(rbase, 0, kEdgeQ * (int)sizeof(double), 0x00020000);
            const double v = erec[t < kEdgeQ ? t : 0];
            typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
            u32x2 d;
            d.x = (unsigned)__double2loint(v);
            d.y = (unsigned)__double2hiint(v);
            __builtin_amdgcn_raw_buffer_store_b64(d, rs, t < kEdgeQ ? (unsigned)t * 8u : 0x7FFFF000u, 0, 0);
        }
        mx = g.max(mx);
        ss = g.sum(ss);
        const int sing = g.any2(el.singular);
        if (t == 0) {
            double *Fq = W.first + 3 * (b * S + s);
            Fq[0] = F;
            Fq[1] = E;
            Fq[2] = tau;
            W.ss[b * S + s] = ss;
            W.mx[b * S + s] = mx;
            W.weigh[b * S + s] = weighed ? 1 : 0;
            if (sing) atomicOr(&W.sing[b], 1);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Pseudo-timestep control (tango/solver.py:198-216, :311-402, :564-595) around the three kernels.  The previous iterate
// sits in the u_{m-1} array (the caller's array: under pseudo_time it has no other use), the pseudo-timestep of every
// profile in SegWork::dtau.
// ------------------------------------------------------------------------------------------
__global__ void seg_tread_init_kernel(tg_state_1f S_, tg_params_1f Q, SegWork W) {
    const int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (b == 0) *W.nredo = 0;
    if (b >= S_.nb) return;
    const double dt = Q.dt_b ? Q.dt_b[b] : Q.dt;
    // treadInitially: a short pseudo-timestep for a profile's first iterations (:210-215)
    W.dtau[b] = (S_.iters[b] < Q.tread_initial_steps) ? Q.tread_initial_dtau : dt;
    W.attempts[b] = 0;
    W.redo[b] = 0;
}

// After an attempt: did the new iterate (in u) move too far from the previous one (in u_{m-1}) somewhere in the inner
// part of the profile?  Then the profile is marked for another attempt with half the pseudo-timestep, up to
// TG_TREAD_MAX_HALVINGS of them; otherwise the iterate stays and gets its verdict (valid: finite and non-negative).
// One workgroup per profile.
__global__ void __launch_bounds__(256) seg_tread_check_kernel(tg_state_1f S_, tg_params_1f Q, SegWork W) {
    const int64_t b = blockIdx.x;
    if (W.flag[b] == 0) return;                 // not stepped by the attempt that just ran
    const double *u = S_.u + b * S_.n, *up = S_.um1 + b * S_.n;
    int far = 0, bad = 0;
    for (int j = threadIdx.x; j < (int)S_.n; j += blockDim.x) {
        const double x = u[j], x0 = up[j];
        far |= (j < Q.tread_inner_count) && (fabs((x - x0) / x0) >= Q.tread_max_change);
        bad |= !(isfinite(x) && x >= 0.0);
    }
    const int both = __syncthreads_or(far | (bad << 1));
    if (threadIdx.x != 0) return;
    far = both & 1;
    bad = both >> 1;
    const int a = W.attempts[b];
    if (far && a < TG_TREAD_MAX_HALVINGS) {
        // dt/2, dt/4, ... of the full timestep, whatever the first attempt used (tango/solver.py:361-371)
        W.dtau[b] = ldexp(Q.dt_b ? Q.dt_b[b] : Q.dt, -(a + 1));
        W.attempts[b] = a + 1;
        W.redo[b] = 1;
        atomicAdd(W.nredo, 1);
    } else {
        W.redo[b] = 0;
        const int bits = (bad ? TG_ST_INVALID : 0) | (far ? TG_ST_TREAD_LIMIT : 0);
        if (bits) atomicOr(&S_.status[b], bits);
    }
}

// ------------------------------------------------------------------------------------------
// the reduced system of one profile: n/8 rows, one workgroup of n/64 threads with 8 rows each
// ------------------------------------------------------------------------------------------
template <class G>
__global__ void __launch_bounds__(G::T)
picard_seg_reduce_kernel(tg_state_1f S_, tg_params_1f Q, SegWork W, int next, int mode) {
    extern __shared__ double lds[];
    constexpr int P = 8;
    G g(lds);
    const int64_t b = blockIdx.x;
    const int n = (int)S_.n;
    const int S = n / kSeg;
    const int rows = n / 8;
    const int st = S_.status[b];
    const bool active = (mode == kRetry) ? (W.redo[b] != 0) : (Q.fixed_count || !(st & (TG_ST_CONVERGED | TG_ST_INVALID)));
    if (!active) {
        if (g.t() == 0) W.flag[b] = 0;
        return;
    }
    const int t = g.t();
    // Another iteration follows in this call: one thread per side of a segment boundary prepares what the kernel that does
    // the second half of this iteration and the first half of the next needs there: copies of the memories next to the
    // boundary as the first halves have just left them (boundary_values) and the new profile there (boundary_finish, from
    // the rows the segments' edge lanes left).  Everything this takes from memory is fetched here, before the first store.
    const bool edge = next && t < 2 * (S - 1);
    double bv[8];
    // the records of the profile's segments (SegWork::edge) wait in LDS, behind the exchange area and the lane unknowns, while
    // the reduced system is solved: fetched by the whole workgroup, coalesced
    double *recs = lds + 8 * G::T + rows;
    if (next) {
        const double *src = W.edge + (int64_t)kEdgeQ * (b * S);
        for (int i = t; i < kEdgeQ * S; i += G::T) recs[i] = src[i];
    }
    const double *rec = recs + kEdgeQ * ((t >> 1) + (t & 1)) + ((t & 1) ? 0 : 3);
    if (edge && (t & 1) == 0) boundary_values(S_, b, t >> 1, bv);
    double seg_ss = 0.0, seg_mx = 0.0;      // residual partial sums of the segments
    int it0 = 0;
    if (t == 0) {
        for (int s = 0; s < S; ++s) {
            seg_ss += W.ss[b * S + s];
            seg_mx = fmax(seg_mx, W.mx[b * S + s]);
        }
        it0 = S_.iters[b];
    }
    ChunkElim<P, kDivRounded> el;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int r = t * P + i;
        const int64_t k = lane_slot(b, n, r);
        const double q0 = W.row[0][k], q1 = W.row[1][k], q2 = W.row[2][k], q3 = W.row[3][k];
        double s2 = q0, p2 = q1, rhs = q2, sig = q3;
        if ((r & 63) == 63) {
            // the last lane of a segment left (e, l, f, rho); the first-row form of the lane to the right comes from the
            // next segment, an identity row past the end of the profile
            const bool more = r + 1 < rows;
            const double *Fq = W.first + 3 * (b * S + (more ? (r + 1) >> 6 : 0));
            const double F0 = more ? Fq[0] : 0.0;
            const double E0 = more ? Fq[1] : 0.0;
            const double T0 = more ? Fq[2] : 1.0;
            sig = q3 - q0 * T0;
            s2 = q1;
            p2 = -(q0 * E0);
            rhs = q2 - q0 * F0;
        }
        const double d = sig - s2 - p2;
        el.push_slack(i, s2, d, p2, rhs, sig);
    }
    double y[P];
    el.finish(g, y);
#pragma unroll
    for (int i = 0; i < P; ++i) W.y[lane_slot(b, n, t * P + i)] = y[i];
    const int sing = g.any2(el.singular);
    if (next) {
        double *yl = lds + 8 * G::T;      // behind the group's exchange area: the lane unknowns in lane order
#pragma unroll
        for (int i = 0; i < P; ++i) yl[t * P + i] = y[i];
        __syncthreads();
        if (edge) {
            if ((t & 1) == 0) {
                double *h = W.halo + kHaloQ * (b * (S - 1) + (t >> 1));
#pragma unroll
                for (int q = 0; q < 8; ++q) h[q] = bv[q];
            }
            boundary_finish(S_, W, b, t >> 1, t & 1, rec, yl);
        }
    }

    // residual norm of the profile from the segments' partial sums (tango/fieldgroups.py:140, tango/solver.py:464-470)
    if (t == 0) {
        const double ss = seg_ss, mx = seg_mx;
        if (mode == kRetry) {       // the first attempt's matrix decided about convergence (tango/solver.py:464-470)
            W.flag[b] = 1;
        } else {
            const double rms = sqrt(1.0 / (double)n * (ss / (mx * mx)));
            if (S_.err && it0 < S_.err_stride) S_.err[b * S_.err_stride + it0] = rms;
            W.flag[b] = (!Q.fixed_count && rms < Q.tol) ? 2 : 1;
        }
        if (sing) W.sing[b] = 1;
    }
}

template <int PHASE>
void seg_launch(bool arr, unsigned grid, size_t lds, hipStream_t s, const tg_state_1f &st, const tg_params_1f &q, const SegWork &W, int mode) {
    if (arr) hipLaunchKernelGGL((picard_seg_kernel<PHASE, true>), dim3(grid), dim3(256), lds, s, st, q, W, mode);
    else hipLaunchKernelGGL((picard_seg_kernel<PHASE, false>), dim3(grid), dim3(256), lds, s, st, q, W, mode);
}

}  // namespace

// entry points for tango_hip.hip (C++ linkage: not part of the C ABI)
size_t tg_seg_work_bytes_tread(int64_t nb, int64_t n) {
    if (n != 4096 && n != 8192 && n != 16384 && n != 32768) return 0;
    return seg_work_bytes_tread(nb, n);
}

size_t tg_seg_work_bytes(int64_t nb, int64_t n) {
    if (n != 4096 && n != 8192 && n != 16384 && n != 32768) return 0;      // n/64 threads solve the reduced system: a power of two
    return seg_work_bytes(nb, n);
}

bool tg_seg_applies(const tg_state_1f *st, const tg_params_1f *q) {
    if (!q->work || tg_seg_work_bytes(st->nb, st->n) == 0 || q->work_bytes < tg_seg_work_bytes(st->nb, st->n)) return false;
    if (q->pseudo_time && q->work_bytes < tg_seg_work_bytes_tread(st->nb, st->n)) return false;      // no room for the previous iterate
    // 4096 points still fit one workgroup with the profile held in registers between the iterations of a call: measured
    // 0.34 ms per iteration at 16 per call against 0.53 for the chain of segment kernels (and 0.66 for either at one per call)
    if (st->n == 4096 && q->iters_per_launch > 1 && !q->pseudo_time) return false;
    if (q->flux_model != TG_FLUX_SHESTAKOV || q->Vprime || q->gxx_turb || q->gradx_turb || (q->tread_lightly && !q->pseudo_time) ||
        q->smooth_window > 1 || q->iterates || q->flux_mult)
        return false;
    uintptr_t al = reinterpret_cast<uintptr_t>(st->u) | reinterpret_cast<uintptr_t>(st->um1) | reinterpret_cast<uintptr_t>(st->uE) |
                   reinterpret_cast<uintptr_t>(st->DE) | reinterpret_cast<uintptr_t>(st->cE) | reinterpret_cast<uintptr_t>(q->work);
    return (al & 15) == 0;
}

int tg_seg_step(const tg_state_1f *st, const tg_params_1f *q, hipStream_t s) {
    const int64_t nb = st->nb;
    const int n = (int)st->n;
    SegWork W = seg_carve(q->work, nb, n);
    TG_HIP(hipMemsetAsync(W.sing, 0, sizeof(int32_t) * (size_t)nb, s));
    const int64_t segs = nb * (n / kSeg);
    const unsigned grid = (unsigned)((segs + 3) / 4);
    const size_t tile = 4 * WaveTile<8>::kDoubles * sizeof(double);     // one tile for each of the four wavefronts
    // + the edge lanes' parking places of the combined kernel + the records of the edge lanes' rows (32 doubles per wavefront)
    const size_t lds0 = 3 * tile + 4 * 16 * sizeof(double) + 4 * 32 * sizeof(double), lds1 = 2 * tile;
    const int64_t bounds = nb * (n / kSeg - 1);
    const int lanes = n / 64;      // threads of the reduce kernel: 8 reduced rows each
    const size_t lds_r = sizeof(double) * 8 * (size_t)lanes;
    if (lanes != 64 && lanes != 128 && lanes != 256 && lanes != 512) return fail(TG_ERR_UNSUPPORTED, "segmented form: n must be 4096, 8192, 16384 or 32768%s");
    // + the lane unknowns of the profile + the records of its segments' edge lanes
    const size_t lds_n = lds_r + sizeof(double) * (size_t)(n / 8) + sizeof(double) * kEdgeQ * (size_t)(n / kSeg);
    if (q->iters_per_launch > 1 && !q->pseudo_time && lds_n > 64 * 1024) {
        // the longest profiles: the reduce kernel's LDS goes beyond what a launch may use without asking
        if (lds_n > max_lds_bytes()) return fail(TG_ERR_UNSUPPORTED, "segmented form: not enough LDS for several iterations per call at this profile length%s");
        TG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(&picard_seg_reduce_kernel<BlockGroup<512>>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_n));
    }
    auto reduce = [&](const tg_state_1f &state, const tg_params_1f &p, int next, int mode) {
        const size_t l = next ? lds_n : lds_r;
        switch (lanes) {
        case 64: hipLaunchKernelGGL((picard_seg_reduce_kernel<WaveGroup<64>>), dim3((unsigned)nb), dim3(64), l, s, state, p, W, next, mode); break;
        case 128: hipLaunchKernelGGL((picard_seg_reduce_kernel<BlockGroup<128>>), dim3((unsigned)nb), dim3(128), l, s, state, p, W, next, mode); break;
        case 256: hipLaunchKernelGGL((picard_seg_reduce_kernel<BlockGroup<256>>), dim3((unsigned)nb), dim3(256), l, s, state, p, W, next, mode); break;
        default: hipLaunchKernelGGL((picard_seg_reduce_kernel<BlockGroup<512>>), dim3((unsigned)nb), dim3(512), l, s, state, p, W, next, mode); break;
        }
    };
    const dim3 hgrid((unsigned)((bounds + 255) / 256));
    const int iters = q->iters_per_launch > 1 ? q->iters_per_launch : 1;
    const bool arr = q->H1 || q->H7;
    if (q->pseudo_time) {
        // Pseudo-timestep control: one iteration at a time.  The previous iterate and the profiles' pseudo-timesteps go
        // into the work space; with tread_lightly every attempt is followed by a check, and while the
        // check marks profiles the host queues another attempt for them: it reads one counter per attempt, so that a call
        // with tread_lightly returns with the stream drained (the one exception to "every call is asynchronous").
        tg_params_1f p = *q;
        p.dt_b = W.dtau;
        tg_state_1f sp = *st;
        sp.um1 = W.uprev;
        const int first = q->tread_lightly ? kFirstAttempt : kPlain;
        const unsigned pgrid = (unsigned)((nb + 255) / 256);
        for (int k = 0; k < iters; ++k) {
            hipLaunchKernelGGL(seg_tread_init_kernel, dim3(pgrid), dim3(256), 0, s, *st, *q, W);
            TG_HIP(hipMemcpyAsync(W.uprev, st->u, sizeof(double) * (size_t)nb * (size_t)n, hipMemcpyDeviceToDevice, s));
            hipLaunchKernelGGL(seg_halo_kernel, hgrid, dim3(256), 0, s, sp, W);
            seg_launch<0>(arr, grid, lds0, s, sp, p, W, first);
            reduce(sp, p, 0, first);
            seg_launch<1>(arr, grid, lds1, s, sp, p, W, first);
            if (!q->tread_lightly) continue;
            tg_params_1f r = p;
            r.frozen_turb = 1;          // further attempts keep D and c (tango/solver.py:361-371)
            for (int attempt = 0;; ++attempt) {
                hipLaunchKernelGGL(seg_tread_check_kernel, dim3((unsigned)nb), dim3(256), 0, s, sp, *q, W);
                int32_t marked = 0;
                TG_HIP(hipMemcpyAsync(&marked, W.nredo, sizeof(int32_t), hipMemcpyDeviceToHost, s));
                TG_HIP(hipStreamSynchronize(s));
                if (marked == 0) break;
                if (attempt > TG_TREAD_MAX_HALVINGS) return fail(TG_ERR_HIP, "segmented form: the halving loop did not end%s");
                TG_HIP(hipMemsetAsync(W.nredo, 0, sizeof(int32_t), s));
                hipLaunchKernelGGL(seg_halo_kernel, hgrid, dim3(256), 0, s, sp, W);      // the memories as they are now
                seg_launch<0>(arr, grid, lds0, s, sp, r, W, kRetry);
                reduce(sp, r, 0, kRetry);
                seg_launch<1>(arr, grid, lds1, s, sp, r, W, kRetry);
            }
        }
        return launch_check();
    }
    // iterations of one call: first half, reduce, then (second half + next first half), reduce, ... and the last second half
    hipLaunchKernelGGL(seg_halo_kernel, hgrid, dim3(256), 0, s, *st, W);
    seg_launch<0>(arr, grid, lds0, s, *st, *q, W, kPlain);
    for (int k = 1; k < iters; ++k) {
        reduce(*st, *q, 1, kPlain);
        seg_launch<2>(arr, grid, lds0, s, *st, *q, W, kPlain);
    }
    reduce(*st, *q, 0, kPlain);
    seg_launch<1>(arr, grid, lds1, s, *st, *q, W, kPlain);
    return launch_check();
}

// ------------------------------------------------------------------------------------------
// The same step with the work space inside the library, for callers that cannot allocate.
// ------------------------------------------------------------------------------------------
// One copy of this array exists on every GPU the library is loaded on, until the process ends.  In host code the
// name `scratch` only identifies the symbol: the device address is looked up once per device with
// hipGetSymbolAddress and every launch and memset below is handed that address.
static __device__ __align__(256) unsigned char scratch[TG_STATIC_WORK_BYTES];

namespace {
constexpr int kMaxDevices = 64;
void *scratch_ptr[kMaxDevices] = {};      // device address of `scratch` on each device, resolved on first use
std::mutex scratch_mutex;

// The array of the device that is current; `stream` must belong to that device (a launch on another device's stream
// would hand that device this device's address).
int scratch_on_current_device(void **out, hipStream_t stream) {
    int dev = -1;
    TG_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return fail(TG_ERR_UNSUPPORTED, "device ordinal beyond the library's table%s");
    if (stream) {
        hipDevice_t sdev = -1;
        TG_HIP(hipStreamGetDevice(stream, &sdev));
        if ((int)sdev != dev)
            return fail(TG_ERR_ARG, "tg_picard_1f_step_static: the stream belongs to another device than the current one "
                                    "(make the stream's device current before the call)%s");
    }
    std::lock_guard<std::mutex> lock(scratch_mutex);
    if (!scratch_ptr[dev]) {
        void *p = nullptr;
        TG_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return fail(TG_ERR_HIP, "hipGetSymbolAddress returned a null address%s");
        scratch_ptr[dev] = p;
    }
    *out = scratch_ptr[dev];
    return TG_OK;
}
}  // namespace

extern "C" int tg_picard_1f_step_static(const tg_state_1f *st, const tg_params_1f *q, tg_stream stream) {
    if (int e = tg_picard_1f_validate(st, q)) return e;
    const size_t need = q->pseudo_time ? tg_seg_work_bytes_tread(st->nb, st->n) : tg_seg_work_bytes(st->nb, st->n);
    if (need == 0) return fail(TG_ERR_UNSUPPORTED, "tg_picard_1f_step_static steps profiles of 4096, 8192, 16384 or 32768 points (the sizes "
                                                   "whose step uses a work space); other sizes need none: call tg_picard_1f_step%s");
    if (need > (size_t)TG_STATIC_WORK_BYTES)
        return fail(TG_ERR_ARG, "tg_picard_1f_step_static: nb*n exceeds what the library's built-in work space of TG_STATIC_WORK_BYTES holds "
                                "(about 4.7 million grid points in all, e.g. 512 profiles of 8192 points)%s");
    if (st->nb == 0) return TG_OK;
    void *ws = nullptr;
    if (int e = scratch_on_current_device(&ws, S(stream))) return e;
    tg_params_1f p = *q;
    p.work = ws;
    p.work_bytes = TG_STATIC_WORK_BYTES;
    if (tg_seg_applies(st, &p)) return tg_seg_step(st, &p, S(stream));
    // 4096 points with several iterations per call: tg_picard_1f_step itself prefers the workgroup kernel there (the
    // profile stays in registers between the iterations), which uses no work space at all, so nothing is allocated either
    tg_params_1f one = p;
    one.iters_per_launch = 1;
    if (st->n == 4096 && q->iters_per_launch > 1 && tg_seg_applies(st, &one)) {
        p.work = nullptr;
        p.work_bytes = 0;
        return tg_picard_1f_dispatch(st, &p, S(stream));
    }
    return fail(TG_ERR_UNSUPPORTED, "tg_picard_1f_step_static: the segmented step needs the built-in flux, Cartesian geometry, no flux "
                                    "smoothing or multiplier, no iterate history and 16-byte aligned arrays%s");
}
