// Coupled-field kernels of libtango_hip.so: the pair-coupled (2x2 block) solve, the 3x3 block-tridiagonal
// solve and the fused three-field Picard iteration.  Same lane-group design as the scalar kernels
// (tango_hip.hip): a lane owns P = 4 consecutive grid points and keeps its block rows in registers.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/tango_hip.h"
#include "tg_assemble.h"
#include "tg_block.h"
#include "tg_comm.h"
#include "tg_host.h"
#include "tg_io.h"
#include "tg_solve.h"
#include "tg_wave_io.h"

namespace {

using namespace tg;

constexpr int PB = 4;   // grid points per lane in the block kernels

// ------------------------------------------------------------------------------------------
// C^U U_{j-1} + B^U U_j + A^U U_{j+1} + G^W W_j = D^U  and the same with U <-> W
// ------------------------------------------------------------------------------------------
struct PairArgs {
    const double *AU, *BU, *CU, *DU, *GW, *AW, *BW, *CW, *DW, *GU;
    double *U, *W;
};

template <int P, class G>
__global__ void __launch_bounds__(G::kBlock ? G::T : 256)
pair_solve_kernel(PairArgs a, int64_t nb, int n, int32_t *__restrict__ status) {
    extern __shared__ double lds[];
    G g(lds);
    const int64_t b = (int64_t)blockIdx.x * G::profiles_per_block(blockDim.x) + g.profile_in_block();
    if (b >= nb) return;
    const int j0 = g.t() * P;
    const int64_t off = b * (int64_t)n;
    double au[P], bu[P], cu[P], du[P], gw[P], aw[P], bw[P], cw[P], dw[P], gu[P];
    load_chunk<P, false>(a.AU + off, n, j0, au, 0.0);
    load_chunk<P, false>(a.BU + off, n, j0, bu, 1.0);
    load_chunk<P, false>(a.CU + off, n, j0, cu, 0.0);
    load_chunk<P, false>(a.DU + off, n, j0, du, 0.0);
    load_chunk<P, false>(a.GW + off, n, j0, gw, 0.0);
    load_chunk<P, false>(a.AW + off, n, j0, aw, 0.0);
    load_chunk<P, false>(a.BW + off, n, j0, bw, 1.0);
    load_chunk<P, false>(a.CW + off, n, j0, cw, 0.0);
    load_chunk<P, false>(a.DW + off, n, j0, dw, 0.0);
    load_chunk<P, false>(a.GU + off, n, j0, gu, 0.0);
    BlockElim<2, P> el;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int j = j0 + i;
        Mat<2> Bm;
        Vec<2> cv, av, d;
        cv.v[0] = (j == 0) ? 0.0 : cu[i];
        cv.v[1] = (j == 0) ? 0.0 : cw[i];
        av.v[0] = (j == n - 1) ? 0.0 : au[i];
        av.v[1] = (j == n - 1) ? 0.0 : aw[i];
        Bm.m[0][0] = bu[i];
        Bm.m[0][1] = gw[i];
        Bm.m[1][0] = gu[i];
        Bm.m[1][1] = bw[i];
        d.v[0] = du[i];
        d.v[1] = dw[i];
        el.push_diag(i, cv, Bm, av, d);
    }
    Vec<2> x[P];
    el.finish(g, x);
    double xu[P], xw[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        xu[i] = x[i].v[0];
        xw[i] = x[i].v[1];
    }
    store_chunk<P, false>(a.U + off, n, j0, xu);
    store_chunk<P, false>(a.W + off, n, j0, xw);
    if (status) {
        const int sing = g.any2(el.singular);
        if (sing && g.t() == 0) atomicOr(&status[b], TG_ST_SINGULAR);
    }
}

// ------------------------------------------------------------------------------------------
// 3x3 block-tridiagonal solve, blocks stored component-major
// ------------------------------------------------------------------------------------------
template <int P, class G>
__global__ void __launch_bounds__(G::kBlock ? G::T : 256)
block3_solve_kernel(const double *__restrict__ Cb, const double *__restrict__ Bb, const double *__restrict__ Ab,
                    const double *__restrict__ d, double *__restrict__ x, int64_t nb, int n, int32_t *__restrict__ status) {
    extern __shared__ double lds[];
    G g(lds);
    const int64_t b = (int64_t)blockIdx.x * G::profiles_per_block(blockDim.x) + g.profile_in_block();
    if (b >= nb) return;
    const int j0 = g.t() * P;
    const int64_t plane = nb * (int64_t)n, off = b * (int64_t)n;
    // every plane's P points of this lane in one piece (16-byte loads where the row allows them; element by element the
    // kernel took 0.93 ms at 16 384 x 512, so 0.64; two points per lane, three wavefronts' worth of registers less
    // and fully coalesced, but twice the cyclic-reduction work per point: 0.78 ms)
    constexpr bool STREAM = false;
    double cb[9][P], bk[9][P], ab[9][P], dd[3][P];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        load_chunk<P, false, STREAM>(Cb + k * plane + off, n, j0, cb[k], 0.0);
        load_chunk<P, false, STREAM>(Bb + k * plane + off, n, j0, bk[k], (k % 4 == 0) ? 1.0 : 0.0);
        load_chunk<P, false, STREAM>(Ab + k * plane + off, n, j0, ab[k], 0.0);
    }
#pragma unroll
    for (int r = 0; r < 3; ++r) load_chunk<P, false, STREAM>(d + r * plane + off, n, j0, dd[r], 0.0);
    BlockElim<3, P> el;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int j = j0 + i;
        Mat<3> Cm, Bm, Am;
        Vec<3> dv;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                Cm.m[r][c] = (j > 0) ? cb[r * 3 + c][i] : 0.0;
                Am.m[r][c] = (j < n - 1) ? ab[r * 3 + c][i] : 0.0;
                Bm.m[r][c] = bk[r * 3 + c][i];
            }
            dv.v[r] = dd[r][i];
        }
        el.push(i, Cm, Bm, Am, dv);
    }
    Vec<3> xs[P];
    el.finish(g, xs);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double xr[P];
#pragma unroll
        for (int i = 0; i < P; ++i) xr[i] = xs[i].v[r];
        store_chunk<P, false, STREAM>(x + r * plane + off, n, j0, xr);
    }
    if (status) {
        const int sing = g.any2(el.singular);
        if (sing && g.t() == 0) atomicOr(&status[b], TG_ST_SINGULAR);
    }
}

// ------------------------------------------------------------------------------------------
// Fused Picard iteration of the three-field problem n, pi <-> pe
// (examples/shestakov_threefield_nonlinearnu.py:52-149 on the stack of tango/solver.py:188-272)
// ------------------------------------------------------------------------------------------
// FULL: n = lanes x P and every array 16-byte aligned; one profile per wavefront or workgroup then moves its rows
// coalesced through wavefront-private LDS tiles (tg_wave_io.h), four tiles per wavefront behind the group's
// exchange area.
constexpr int kTiles3f = 4;
constexpr int kExch3f = 18;     // doubles per thread of the workgroup exchange area

// TREAD: pseudo-timestep control (tango/solver.py:198-216, :311-372 for three fields): the previous iterates take the
// u_{m-1} slots, the first iterations of a profile may use a short pseudo-timestep, and a step that moves any of the
// three profiles too far is redone from the previous iterates with dtau = dt/2, dt/4, ... (same turbulent coefficients,
// same collision frequency) until acceptable.
// EXT: the inputs come from outside the kernel (a flux model and compute_all_H callbacks in Python, or a turbulence code):
// the three fluxes as arrays evaluated on the EWMA of the profiles, which the caller has advanced (tg_ewma_profiles;
// tango/lodestro_method.py:342-350), and the callbacks' H6, H7, H8 as arrays (tango/fieldgroups.py:260-273,
// tango/JK_to_matrix_fd.py:97-112) in place of the built-in flux, sources and collision frequency.
template <int P, class G, bool FULL, bool TREAD, bool EXT>
__global__ void __launch_bounds__(G::kBlock ? G::T : 256)
picard_3f_kernel(tg_state_3f S_, tg_params_3f Q) {
    static_assert(!(TREAD && EXT), "pseudo-timestep control is compiled for the built-in closure only");
    extern __shared__ double lds[];
    G g(lds);
    const int64_t b = (int64_t)blockIdx.x * G::profiles_per_block(blockDim.x) + g.profile_in_block();
    if (b >= S_.nb) return;
    const int st = S_.status[b];
    const int it = S_.iters[b];
    if (Q.frozen_turb) {
        if ((st & TG_ST_INVALID) || (Q.outer_count > 0 && it != Q.outer_count)) return;
    } else if (!Q.fixed_count && (st & (TG_ST_CONVERGED | TG_ST_INVALID))) {
        return;
    }
    const bool seeded = (st & TG_ST_SEEDED) != 0;
    const int n = (int)S_.n;
    const int t = g.t();
    const int j0 = t * P;
    const int64_t off = b * S_.n;
    // treadInitially's short pseudo-timestep for a profile's first iterations (tango/solver.py:210-215)
    Recip rc(Q.dx, (TREAD && it < Q.tread_initial_steps) ? Q.tread_initial_dtau : Q.dt);
    const bool lane_first = (t == 0);
    const Theta th(Q.Dmin, Q.Dmax, Q.dpdx_thr);
    const double omp = 1.0 - Q.a_prof, omt = 1.0 - Q.a_turb;
    // grids of more than 512 points keep the reference's divisions (see tg_assemble.h)
    constexpr bool EXACT = (G::T * P > 512);
    constexpr int DIV = EXACT ? kDivRounded : kDivFast;      // pivots: correctly rounded reciprocals on those grids

    constexpr bool COOP = FULL && (P == 4 || P == 8) && (G::kBlock || G::T == 64);
    using Tile = WaveTile<(P == 8) ? 8 : 4>;
    constexpr int KP = Tile::KP;
    Tile wt(lds + (COOP ? (G::kBlock ? kExch3f * G::T : 0) + wave_in_block() * kTiles3f * Tile::kDoubles : 0));
    const int64_t woff = off + (G::kBlock ? (int64_t)wave_in_block() * 64 * P : 0);   // this wavefront's segment
    const bool turb = !Q.frozen_turb;

    // Per-profile scalars and the sources of this lane's rows, fetched before anything is stored: a wait for a load also
    // waits for every store issued before it (see `landed`), and the first stores come early in this kernel.
    const double nu0 = EXT ? 0.0 : Q.nu0[b];
    double UL[3], h7v[3][P];
    double h6v[EXT ? 3 : 1][EXT ? P : 1], h8v[EXT ? 2 : 1][EXT ? P : 1];      // H6 of the three fields, H8 of the pair
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        UL[s] = Q.UL[s][b];
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int jc = (FULL || j0 + i < n) ? j0 + i : n - 1;
            if constexpr (EXT) {
                h7v[s][i] = Q.H7[s] ? Q.H7[s][b * Q.h7_stride[s] + jc] : 0.0;
                h6v[s][i] = Q.H6[s] ? Q.H6[s][b * Q.h6_stride[s] + jc] : 0.0;
                if (s > 0) h8v[s > 0 ? s - 1 : 0][i] = Q.H8[s] ? Q.H8[s][b * Q.h8_stride[s] + jc] : 0.0;
            } else {
                h7v[s][i] = Q.H7[s][jc];
            }
        }
    }

    double u[3][P + 2], DE[3][P], cE[3][P];
    double2 rE[3][KP], rm[3][KP];      // coalesced path: raw uE and u_{m-1} pieces, staged when they are needed
    double2 rF[EXT ? 3 : 1][KP];       // and the raw pieces of the three outside fluxes
    if constexpr (COOP) {
        double2 ru[3][KP], rD[3][KP], rC[3][KP];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            wt.fetch(S_.u[s] + woff, ru[s]);
            wt.fetch(S_.DE[s] + woff, rD[s]);
            wt.fetch(S_.cE[s] + woff, rC[s]);
        }
        if (turb) {
#pragma unroll
            for (int s = 0; s < 3; ++s) wt.fetch(S_.uE[s] + woff, rE[s]);
        }
        if constexpr (!TREAD) {
#pragma unroll
            for (int s = 0; s < 3; ++s) wt.fetch(S_.um1[s] + woff, rm[s]);
        }
        if constexpr (EXT) {
            if (turb) {
#pragma unroll
                for (int s = 0; s < 3; ++s) wt.fetch(Q.flux[s] + woff, rF[s]);
            }
        }
        wt.stage(0, ru[0]); wt.stage(1, ru[1]); wt.stage(2, ru[2]); wt.stage(3, rD[0]);
        Tile::sync();
        wt.take(0, u[0] + 1); wt.take(1, u[1] + 1); wt.take(2, u[2] + 1); wt.take(3, DE[0]);
        Tile::sync();
        wt.stage(0, rD[1]); wt.stage(1, rD[2]); wt.stage(2, rC[0]); wt.stage(3, rC[1]);
        Tile::sync();
        wt.take(0, DE[1]); wt.take(1, DE[2]); wt.take(2, cE[0]); wt.take(3, cE[1]);
        Tile::sync();
        wt.stage(0, rC[2]);
        if (turb) {
            wt.stage(1, rE[0]); wt.stage(2, rE[1]); wt.stage(3, rE[2]);
        }
        Tile::sync();
        wt.take(0, cE[2]);
    } else {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            load_chunk<P, false>(S_.u[s] + off, n, j0, u[s] + 1, 1.0);
            load_chunk<P, false>(S_.DE[s] + off, n, j0, DE[s], 0.0);
            load_chunk<P, false>(S_.cE[s] + off, n, j0, cE[s], 0.0);
        }
    }
    // a y + (1 - a) memory, the first iterate seeding the memory (tango/lodestro_method.py:706-708).  `seeded` is the
    // same for the whole wavefront unless several profiles share it: a branch then, not a select.
    auto ewma = [&](double a, double y, double om, double mem) -> double {
        if constexpr (G::kBlock || G::T == 64) {
            if (seeded) return a * y + om * mem;
            return a * y + om * y;
        } else {
            return a * y + om * (seeded ? mem : y);
        }
    };
    // u_{m-1} from its raw pieces, through tile 0 (the others may hold the staged profile memories), and the waits for the
    // other late loads: here, on the one path every wavefront takes, so that no wait for a load is left for after the
    // first store
    double um1[3][TREAD ? 1 : P];
    if constexpr (COOP && !TREAD) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            Tile::sync();
            wt.stage(0, rm[s]);
            Tile::sync();
            wt.take(0, um1[s]);
        }
    }
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        landed(UL[s]);
#pragma unroll
        for (int i = 0; i < P; ++i) {
            landed(h7v[s][i]);
            if constexpr (EXT) {
                landed(h6v[s][i]);
                if (s > 0) landed(h8v[s > 0 ? s - 1 : 0][i]);
            }
        }
    }
    if (turb) {
        double ue[3][P + 2];
        double fx[EXT ? 3 : 1][EXT ? P : 1];      // the outside fluxes at this lane's points
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            if constexpr (COOP) wt.take(1 + s, ue[s] + 1);
            else load_chunk<P, false>(S_.uE[s] + off, n, j0, ue[s] + 1, 1.0);
            if constexpr (!EXT) {      // with outside fluxes the caller has advanced the memory: the flux was evaluated on it
#pragma unroll
                for (int i = 0; i < P; ++i) ue[s][i + 1] = ewma(Q.a_prof, u[s][i + 1], omp, ue[s][i + 1]);
                if constexpr (!COOP) store_chunk<P, false>(S_.uE[s] + off, n, j0, ue[s] + 1);
            } else if constexpr (!COOP) {
                load_chunk<P, false>(Q.flux[s] + off, n, j0, fx[s], 0.0);
            }
        }
        if constexpr (COOP && !EXT) {
            Tile::sync();
            wt.put(1, ue[0] + 1); wt.put(2, ue[1] + 1); wt.put(3, ue[2] + 1);
            Tile::sync();
            wt.flush(1, S_.uE[0] + woff); wt.flush(2, S_.uE[1] + woff); wt.flush(3, S_.uE[2] + woff);
            Tile::sync();
        }
        if constexpr (COOP && EXT) {
            Tile::sync();
            wt.stage(1, rF[0]); wt.stage(2, rF[1]); wt.stage(3, rF[2]);
            Tile::sync();
            wt.take(1, fx[0]); wt.take(2, fx[1]); wt.take(3, fx[2]);
            Tile::sync();
        }
        {
            double v[6], l[6], r[6];
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                v[s] = ue[s][P];
                v[3 + s] = ue[s][1];
            }
            g.template exchange<6>(v, 1, l, r);
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                ue[s][0] = l[s];
                ue[s][P + 1] = r[3 + s];
            }
        }
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const int j = j0 + i;
            const bool first = FULL ? (i == 0 && lane_first) : (j == 0);
            const bool last = FULL ? (i == P - 1 && t == G::T - 1) : (j == n - 1);
            double dd[3];
#pragma unroll
            for (int s = 0; s < 3; ++s) dd[s] = ddx_edge1_point<EXACT>(ue[s][i], ue[s][i + 1], ue[s][i + 2], first, last, rc);
            const double pi = ue[1][i + 1];
            if constexpr (EXT) {
                // tango/lodestro_method.py:365-369: the split of each field's own flux, then the EWMA of D and c
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    double D, c, DH, cH, thv;
                    flux_split_point<DIV>(fx[s][i], ue[s][i + 1], dd[s], 1.0, 1.0, th, D, c, DH, cH, thv);
                    DE[s][i] = ewma(Q.a_turb, D, omt, DE[s][i]);
                    cE[s][i] = ewma(Q.a_turb, c, omt, cE[s][i]);
                }
            } else if constexpr (EXACT) {
                const double Dsh = quot<DIV>(dd[1] * dd[1], pi * pi);     // D = (dpi/dx)^2 / pi^2, shared by the three fluxes
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const double flux = -Dsh * dd[s];
                    double D, c, DH, cH, thv;
                    flux_split_point<DIV>(flux, ue[s][i + 1], dd[s], 1.0, 1.0, th, D, c, DH, cH, thv);
                    DE[s][i] = ewma(Q.a_turb, D, omt, DE[s][i]);
                    cE[s][i] = ewma(Q.a_turb, c, omt, cE[s][i]);
                }
            } else {
                // the same through the quotient r = (dpi/dx)/pi: DHat = -flux/(du/dx) = r^2 for every field (0 where
                // du/dx is 0), cHat = flux/u = -r^2 (du/dx)/u, which for pi itself is -r^3
                const double r = quot<true>(dd[1], pi);
                const double Dsh = r * r;
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    const double dh = (dd[s] == 0.0) ? 0.0 : Dsh;
                    const double cHat = (s == 1) ? -(Dsh * r) : quot<true>(-(Dsh * dd[s]), ue[s][i + 1]);
                    double D, c;
                    split_from_hats<true>(dh, cHat, dd[s], th, D, c);
                    DE[s][i] = ewma(Q.a_turb, D, omt, DE[s][i]);
                    cE[s][i] = ewma(Q.a_turb, c, omt, cE[s][i]);
                }
            }
        }
        if constexpr (COOP) {
            wt.put(0, DE[0]); wt.put(1, DE[1]); wt.put(2, DE[2]); wt.put(3, cE[0]);
            Tile::sync();
            wt.flush(0, S_.DE[0] + woff); wt.flush(1, S_.DE[1] + woff); wt.flush(2, S_.DE[2] + woff); wt.flush(3, S_.cE[0] + woff);
            Tile::sync();
            wt.put(0, cE[1]); wt.put(1, cE[2]);
            Tile::sync();
            wt.flush(0, S_.cE[1] + woff); wt.flush(1, S_.cE[2] + woff);
        } else {
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                store_chunk<P, false>(S_.DE[s] + off, n, j0, DE[s]);
                store_chunk<P, false>(S_.cE[s] + off, n, j0, cE[s]);
            }
        }
    }
    if constexpr (COOP) Tile::sync();

    // halos of H2 = DE, H3 = -cE and of the current profiles
    double h2l[3], h3l[3], h2r[3], h3r[3];
    {
        double v[18], l[18], r[18];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            v[s] = DE[s][P - 1];
            v[3 + s] = -cE[s][P - 1];
            v[6 + s] = u[s][P];
            v[9 + s] = DE[s][0];
            v[12 + s] = -cE[s][0];
            v[15 + s] = u[s][1];
        }
        g.template exchange<18>(v, 1, l, r);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            h2l[s] = l[s];
            h3l[s] = l[3 + s];
            u[s][0] = l[6 + s];
            h2r[s] = r[9 + s];
            h3r[s] = r[12 + s];
            u[s][P + 1] = r[15 + s];
        }
    }
    // the zero-padded shifts of the residual (tango/fieldgroups.py:139): nothing beyond either end
    if (lane_first) {
#pragma unroll
        for (int s = 0; s < 3; ++s) u[s][0] = 0.0;
    }

    if constexpr (TREAD) {
        // the previous iterate takes the u_{m-1} slot (tango/solver.py:198-201): read from u below
    } else if constexpr (COOP) {
        // staged above
    } else {
#pragma unroll
        for (int s = 0; s < 3; ++s) load_chunk<P, false>(S_.um1[s] + off, n, j0, um1[s], 0.0);
    }

    // Upwinding weights (tango/HToMatrixFD.py:277-297); faces with H3 = 0 need none either (weight_is_moot, tg_assemble.h).  Where diffusion dominates, H2 >= |H3| dx / 2 with H3 != 0, the
    // weight is 1/2 whatever the quotient H2 / (c dx) (the test is exact: halving is, and the rounded quotient is monotone
    // in its operands).  When that holds at every face of the three fields that this wavefront uses, the quotients are
    // skipped (a branch the whole wavefront takes or not); weights and rows are the same numbers either way.  The face left
    // of the first grid point carries H2 = H3 = 0 and its weight multiplies zeros; rows past the end of a ragged grid and
    // the Dirichlet row do not use theirs.
    bool half = true;
    {
        // (plain & and |: evaluated for every face, no branch per face)
        auto dominated = [&](double h2, double h3) { return weight_is_moot(h2, h3, rc.dx); };
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            if (!lane_first && (FULL || j0 < n - 1)) half = half & dominated((h2l[s] + DE[s][0]) / 2.0, (h3l[s] + -cE[s][0]) / 2.0);
            double a2 = DE[s][0], a3 = -cE[s][0];
#pragma unroll
            for (int i = 0; i < P; ++i) {
                const bool used = FULL ? !(i == P - 1 && t == G::T - 1) : (j0 + i < n - 1);
                const double n2 = (i == P - 1) ? h2r[s] : DE[s][i < P - 1 ? i + 1 : 0];
                const double n3 = (i == P - 1) ? h3r[s] : -cE[s][i < P - 1 ? i + 1 : 0];
                half = half & (!used | dominated((a2 + n2) / 2.0, (a3 + n3) / 2.0));
                a2 = n2;
                a3 = n3;
            }
        }
    }
    const bool weigh = !__all(half);

    double xn[P];
    Vec<2> xp[P];
    double rms = 0.0;
    int singular = 0;
    bool settled = false, tread_limit = false;
    double dtau = Q.dt;
#pragma unroll 1
    for (int attempt = 0;; ++attempt) {
    ChunkElim<P, DIV> eln;
    BlockElim<2, P, DIV> elp;
    double hm2[3], hm3[3], bm[3], h2c[3], h3c[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        h2c[s] = DE[s][0];
        h3c[s] = -cE[s][0];
        hm2[s] = lane_first ? 0.0 : (h2l[s] + h2c[s]) / 2.0;
        hm3[s] = lane_first ? 0.0 : (h3l[s] + h3c[s]) / 2.0;
        bm[s] = 0.5;
        if (weigh) bm[s] = fbeta_point<DIV>(hm2[s], hm3[s], rc.dx);
    }
    double mx[3] = {0.0, 0.0, 0.0}, ss[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int j = j0 + i;
        const bool valid = FULL || j < n;      // FULL: n = lanes x P, the tests fold away
        const bool last = FULL ? (i == P - 1 && t == G::T - 1) : (j == n - 1);
        // collisional coupling nu = nu0 / ((pi + pe)/n)^{3/2} from the current profiles (example :108-110)
        double nu = 0.0;
        if constexpr (!EXT) {
            const double q = quot<DIV>(u[1][i + 1], u[0][i + 1]) + quot<DIV>(u[2][i + 1], u[0][i + 1]);
            nu = quot<DIV>(nu0, q * sqrt(q));
        }
        double A[3], B[3], C[3], f[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const double h2n = (i == P - 1) ? h2r[s] : DE[s][i < P - 1 ? i + 1 : 0];
            const double h3n = (i == P - 1) ? h3r[s] : -cE[s][i < P - 1 ? i + 1 : 0];
            const double hp2 = (h2c[s] + h2n) / 2.0;
            const double hp3 = (h3c[s] + h3n) / 2.0;
            double bp = 0.5;
            if (weigh) bp = fbeta_point<DIV>(hp2, hp3, rc.dx);
            double h6 = (s == 0) ? 0.0 : -nu;       // H6 = -nu for pi and pe (example :125, :144)
            if constexpr (EXT) h6 = h6v[s][i];
            fd_row<EXACT>(last, 1.0, TREAD ? u[s][i + 1] : um1[s][TREAD ? 0 : i], rc, UL[s], hm2[s], hm3[s], bm[s], hp2, hp3, bp, 0.0, h6, h7v[s][i], A[s], B[s],
                         C[s], f[s]);
            if (!valid) {
                A[s] = 0.0; B[s] = 1.0; C[s] = 0.0; f[s] = 0.0;
            }
            hm2[s] = hp2;
            hm3[s] = hp3;
            bm[s] = bp;
            h2c[s] = h2n;
            h3c[s] = h3n;
        }
        // G = -H8 (= -nu in the built-in closure), zero on the Dirichlet row (JK_to_matrix_fd.py:126-129); the first field of
        // the pair carries the coefficient of the second one's profile in its equation and the other way round
        double Gu = (valid && !last) ? -nu : 0.0, Gw = Gu;
        if constexpr (EXT) {
            Gu = (valid && !last) ? -h8v[0][i] : 0.0;
            Gw = (valid && !last) ? -h8v[1][i] : 0.0;
        }
        const bool inner = FULL ? !last : (j < n - 1);
        const double up[3] = {inner ? u[0][i + 2] : 0.0, inner ? u[1][i + 2] : 0.0, inner ? u[2][i + 2] : 0.0};
        double r[3];
        r[0] = A[0] * up[0] + B[0] * u[0][i + 1] + C[0] * u[0][i] - f[0];
        r[1] = A[1] * up[1] + B[1] * u[1][i + 1] + C[1] * u[1][i] + Gu * u[2][i + 1] - f[1];
        r[2] = A[2] * up[2] + B[2] * u[2][i + 1] + C[2] * u[2][i] + Gw * u[1][i + 1] - f[2];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            ss[s] += valid ? r[s] * r[s] : 0.0;
            mx[s] = fmax(mx[s], valid ? fabs(f[s]) : 0.0);
        }
        eln.push(i, C[0], B[0], A[0], f[0]);
        Mat<2> Bm;
        Vec<2> cv, av, d;
        cv.v[0] = C[1];
        cv.v[1] = C[2];
        av.v[0] = A[1];
        av.v[1] = A[2];
        Bm.m[0][0] = B[1];
        Bm.m[0][1] = Gu;
        Bm.m[1][0] = Gw;
        Bm.m[1][1] = B[2];
        d.v[0] = f[1];
        d.v[1] = f[2];
        elp.push_diag(i, cv, Bm, av, d);
    }
    // rms of the three normalised residuals (tango/solver.py:464-470): the first solve's matrix decides
    if (attempt == 0) {
        double tot = 0.0;
        g.template max_and_sum<3>(mx, ss);
#pragma unroll
        for (int s = 0; s < 3; ++s) tot += ss[s] / (mx[s] * mx[s]);
        rms = sqrt(1.0 / (3.0 * (double)n) * tot);
    }
    double tn[P];
    Vec<2> tp[P];
    eln.finish(g, tn);
    elp.finish(g, tp);
    if constexpr (!TREAD) {
#pragma unroll
        for (int i = 0; i < P; ++i) {
            xn[i] = tn[i];
            xp[i] = tp[i];
        }
        singular = eln.singular | elp.singular;
        break;
    } else {
#pragma unroll
        for (int i = 0; i < P; ++i) {
            xn[i] = settled ? xn[i] : tn[i];
            xp[i].v[0] = settled ? xp[i].v[0] : tp[i].v[0];
            xp[i].v[1] = settled ? xp[i].v[1] : tp[i].v[1];
        }
        singular |= settled ? 0 : (eln.singular | elp.singular);
        if (!Q.tread_lightly) break;
        // tango/solver.py:374-402, :564-595: a change of maxFractionalChange or more anywhere in the inner part of ANY profile
        int far = 0;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            const bool in = j0 + i < Q.tread_inner_count;
            far |= in && (fabs((tn[i] - u[0][i + 1]) / u[0][i + 1]) >= Q.tread_max_change);
            far |= in && (fabs((tp[i].v[0] - u[1][i + 1]) / u[1][i + 1]) >= Q.tread_max_change);
            far |= in && (fabs((tp[i].v[1] - u[2][i + 1]) / u[2][i + 1]) >= Q.tread_max_change);
        }
        far = g.any2(far);
        if (!settled && far && attempt >= TG_TREAD_MAX_HALVINGS) tread_limit = true;
        settled = settled || !far || attempt >= TG_TREAD_MAX_HALVINGS;
        if (G::kBlock ? settled : !__any(!settled)) break;
        dtau = settled ? dtau : dtau / 2.0;
        rc = Recip(Q.dx, dtau);
    }
    }   // attempts
    double xi[P], xe[P];
    int bad = 0;
#pragma unroll
    for (int i = 0; i < P; ++i) {
        xi[i] = xp[i].v[0];
        xe[i] = xp[i].v[1];
        if (FULL || j0 + i < n) bad |= !(isfinite(xn[i]) && xn[i] >= 0.0 && isfinite(xi[i]) && xi[i] >= 0.0 && isfinite(xe[i]) && xe[i] >= 0.0);
    }
    if constexpr (COOP) {
        wt.put(0, xn); wt.put(1, xi); wt.put(2, xe);
        Tile::sync();
        wt.flush(0, S_.u[0] + woff); wt.flush(1, S_.u[1] + woff); wt.flush(2, S_.u[2] + woff);
    } else {
        store_chunk<P, false>(S_.u[0] + off, n, j0, xn);
        store_chunk<P, false>(S_.u[1] + off, n, j0, xi);
        store_chunk<P, false>(S_.u[2] + off, n, j0, xe);
    }
    bad = g.any2(bad | (singular << 1));
    if (t == 0) {
        int ns = st;
        if (!Q.frozen_turb) {
            if (S_.err && it < S_.err_stride) S_.err[b * S_.err_stride + it] = rms;
            S_.iters[b] = it + 1;
            ns |= TG_ST_SEEDED;
            if (!Q.fixed_count && rms < Q.tol) ns |= TG_ST_CONVERGED;
        }
        if (bad & 1) ns |= TG_ST_INVALID;
        if (bad & 2) ns |= TG_ST_SINGULAR;
        if (tread_limit) ns |= TG_ST_TREAD_LIMIT;
        S_.status[b] = ns;
    }
}

// ------------------------------------------------------------------------------------------
// dispatch on the number of lanes per profile
// ------------------------------------------------------------------------------------------
template <template <int, class> class Launcher, class... Args>
int dispatch_block(int64_t nb, int n, int max_lanes, hipStream_t s, Args... args) {
    const int lanes = pow2_at_least((n + PB - 1) / PB);
    if (lanes > max_lanes) return fail(TG_ERR_UNSUPPORTED, "grid too large for the block-tridiagonal kernels%s");
    if (lanes <= 8) return Launcher<PB, WaveGroup<8>>::go(nb, n, s, args...);
    if (lanes == 16) return Launcher<PB, WaveGroup<16>>::go(nb, n, s, args...);
    if (lanes == 32) return Launcher<PB, WaveGroup<32>>::go(nb, n, s, args...);
    if (lanes == 64) return Launcher<PB, WaveGroup<64>>::go(nb, n, s, args...);
    if (lanes == 128) return Launcher<PB, BlockGroup<128>>::go(nb, n, s, args...);
    if (lanes == 256) return Launcher<PB, BlockGroup<256>>::go(nb, n, s, args...);
    if (lanes == 512) return Launcher<PB, BlockGroup<512>>::go(nb, n, s, args...);
    return Launcher<PB, BlockGroup<1024>>::go(nb, n, s, args...);
}

template <int P, class G>
struct PairLauncher {
    static int go(int64_t nb, int n, hipStream_t s, PairArgs a, int32_t *status) {
        int grid, block;
        size_t lds;
        geometry<G>(nb, grid, block, lds, RowPack<2>::N);
        hipLaunchKernelGGL((pair_solve_kernel<P, G>), dim3(grid), dim3(block), lds, s, a, nb, n, status);
        return launch_check();
    }
};

template <int P, class G>
struct Block3Launcher {
    static int go(int64_t nb, int n, hipStream_t s, const double *Cb, const double *Bb, const double *Ab, const double *d,
                  double *x, int32_t *status) {
        int grid, block;
        size_t lds;
        geometry<G>(nb, grid, block, lds, RowPack<3>::N);
        hipLaunchKernelGGL((block3_solve_kernel<P, G>), dim3(grid), dim3(block), lds, s, Cb, Bb, Ab, d, x, nb, n, status);
        return launch_check();
    }
};

template <int P, class G>
struct Picard3fLauncher {
    static int go(int64_t nb, int n, hipStream_t s, const tg_state_3f *st, const tg_params_3f *q) {
        int grid, block;
        size_t lds;
        geometry<G>(nb, grid, block, lds, kExch3f);
        uintptr_t al = 0;
        for (int k = 0; k < 3; ++k)
            al |= reinterpret_cast<uintptr_t>(st->u[k]) | reinterpret_cast<uintptr_t>(st->um1[k]) | reinterpret_cast<uintptr_t>(st->uE[k]) |
                  reinterpret_cast<uintptr_t>(st->DE[k]) | reinterpret_cast<uintptr_t>(st->cE[k]) |
                  (q->flux_model == TG_FLUX_EXTERNAL ? reinterpret_cast<uintptr_t>(q->flux[k]) : 0);
        const size_t tiles = (size_t)(block / 64) * kTiles3f * WaveTile<(P == 8) ? 8 : 4>::kDoubles * sizeof(double);
        const size_t cap = max_lds_bytes();
        if (lds > cap) return fail(TG_ERR_UNSUPPORTED, "grid too large for the LDS of this device%s");
        const bool full = (n == P * G::T) && (al & 15) == 0 && (P == 4 || P == 8) && (G::kBlock || G::T == 64) && lds + tiles <= cap;
        if (full) lds += tiles;
        if (q->flux_model == TG_FLUX_EXTERNAL) {
            if (full) hipLaunchKernelGGL((picard_3f_kernel<P, G, true, false, true>), dim3(grid), dim3(block), lds, s, *st, *q);
            else hipLaunchKernelGGL((picard_3f_kernel<P, G, false, false, true>), dim3(grid), dim3(block), lds, s, *st, *q);
        } else if (q->pseudo_time) {
            if (full) hipLaunchKernelGGL((picard_3f_kernel<P, G, true, true, false>), dim3(grid), dim3(block), lds, s, *st, *q);
            else hipLaunchKernelGGL((picard_3f_kernel<P, G, false, true, false>), dim3(grid), dim3(block), lds, s, *st, *q);
        } else {
            if (full) hipLaunchKernelGGL((picard_3f_kernel<P, G, true, false, false>), dim3(grid), dim3(block), lds, s, *st, *q);
            else hipLaunchKernelGGL((picard_3f_kernel<P, G, false, false, false>), dim3(grid), dim3(block), lds, s, *st, *q);
        }
        return launch_check();
    }
};

}  // namespace

extern "C" {

int tg_pair_solve(const double *AU, const double *BU, const double *CU, const double *DU, const double *GW, const double *AW,
                  const double *BW, const double *CW, const double *DW, const double *GU, double *U, double *W, int64_t nb,
                  int64_t n, int32_t *status, tg_stream stream) {
    if (int e = check_dims(nb, n, 1)) return e;
    if (!AU || !BU || !CU || !DU || !GW || !AW || !BW || !CW || !DW || !GU || !U || !W) return fail(TG_ERR_ARG, "NULL array%s");
    if (nb == 0) return TG_OK;
    PairArgs a{AU, BU, CU, DU, GW, AW, BW, CW, DW, GU, U, W};
    return dispatch_block<PairLauncher>(nb, (int)n, 1024, S(stream), a, status);
}

int tg_block3_solve(const double *Cb, const double *Bb, const double *Ab, const double *d, double *x, int64_t nb, int64_t n,
                    int32_t *status, tg_stream stream) {
    if (int e = check_dims(nb, n, 1)) return e;
    if (!Cb || !Bb || !Ab || !d || !x) return fail(TG_ERR_ARG, "NULL array%s");
    if (nb == 0) return TG_OK;
    return dispatch_block<Block3Launcher>(nb, (int)n, 512, S(stream), Cb, Bb, Ab, d, x, status);
}

int tg_picard_3f_step(const tg_state_3f *st, const tg_params_3f *q, tg_stream stream) {
    if (!st || !q) return fail(TG_ERR_ARG, "NULL state or params%s");
    if (int e = check_dims(st->nb, st->n, 3)) return e;
    if (q->flux_model != TG_FLUX_SHESTAKOV && q->flux_model != TG_FLUX_EXTERNAL) return fail(TG_ERR_ARG, "unknown flux model%s");
    const bool ext = q->flux_model == TG_FLUX_EXTERNAL;
    for (int s = 0; s < 3; ++s) {
        if (!st->u[s] || !st->um1[s] || !st->uE[s] || !st->DE[s] || !st->cE[s]) return fail(TG_ERR_ARG, "NULL state array%s");
        if (!q->UL[s]) return fail(TG_ERR_ARG, "UL is required for the three fields%s");
        if (!ext && !q->H7[s]) return fail(TG_ERR_ARG, "H7 is required for the three fields%s");
        if (ext) {
            if (!q->frozen_turb && !q->flux[s]) return fail(TG_ERR_ARG, "outside inputs: a flux array is required for the three fields%s");
            const int64_t strides[3] = {q->h6_stride[s], q->h7_stride[s], q->h8_stride[s]};
            for (int k = 0; k < 3; ++k)
                if (strides[k] != 0 && strides[k] != st->n) return fail(TG_ERR_ARG, "H6, H7, H8 strides must be 0 (one array for all profiles) or n%s");
        }
    }
    if (ext && q->H8[0]) return fail(TG_ERR_ARG, "the first field is solved on its own: it has no H8%s");
    if (ext && q->pseudo_time) return fail(TG_ERR_UNSUPPORTED, "pseudo-timestep control is available with the built-in closure only%s");
    if (!st->status || !st->iters || (!ext && !q->nu0)) return fail(TG_ERR_ARG, "NULL status, iters or nu0%s");
    if (st->err && st->err_stride <= 0) return fail(TG_ERR_ARG, "err_stride must be positive%s");
    if ((q->tread_lightly || q->tread_initial_steps > 0) && !q->pseudo_time)
        return fail(TG_ERR_ARG, "tread_lightly and tread_initial_steps need pseudo_time (the previous iterates in the u_{m-1} slots)%s");
    if (q->pseudo_time && q->frozen_turb) return fail(TG_ERR_UNSUPPORTED, "pseudo-timestep control and inner iterations do not combine%s");
    if (q->tread_lightly && (q->tread_inner_count < 0 || q->tread_inner_count > st->n || !(q->tread_max_change > 0.0)))
        return fail(TG_ERR_ARG, "tread_lightly needs 0 <= tread_inner_count <= n and tread_max_change > 0%s");
    if (q->tread_initial_steps < 0 || (q->tread_initial_steps > 0 && !(q->tread_initial_dtau > 0.0)))
        return fail(TG_ERR_ARG, "tread_initial_steps needs tread_initial_dtau > 0%s");
    if (st->nb == 0) return TG_OK;
    return dispatch_block<Picard3fLauncher>(st->nb, (int)st->n, 1024, S(stream), st, q);
}

}  // extern "C"
