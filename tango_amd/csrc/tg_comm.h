// Lane-group communication policies for the profile solvers (gfx950, 64-wide wavefronts).
//
// A profile is handled by a *group* of T cooperating lanes, lane t owning P contiguous grid
// points.  Two group shapes exist:
//   WaveGroup<LPP>  : T = LPP <= 64 lanes inside one wavefront (64/LPP profiles per wave); all
//                     exchanges are cross-lane shuffles, no LDS and no barriers.
//   BlockGroup<T>   : T = blockDim.x > 64 lanes = one workgroup per profile; exchanges go through
//                     LDS with workgroup barriers.
// Both expose the same interface so that the assembly and solve code is written once.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tg {

// 1 / d and n / d without the scaling and fix-up steps of the IEEE sequence (12 issue slots -> 5 and 8): the
// hardware estimate refined by two Newton steps, and one residual correction for the quotient.  The quotient is the
// arithmetic of the compiler's own division sequence, so it is the correctly rounded one as long as d and the
// quotient are normal numbers far from the exponent limits, which holds for every divisor of the transport problem;
// the reciprocal without the last correction is within an ulp of it.  A zero, infinite or NaN divisor gives NaN or
// infinity where the true division would, so validity checks downstream see the same thing.
// MODE 0 is the division itself (standalone kernels: arbitrary operands); MODE 1 the short forms (reciprocal within
// an ulp); MODE 2 the short forms with the reciprocal taken as the quotient 1/d, i.e. correctly rounded (grids of more
// than 512 points, whose condition numbers above 1e9 leave no room for an extra ulp in the pivots).
enum { kDivIeee = 0, kDivFast = 1, kDivRounded = 2 };
template <int MODE>
__device__ __forceinline__ double quot(double n, double d);
template <int MODE>
__device__ __forceinline__ double recip(double d) {
    if constexpr (MODE == kDivFast) {
        double x = __builtin_amdgcn_rcp(d);
        double e = __builtin_fma(-d, x, 1.0);
        x = __builtin_fma(x, e, x);
        e = __builtin_fma(-d, x, 1.0);
        return __builtin_fma(x, e, x);
    } else if constexpr (MODE == kDivRounded) {
        return quot<kDivFast>(1.0, d);
    } else {
        return 1.0 / d;
    }
}
template <int MODE>
__device__ __forceinline__ double quot(double n, double d) {
    if constexpr (MODE != kDivIeee) {
        const double x = recip<kDivFast>(d);
        const double q = n * x;
        return __builtin_fma(__builtin_fma(-d, q, n), x, q);
    } else {
        return n / d;
    }
}

// n / d for a divisor whose correctly rounded reciprocal `inv` = 1/d is at hand (a launch constant, formed once by a
// true division): the product and one residual correction, the last two steps of the division sequence above with the
// exact reciprocal in place of the refined estimate, 3 issue slots instead of 8.  The product q is within 1.5 ulp of
// n/d (half an ulp of inv, half an ulp of the product), the residual n - d q is exact or rounded in its 54th bit, and
// the result is the correct rounding of n/d + delta with |delta| of a few 2^-53 ulp: the correctly rounded quotient
// unless n/d lies that close to the midpoint of two doubles (the nearest a quotient of two doubles can come to one is
// 2^-54 ulp).  No such operand pair
// turned up among 300 000 random and hard-to-round ones (mantissas of all ones, next to a power of two) run through this
// sequence in exact rational arithmetic (tests/test_host_logic.py::test_short_quotient_rounds_like_a_division), and the
// theta golden vectors, whose values depend on this rounding to the bit, reproduce.  Normal operands far from the
// exponent limits only.
__device__ __forceinline__ double quot_by(double n, double d, double inv) {
    const double q = n * inv;
    return __builtin_fma(__builtin_fma(-d, q, n), inv, q);
}

// Neighbour exchange over the whole wavefront by data-parallel-primitive moves (no trip through the LDS crossbar):
// lane t receives the value of lane t-1 (wave_shr:1) or t+1 (wave_shl:1); the lane at the end keeps its own value.
__device__ __forceinline__ double dpp_from_lane_below(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_from_lane_above(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
// The same with zero for the lane at the end (bound_ctrl: a lane without a source receives 0): no select afterwards.
__device__ __forceinline__ double dpp_from_lane_below_or_zero(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x138, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x138, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_from_lane_above_or_zero(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, 0x130, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, 0x130, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}

// Exchange between the two halves of a wavefront: lane t receives the value of lane t ^ 32 (one v_permlane32_swap per
// dword and a select, no trip through the LDS crossbar).
__device__ __forceinline__ double swap_halves(double v, bool upper) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    // element 0: lanes 32-63 hold the lower half's values; element 1: lanes 0-31 hold the upper half's values
    return __hiloint2double((int)(upper ? b[0] : b[1]), (int)(upper ? a[0] : a[1]));
}

// Marks the point by which a loaded value must have arrived: the compiler places the wait for its load here and none
// later.  Used ahead of a kernel's first store, because a wait for a load also waits for every store issued before it
// (loads and stores share one counter and leave it in order).
__device__ __forceinline__ void landed(double &x) { asm volatile("" : "+v"(x)); }
__device__ __forceinline__ void landed(double2 &x) {
    landed(x.x);
    landed(x.y);
}

// Index of this wavefront within its workgroup.  It is the same in every lane; read through the first lane so that the
// compiler knows, and keeps what is derived from it (row offsets, tile addresses, per-profile parameters) in scalar registers.
__device__ __forceinline__ int wave_in_block() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

template <int LPP>
struct WaveGroup {
    static constexpr int T = LPP;
    static constexpr bool kBlock = false;
    int t_;
    __device__ explicit WaveGroup(double * /*lds*/) : t_(threadIdx.x & (LPP - 1)) {}
    __device__ int t() const { return t_; }
    // number of profiles handled by one workgroup of `threads` threads
    static constexpr int profiles_per_block(int threads) { return threads / LPP; }
    // a whole wavefront per profile: the index is the same in every lane, and saying so keeps everything derived from it
    // (status, per-profile parameters, row offsets) in scalar registers
    __device__ int64_t profile_in_block() const {
        if constexpr (LPP == 64) return wave_in_block();
        else return threadIdx.x / LPP;
    }

    // value held by lane t-h (fill when there is none)
    __device__ double from_left(double v, int h, double fill) const {
        // distance 1: a DPP move over the whole wavefront, also where several profiles share it (the first lane of a
        // group then receives the last lane of the group before it, and drops it)
        double r;
        if (__builtin_constant_p(h) && h == 1) {
            if (LPP == 64 && __builtin_constant_p(fill) && fill == 0.0) return dpp_from_lane_below_or_zero(v);
            r = dpp_from_lane_below(v);
        } else {
            r = __shfl_up(v, (unsigned)h, LPP);
        }
        return (t_ >= h) ? r : fill;
    }
    __device__ double from_right(double v, int h, double fill) const {
        double r;
        if (__builtin_constant_p(h) && h == 1) {
            if (LPP == 64 && __builtin_constant_p(fill) && fill == 0.0) return dpp_from_lane_above_or_zero(v);
            r = dpp_from_lane_above(v);
        } else {
            r = __shfl_down(v, (unsigned)h, LPP);
        }
        return (t_ + h < LPP) ? r : fill;
    }
    // CNT values from the lanes t-h (l) and t+h (r), zeros where there is no such lane
    template <int CNT>
    __device__ void exchange(const double (&v)[CNT], int h, double (&l)[CNT], double (&r)[CNT]) const {
#pragma unroll
        for (int q = 0; q < CNT; ++q) {
            l[q] = from_left(v[q], h, 0.0);
            r[q] = from_right(v[q], h, 0.0);
        }
    }
    // the same without fill values: a lane without such a neighbour receives its own values (for callers that
    // multiply them by an exact zero, like the cyclic reductions)
    template <int CNT>
    __device__ void exchange_nofill(const double (&v)[CNT], int h, double (&l)[CNT], double (&r)[CNT]) const {
#pragma unroll
        for (int q = 0; q < CNT; ++q) {
            l[q] = __shfl_up(v[q], (unsigned)h, LPP);
            r[q] = __shfl_down(v[q], (unsigned)h, LPP);
        }
    }
    __device__ double sum(double v) const {
#pragma unroll
        for (int m = LPP >> 1; m > 0; m >>= 1) v += __shfl_xor(v, m, LPP);
        return v;
    }
    __device__ double max(double v) const {
#pragma unroll
        for (int m = LPP >> 1; m > 0; m >>= 1) v = fmax(v, __shfl_xor(v, m, LPP));
        return v;
    }
    // OR over the group of the two flag bits 0 and 1 of `pred` (what the callers pass: invalid | singular << 1 and the
    // like); the other bits are dropped in every group shape.  A whole wavefront combines the two bits by two ballots
    // instead of six shuffle steps.
    __device__ int any2(int pred) const {
        pred &= 3;
        if constexpr (LPP == 64) {
            return (__ballot(pred & 1) != 0 ? 1 : 0) | (__ballot(pred & 2) != 0 ? 2 : 0);
        } else {
#pragma unroll
            for (int m = LPP >> 1; m > 0; m >>= 1) pred |= __shfl_xor(pred, m, LPP);
            return pred;
        }
    }
    // CNT maxima and CNT sums over the group in one go (one pair of barriers for a workgroup-sized group)
    template <int CNT>
    __device__ void max_and_sum(double (&mx)[CNT], double (&sm)[CNT]) const {
#pragma unroll
        for (int q = 0; q < CNT; ++q) {
            mx[q] = max(mx[q]);
            sm[q] = sum(sm[q]);
        }
    }
    // Parallel cyclic reduction of the unit-diagonal system s_t y_{t-1} + y_t + p_t y_{t+1} = r_t.
    // dl_t = 1 + s_t + p_t (the row's diagonal-dominance slack) is carried along and every new diagonal is
    // formed as slack - s - p: for the M-matrices of the transport problem (s, p <= 0 <= dl) all those sums
    // have terms of one sign, so no cancellation eats the small slack as 1 - pm*s - sp*p would.
    //
    // No fill values for the lanes without a neighbour at distance h: the first lane has no left coupling (s = 0,
    // enforced on entry) and by induction s = 0 in every lane t < h at the level of distance h (s2 = -sm s with
    // sm taken from a lane below h), likewise p = 0 in the lanes t >= LPP - h.  Whatever those lanes fetch (the
    // shuffles hand them their own, finite, values) is multiplied by an exact zero.
    template <int FAST = 0>
    __device__ double pcr(double s, double p, double dl, double r, int &singular) const {
        s = (t_ == 0) ? 0.0 : s;
        p = (t_ == LPP - 1) ? 0.0 : p;
#pragma unroll
        for (int h = 1; h < LPP; h <<= 1) {
            double sm, rm, dm, pp, rp, dp;
            if (LPP == 64 && h == 1) {      // nearest neighbours over a whole wavefront: DPP moves (not where profiles share a
                                            // wavefront: an end lane would multiply ANOTHER profile's value by its zero, and
                                            // that value may be NaN)
                sm = dpp_from_lane_below(s), rm = dpp_from_lane_below(r), dm = dpp_from_lane_below(dl);
                pp = dpp_from_lane_above(p), rp = dpp_from_lane_above(r), dp = dpp_from_lane_above(dl);
            } else {
                sm = __shfl_up(s, (unsigned)h, LPP), rm = __shfl_up(r, (unsigned)h, LPP), dm = __shfl_up(dl, (unsigned)h, LPP);
                pp = __shfl_down(p, (unsigned)h, LPP), rp = __shfl_down(r, (unsigned)h, LPP), dp = __shfl_down(dl, (unsigned)h, LPP);
            }
            const double d2 = dl - s * dm - p * dp;
            const double s2 = -(sm * s);
            const double p2 = -(pp * p);
            const double g = d2 - s2 - p2;
            singular |= (g == 0.0);
            const double inv = recip<FAST>(g);
            r = (r - rm * s - rp * p) * inv;
            s = s2 * inv;
            p = p2 * inv;
            dl = d2 * inv;
            if (LPP == 64 && h == 16) break;      // the last level of a whole wavefront is done below
        }
        if constexpr (LPP == 64) {
            // Distance 32: a lane of the lower half has no left neighbour (s = 0 exactly, see above) and a lane of the upper
            // half no right one (p = 0), so the value of lane t ^ 32 serves as both neighbours (the absent one is
            // multiplied by zero), and the new couplings s2 = -(sm s), p2 = -(pp p) are zero: sm and pp come from lanes
            // whose own s and p are zero.  The results are those of the general level, bit for bit.
            const bool upper = t_ >= 32;
            const double rx = swap_halves(r, upper), dx = swap_halves(dl, upper);
            const double d2 = dl - s * dx - p * dx;
            singular |= (d2 == 0.0);
            const double inv = recip<FAST>(d2);
            r = (r - rx * s - rx * p) * inv;
        }
        return r;
    }
};

// One workgroup of T_ threads per profile.  `lds` must hold at least 8*T_ doubles (CNT*T_ for exchange<CNT>).
template <int T_>
struct BlockGroup {
    static constexpr int T = T_;
    static constexpr bool kBlock = true;
    double *lds_;
    int t_;
    __device__ explicit BlockGroup(double *lds) : lds_(lds), t_(threadIdx.x) {}
    __device__ int t() const { return t_; }
    static constexpr int profiles_per_block(int) { return 1; }
    __device__ int64_t profile_in_block() const { return 0; }

    __device__ double from_left(double v, int h, double fill) const {
        lds_[t_] = v;
        __syncthreads();
        double r = (t_ >= h) ? lds_[t_ - h] : fill;
        __syncthreads();
        return r;
    }
    __device__ double from_right(double v, int h, double fill) const {
        lds_[t_] = v;
        __syncthreads();
        double r = (t_ + h < T_) ? lds_[t_ + h] : fill;
        __syncthreads();
        return r;
    }
    // CNT values from the threads t-h (l) and t+h (r), zeros where there is none; needs CNT*T_ doubles of LDS
    template <int CNT>
    __device__ void exchange(const double (&v)[CNT], int h, double (&l)[CNT], double (&r)[CNT]) const {
#pragma unroll
        for (int q = 0; q < CNT; ++q) lds_[q * T_ + t_] = v[q];
        __syncthreads();
        // every thread reads (a thread at the end reads the row T_ places away and drops it): selects, not branches
        const int tl = (t_ - h) & (T_ - 1), tr = (t_ + h) & (T_ - 1);
        const bool hasl = t_ >= h, hasr = t_ + h < T_;
#pragma unroll
        for (int q = 0; q < CNT; ++q) {
            const double a = lds_[q * T_ + tl], c = lds_[q * T_ + tr];
            l[q] = hasl ? a : 0.0;
            r[q] = hasr ? c : 0.0;
        }
        __syncthreads();
    }
    // the same without fill values: a thread without such a neighbour receives the values of the thread T_ places
    // further on or back (for callers that multiply them by an exact zero, like the cyclic reductions)
    template <int CNT>
    __device__ void exchange_nofill(const double (&v)[CNT], int h, double (&l)[CNT], double (&r)[CNT]) const {
#pragma unroll
        for (int q = 0; q < CNT; ++q) lds_[q * T_ + t_] = v[q];
        __syncthreads();
        const int tl = (t_ - h) & (T_ - 1), tr = (t_ + h) & (T_ - 1);
#pragma unroll
        for (int q = 0; q < CNT; ++q) {
            l[q] = lds_[q * T_ + tl];
            r[q] = lds_[q * T_ + tr];
        }
        __syncthreads();
    }
    __device__ double sum(double v) const {
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
        if ((t_ & 63) == 0) lds_[t_ >> 6] = v;
        __syncthreads();
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < T_ / 64; ++w) tot += lds_[w];
        __syncthreads();
        return tot;
    }
    __device__ double max(double v) const {
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) v = fmax(v, __shfl_xor(v, m, 64));
        if ((t_ & 63) == 0) lds_[t_ >> 6] = v;
        __syncthreads();
        double tot = lds_[0];
#pragma unroll
        for (int w = 1; w < T_ / 64; ++w) tot = fmax(tot, lds_[w]);
        __syncthreads();
        return tot;
    }
    // OR of the two flag bits 0 and 1 of `pred` over the workgroup, the other bits dropped.  (Not __syncthreads_or: that
    // one reads the workgroup's size from memory, and a wait for that load also waits for every store the wavefront has issued.)
    __device__ int any2(int pred) const {
        const int mine = (__ballot(pred & 1) != 0 ? 1 : 0) | (__ballot(pred & 2) != 0 ? 2 : 0);
        int *flags = reinterpret_cast<int *>(lds_);
        if ((t_ & 63) == 0) flags[t_ >> 6] = mine;
        __syncthreads();
        int all = 0;
#pragma unroll
        for (int w = 0; w < T_ / 64; ++w) all |= flags[w];
        __syncthreads();
        return all;
    }
    // CNT maxima and CNT sums over the workgroup with one pair of barriers; needs 2*CNT*(T_/64) doubles of LDS.
    // The per-wavefront partial results are combined in wavefront order, as sum() and max() combine them.
    template <int CNT>
    __device__ void max_and_sum(double (&mx)[CNT], double (&sm)[CNT]) const {
        constexpr int W = T_ / 64;
#pragma unroll
        for (int q = 0; q < CNT; ++q) {
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) {
                mx[q] = fmax(mx[q], __shfl_xor(mx[q], m, 64));
                sm[q] += __shfl_xor(sm[q], m, 64);
            }
        }
        if ((t_ & 63) == 0) {
#pragma unroll
            for (int q = 0; q < CNT; ++q) {
                lds_[(2 * q) * W + (t_ >> 6)] = mx[q];
                lds_[(2 * q + 1) * W + (t_ >> 6)] = sm[q];
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < CNT; ++q) {
            double m = lds_[(2 * q) * W], tot = 0.0;
#pragma unroll
            for (int w = 1; w < W; ++w) m = fmax(m, lds_[(2 * q) * W + w]);
#pragma unroll
            for (int w = 0; w < W; ++w) tot += lds_[(2 * q + 1) * W + w];
            mx[q] = m;
            sm[q] = tot;
        }
        __syncthreads();
    }

    template <int FAST = 0>
    __device__ double pcr(double s, double p, double dl, double r, int &singular) const {
        // double-buffered rows in LDS: buffer k holds s,p,dl,r at lds_[k*4*T_ + {0,1,2,3}*T_ + t]
        // no special cases at the ends: a thread without a neighbour at distance h has an exactly zero coupling on
        // that side (see WaveGroup::pcr) and reads the row T_ places away instead, whose values it multiplies by zero
        s = (t_ == 0) ? 0.0 : s;
        p = (t_ == T_ - 1) ? 0.0 : p;
        int cur = 0;
#pragma unroll
        for (int h = 1; h < T_; h <<= 1) {
            double *bs = lds_ + cur * 4 * T_, *bp = bs + T_, *bd = bp + T_, *br = bd + T_;
            bs[t_] = s;
            bp[t_] = p;
            bd[t_] = dl;
            br[t_] = r;
            __syncthreads();
            const int tl = (t_ - h) & (T_ - 1), tr = (t_ + h) & (T_ - 1);
            const double sm = bs[tl], dm = bd[tl], rm = br[tl];
            const double pp = bp[tr], dp = bd[tr], rp = br[tr];
            const double d2 = dl - s * dm - p * dp;
            const double s2 = -(sm * s);
            const double p2 = -(pp * p);
            const double g = d2 - s2 - p2;
            singular |= (g == 0.0);
            const double inv = recip<FAST>(g);
            r = (r - rm * s - rp * p) * inv;
            s = s2 * inv;
            p = p2 * inv;
            dl = d2 * inv;
            cur ^= 1;
        }
        __syncthreads();
        return r;
    }
};

}  // namespace tg
