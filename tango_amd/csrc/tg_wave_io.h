// Coalesced row I/O for a wavefront that owns 64 P consecutive doubles of a row (lane t holds points P t .. P t + P-1;
// P = 8: 4 KiB per array, P = 4: 2 KiB).
//
// A wavefront reading its points lane by lane issues P/2 16-byte loads per lane at a lane stride of 8 P bytes:
// every instruction touches the whole segment and relies on the caches to merge the pieces; measured, the fused
// one-field step then streams at ~4.3 TB/s while the same kernel with fully coalesced nontemporal accesses
// reaches the streaming ceiling of the device (~5.5 TB/s).  So the rows are moved coalesced (instruction i,
// lane l <-> 16-byte element 64 i + l) and transposed through an LDS tile that is private to the wavefront: no
// workgroup barrier, only program order inside the wave.
//
// Tile layout, with KP = P/2 16-byte pieces per lane and AW = 64/KP lanes per 1 KiB piece: element
// e = 64 i + KP a + k (i: 1 KiB piece, a < AW, k < KP) lives in slot 64 i + AW k + (a ^ X(k)), X(k) = (10 k) & 15 for
// P = 8 and 4 k for P = 4.  With these swizzles (found by exhaustive search over the gfx950 lane groupings) all
// four access shapes are free of bank conflicts: the coalesced side (lane l <-> element 64 i + l) as 8-lane
// ds_write_b128 groups and 16-lane ds_read_b128 groups, and the per-lane side (lane L <-> elements KP L + k).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tg {

template <int P>
struct WaveTile {
    static_assert(P == 8 || P == 4, "tiles exist for 8 and 4 points per lane");
    static constexpr int KP = P / 2;          // 16-byte pieces per lane = 1 KiB pieces per tile
    static constexpr int AW = 64 / KP;
    static constexpr int kDoubles = 64 * P;   // one tile
    static constexpr int kSlots = 64 * KP;    // double2 slots per tile
    static __device__ __forceinline__ constexpr int X(int k) { return P == 8 ? ((10 * k) & 15) : 4 * k; }

    double2 *tile_;   // this wavefront's staging area (as many tiles as the caller reserved)
    int lane_;
    int cslot_;       // slot of element 64 i + lane within piece i (add 64 i)
    int pbase_;       // slot of element KP lane + k without the k terms

    __device__ WaveTile(double *wave_lds) : tile_(reinterpret_cast<double2 *>(wave_lds)), lane_(threadIdx.x & 63) {
        const int k = lane_ % KP, a = lane_ / KP;
        cslot_ = AW * k + (a ^ X(k));
        pbase_ = 64 * (lane_ / AW);
    }
    __device__ __forceinline__ int pslot(int k) const { return pbase_ + AW * k + ((lane_ % AW) ^ X(k)); }

    // order LDS traffic of the lanes of this wavefront (program order inside a wave; the fences only stop the
    // compiler from moving LDS accesses across)
    static __device__ __forceinline__ void sync() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    // coalesced nontemporal fetch of one row segment into registers (KP x 16 bytes per lane)
    __device__ __forceinline__ void fetch(const double *__restrict__ src, double2 (&raw)[KP]) const {
        const double2 *q = reinterpret_cast<const double2 *>(src) + lane_;
#pragma unroll
        for (int i = 0; i < KP; ++i) {
            raw[i].x = __builtin_nontemporal_load(&q[64 * i].x);
            raw[i].y = __builtin_nontemporal_load(&q[64 * i].y);
        }
    }
    // registers (coalesced order) -> tile `buf`
    __device__ __forceinline__ void stage(int buf, const double2 (&raw)[KP]) const {
#pragma unroll
        for (int i = 0; i < KP; ++i) tile_[kSlots * buf + 64 * i + cslot_] = raw[i];
    }
    // tile `buf` -> the P points of this lane
    __device__ __forceinline__ void take(int buf, double *v) const {
#pragma unroll
        for (int k = 0; k < KP; ++k) {
            const double2 w = tile_[kSlots * buf + pslot(k)];
            v[2 * k] = w.x;
            v[2 * k + 1] = w.y;
        }
    }
    // the P points of this lane -> tile `buf`
    __device__ __forceinline__ void put(int buf, const double *v) const {
#pragma unroll
        for (int k = 0; k < KP; ++k) tile_[kSlots * buf + pslot(k)] = make_double2(v[2 * k], v[2 * k + 1]);
    }
    // tile `buf` -> coalesced nontemporal store of the row segment.  PRED: only the elements whose owning
    // lane (element / KP) has its bit set in `owners` are written.
    template <bool PRED = false>
    __device__ __forceinline__ void flush(int buf, double *__restrict__ dst, uint64_t owners = ~0ull) const {
        double2 *q = reinterpret_cast<double2 *>(dst) + lane_;
#pragma unroll
        for (int i = 0; i < KP; ++i) {
            const double2 w = tile_[kSlots * buf + 64 * i + cslot_];
            if (!PRED || ((owners >> ((64 * i + lane_) / KP)) & 1)) {
                __builtin_nontemporal_store(w.x, &q[64 * i].x);
                __builtin_nontemporal_store(w.y, &q[64 * i].y);
            }
        }
    }
};

}  // namespace tg
