// Coalesced row I/O for a wavefront that owns 512 consecutive doubles (64 lanes x 8 points, 4 KiB).
//
// The solvers want lane t to hold points 8t..8t+7 (64 contiguous bytes), but a wavefront reading that way issues
// four 16-byte loads per lane at a 64-byte lane stride: every instruction touches all 4 KiB and relies on the
// caches to merge the quarters; measured, the fused step then streams at ~4.3 TB/s while the same kernel with
// fully coalesced nontemporal accesses reaches the streaming ceiling of the device (~5.5 TB/s).  So the rows are
// moved coalesced (instruction i, lane l <-> 16-byte element 64 i + l) and transposed through a 4 KiB LDS tile
// that is private to the wavefront: no workgroup barrier, only program order inside the wave.
//
// Tile layout: element e = 64 i + 4 a + k (i: 1 KiB piece, a: lane within a 16-lane quarter, k: 16-byte piece of
// a lane's 64 bytes) lives in slot 64 i + 16 k + (a ^ ((10 k) & 15)).  With this swizzle all four access
// shapes are free of bank conflicts on gfx950: the coalesced side (lane l <-> element 64 i + l) as 8-lane
// ds_write_b128 groups / 16-lane ds_read_b128 groups, and the per-lane side (lane L <-> elements 4 L + k).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tg {

constexpr int kWaveTileDoubles = 512;   // one array's worth: 4 KiB

struct WaveTile {
    double2 *tile_;   // this wavefront's staging area: NBUF tiles of 256 double2
    int lane_;
    int cslot_;       // slot of element 64 i + lane within piece i (add 64 i)
    int pbase_;       // slot of element 4 lane + k without the k terms: 64 (lane >> 4) + (lane & 15)

    __device__ WaveTile(double *wave_lds) : tile_(reinterpret_cast<double2 *>(wave_lds)), lane_(threadIdx.x & 63) {
        const int k = lane_ & 3, a = lane_ >> 2;
        cslot_ = 16 * k + (a ^ ((10 * k) & 15));
        pbase_ = 64 * (lane_ >> 4);
    }
    __device__ __forceinline__ int pslot(int k) const { return pbase_ + 16 * k + ((lane_ & 15) ^ ((10 * k) & 15)); }

    // order LDS traffic of the lanes of this wavefront (program order inside a wave; the fences only stop the
    // compiler from moving LDS accesses across)
    static __device__ __forceinline__ void sync() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    // coalesced nontemporal fetch of one 4 KiB row segment into registers (4 x 16 bytes per lane)
    __device__ __forceinline__ void fetch(const double *__restrict__ src, double2 (&raw)[4]) const {
        const double2 *q = reinterpret_cast<const double2 *>(src) + lane_;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            raw[i].x = __builtin_nontemporal_load(&q[64 * i].x);
            raw[i].y = __builtin_nontemporal_load(&q[64 * i].y);
        }
    }
    // registers (coalesced order) -> tile `buf`
    __device__ __forceinline__ void stage(int buf, const double2 (&raw)[4]) const {
#pragma unroll
        for (int i = 0; i < 4; ++i) tile_[256 * buf + 64 * i + cslot_] = raw[i];
    }
    // tile `buf` -> the 8 points of this lane
    __device__ __forceinline__ void take(int buf, double *v) const {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double2 w = tile_[256 * buf + pslot(k)];
            v[2 * k] = w.x;
            v[2 * k + 1] = w.y;
        }
    }
    // the 8 points of this lane -> tile `buf`
    __device__ __forceinline__ void put(int buf, const double *v) const {
#pragma unroll
        for (int k = 0; k < 4; ++k) tile_[256 * buf + pslot(k)] = make_double2(v[2 * k], v[2 * k + 1]);
    }
    // tile `buf` -> coalesced nontemporal store of the 4 KiB row segment.  PRED: only the elements whose owning
    // lane (element >> 2) has its bit set in `owners` are written.
    template <bool PRED = false>
    __device__ __forceinline__ void flush(int buf, double *__restrict__ dst, uint64_t owners = ~0ull) const {
        double2 *q = reinterpret_cast<double2 *>(dst) + lane_;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const double2 w = tile_[256 * buf + 64 * i + cslot_];
            if (!PRED || ((owners >> ((64 * i + lane_) >> 2)) & 1)) {
                __builtin_nontemporal_store(w.x, &q[64 * i].x);
                __builtin_nontemporal_store(w.y, &q[64 * i].y);
            }
        }
    }
};

}  // namespace tg
