// Row I/O shared by the profile kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tg {

// ------------------------------------------------------------------------------------------
// Row I/O: lane t of a group reads / writes its P consecutive points of a profile row.
// ------------------------------------------------------------------------------------------
// FULL: the row is exactly T*P points and 16-byte aligned (n even), so every lane takes the
// vector path with no bounds checks.
// STREAM: nontemporal accesses.  They pay (4-5 % on a 5-read/4-write stream) only when a wavefront's access is
// fully coalesced; a lane reading its own 64 contiguous bytes in four 16-byte pieces relies on the cache to merge
// the pieces, and measured 1.8x slower with them, so the profile kernels leave STREAM off.
template <bool STREAM>
__device__ __forceinline__ double2 load2(const double2 *q) {
    if constexpr (STREAM) {
        double2 w;
        w.x = __builtin_nontemporal_load(&q->x);
        w.y = __builtin_nontemporal_load(&q->y);
        return w;
    } else {
        return *q;
    }
}
template <bool STREAM>
__device__ __forceinline__ void store2(double2 *q, double a, double b) {
    if constexpr (STREAM) {
        __builtin_nontemporal_store(a, &q->x);
        __builtin_nontemporal_store(b, &q->y);
    } else {
        *q = make_double2(a, b);
    }
}

template <int P, bool FULL, bool STREAM = false>
__device__ __forceinline__ void load_chunk(const double *__restrict__ row, int n, int j0, double *v, double fill) {
    if constexpr (P % 2 == 0) {
        if (FULL || (j0 + P <= n && ((reinterpret_cast<uintptr_t>(row + j0) & 15) == 0))) {
            const double2 *q = reinterpret_cast<const double2 *>(row + j0);
#pragma unroll
            for (int i = 0; i < P / 2; ++i) {
                const double2 w = load2<STREAM>(q + i);
                v[2 * i] = w.x;
                v[2 * i + 1] = w.y;
            }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < P; ++i) v[i] = (j0 + i < n) ? row[j0 + i] : fill;
}

template <int P, bool FULL, bool STREAM = false>
__device__ __forceinline__ void store_chunk(double *__restrict__ row, int n, int j0, const double *v) {
    if constexpr (P % 2 == 0) {
        if (FULL || (j0 + P <= n && ((reinterpret_cast<uintptr_t>(row + j0) & 15) == 0))) {
            double2 *q = reinterpret_cast<double2 *>(row + j0);
#pragma unroll
            for (int i = 0; i < P / 2; ++i) store2<STREAM>(q + i, v[2 * i], v[2 * i + 1]);
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < P; ++i)
        if (j0 + i < n) row[j0 + i] = v[i];
}

}  // namespace tg
