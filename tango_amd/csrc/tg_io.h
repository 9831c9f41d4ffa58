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
template <int P, bool FULL>
__device__ __forceinline__ void load_chunk(const double *__restrict__ row, int n, int j0, double *v, double fill) {
    if constexpr (P % 2 == 0) {
        if (FULL || (j0 + P <= n && ((reinterpret_cast<uintptr_t>(row + j0) & 15) == 0))) {
#ifdef TG_EXPERIMENT_COALESCED
            const double2 *q = reinterpret_cast<const double2 *>(row) + (threadIdx.x & 63);
#define TG_QI(i) ((i) * 64)
#else
            const double2 *q = reinterpret_cast<const double2 *>(row + j0);
#define TG_QI(i) (i)
#endif
#pragma unroll
            for (int i = 0; i < P / 2; ++i) {
                double2 w = q[TG_QI(i)];
                v[2 * i] = w.x;
                v[2 * i + 1] = w.y;
            }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < P; ++i) v[i] = (j0 + i < n) ? row[j0 + i] : fill;
}

template <int P, bool FULL>
__device__ __forceinline__ void store_chunk(double *__restrict__ row, int n, int j0, const double *v) {
    if constexpr (P % 2 == 0) {
        if (FULL || (j0 + P <= n && ((reinterpret_cast<uintptr_t>(row + j0) & 15) == 0))) {
#ifdef TG_EXPERIMENT_COALESCED
            double2 *q = reinterpret_cast<double2 *>(row) + (threadIdx.x & 63);
#else
            double2 *q = reinterpret_cast<double2 *>(row + j0);
#endif
#pragma unroll
            for (int i = 0; i < P / 2; ++i) q[TG_QI(i)] = make_double2(v[2 * i], v[2 * i + 1]);
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < P; ++i)
        if (j0 + i < n) row[j0 + i] = v[i];
}

}  // namespace tg
