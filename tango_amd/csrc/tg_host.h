// Host-side helpers shared by the translation units of libtango_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/tango_hip.h"

extern thread_local char tg_err_buf[512];

// segmented form of the one-field step for long profiles (tango_seg.hip)
size_t tg_seg_work_bytes(int64_t nb, int64_t n);
size_t tg_seg_work_bytes_tread(int64_t nb, int64_t n);
bool tg_seg_applies(const tg_state_1f *st, const tg_params_1f *q);
int tg_seg_step(const tg_state_1f *st, const tg_params_1f *q, hipStream_t s);
// argument checks and the non-segmented launch of the one-field step (tango_hip.hip)
int tg_picard_1f_validate(const tg_state_1f *st, const tg_params_1f *q);
int tg_picard_1f_dispatch(const tg_state_1f *st, const tg_params_1f *q, hipStream_t s);

namespace {

inline int fail(int code, const char *fmt, const char *a = "") {
    snprintf(tg_err_buf, sizeof(tg_err_buf), fmt, a);
    return code;
}

#define TG_HIP(call)                                                        \
    do {                                                                    \
        hipError_t e_ = (call);                                             \
        if (e_ != hipSuccess) return fail(TG_ERR_HIP, #call ": %s", hipGetErrorString(e_)); \
    } while (0)

inline hipStream_t S(tg_stream s) { return reinterpret_cast<hipStream_t>(s); }

inline int launch_check() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(TG_ERR_HIP, "kernel launch: %s", hipGetErrorString(e));
    return TG_OK;
}

// dimensions every entry point accepts: the kernels index a profile with 32-bit integers
inline int check_dims(int64_t nb, int64_t n, int64_t nmin) {
    if (nb < 0 || n < nmin) return fail(TG_ERR_ARG, "bad dimensions (nb >= 0, n >= minimum for this operation)%s");
    if (n > (int64_t)1 << 30) return fail(TG_ERR_ARG, "n too large%s");
    return TG_OK;
}

// LDS bytes one workgroup may use on the current device (0 if the query fails)
inline size_t max_lds_bytes() {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) return 0;
    return (size_t)v;
}

inline int pow2_at_least(int v) {
    int p = 1;
    while (p < v) p <<= 1;
    return p;
}

// Launch geometry of a lane-group kernel: WaveGroup kernels use 256-thread workgroups holding 256/T
// profiles; BlockGroup kernels one workgroup of T threads per profile with lds_doubles_per_thread*T doubles.
template <class G>
inline void geometry(int64_t nb, int &grid, int &block, size_t &lds, int lds_doubles_per_thread = 8) {
    if (G::kBlock) {
        block = G::T;
        grid = (int)nb;
        lds = sizeof(double) * (size_t)lds_doubles_per_thread * G::T;
    } else {
        block = 256;
        const int ppb = 256 / G::T;
        grid = (int)((nb + ppb - 1) / ppb);
        lds = 0;
    }
}

}  // namespace
