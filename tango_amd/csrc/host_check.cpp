// Host-side checks of libtango_hip.so for a sanitizer build (python tango_amd/build.py --asan-host): every entry point is
// called with arguments its validation must refuse (NULL arrays, bad sizes, inconsistent options) and with empty
// ensembles, so that the host code in front of every launch runs under AddressSanitizer.  No GPU is needed: where a call
// gets as far as the HIP runtime on a machine without one it must come back with TG_ERR_HIP, not crash.
#include <stdio.h>
#include <string.h>

#include "../../include/tango_hip.h"

static int failures = 0;
#define EXPECT(cond)                                                   \
    do {                                                               \
        if (!(cond)) {                                                 \
            fprintf(stderr, "host_check: %s failed (line %d): %s\n", #cond, __LINE__, tg_last_error()); \
            ++failures;                                                \
        }                                                              \
    } while (0)

int main() {
    double dummy[8] = {0};
    int32_t istat[2] = {0};
    EXPECT(tg_version() >= 100);
    EXPECT(tg_device_count(nullptr) == TG_ERR_ARG);
    EXPECT(tg_malloc(nullptr, 8) == TG_ERR_ARG);
    EXPECT(tg_host_alloc(nullptr, 8) == TG_ERR_ARG);
    EXPECT(tg_stream_create(nullptr) == TG_ERR_ARG);
    EXPECT(tg_copy_d2d_2d(dummy, 8, dummy, 8, 16, 1, nullptr) == TG_ERR_ARG);
    EXPECT(tg_tridiag_work_bytes(4, 16, TG_POINT_FASTEST) == 0);
    EXPECT(tg_tridiag_work_bytes(4, 16, TG_BATCH_FASTEST) == 3 * 4 * 16 * sizeof(double));
    EXPECT(tg_tridiag_solve(nullptr, dummy, dummy, dummy, dummy, 1, 4, 0, nullptr, nullptr, 0, nullptr) == TG_ERR_ARG);
    EXPECT(tg_tridiag_solve(dummy, dummy, dummy, dummy, dummy, -1, 4, 0, nullptr, nullptr, 0, nullptr) == TG_ERR_ARG);
    EXPECT(tg_tridiag_solve(dummy, dummy, dummy, dummy, dummy, 1, 4, 7, nullptr, nullptr, 0, nullptr) == TG_ERR_ARG);
    EXPECT(tg_tridiag_solve(dummy, dummy, dummy, dummy, dummy, 1, 4, TG_BATCH_FASTEST, nullptr, nullptr, 0, nullptr) == TG_ERR_ARG);
    EXPECT(tg_tridiag_solve(dummy, dummy, dummy, dummy, dummy, 0, 4, 0, nullptr, nullptr, 0, nullptr) == TG_OK);
    tg_hcoeffs H;
    memset(&H, 0, sizeof(H));
    EXPECT(tg_h_to_matrix(1.0, nullptr, 0.1, dummy, dummy, &H, dummy, dummy, dummy, dummy, nullptr, 1, 4, nullptr) == TG_ERR_ARG);
    EXPECT(tg_h_to_matrix(1.0, nullptr, 0.1, dummy, dummy, nullptr, dummy, dummy, dummy, dummy, nullptr, 1, 4, nullptr) == TG_ERR_ARG);
    EXPECT(tg_residual(dummy, dummy, dummy, dummy, dummy, dummy, nullptr, nullptr, nullptr, 1, 4, nullptr) == TG_ERR_ARG);
    EXPECT(tg_ewma(nullptr, dummy, 0.3, 1, 4, nullptr) == TG_ERR_ARG);
    EXPECT(tg_ewma_profiles(dummy, dummy, 0.3, nullptr, 1, 4, nullptr) == TG_ERR_ARG);
    EXPECT(tg_flux_split(dummy, dummy, 0.1, 1e-5, 1e13, 10, nullptr, nullptr, nullptr, dummy, nullptr, nullptr, nullptr, 1, 4, nullptr) == TG_ERR_ARG);
    EXPECT(tg_ddx(dummy, 0.1, 3, dummy, 1, 4, nullptr) == TG_ERR_ARG);
    EXPECT(tg_ddx(dummy, 0.1, 2, dummy, 1, 2, nullptr) == TG_ERR_ARG);
    EXPECT(tg_moving_average(dummy, 4, dummy, 1, 8, nullptr) == TG_ERR_ARG);
    EXPECT(tg_zone_apply(dummy, 1, 8, 4, 5, dummy, nullptr, dummy, 8, 0, 2, 0, nullptr) == TG_ERR_ARG);
    EXPECT(tg_zone_apply(nullptr, 1, 8, 0, 4, dummy, nullptr, dummy, 8, 0, 2, 0, nullptr) == TG_ERR_ARG);
    EXPECT(tg_moving_average(dummy, 9, dummy, 1, 8, nullptr) == TG_ERR_ARG);
    EXPECT(tg_spline_work_bytes(0, 10) == 0 && tg_spline_work_bytes(2, 3) == 0 && tg_spline_work_bytes(2, 10) >= 5 * 2 * 8 * sizeof(double));
    EXPECT(tg_spline_resample(dummy, dummy, 1, 3, dummy, 4, 0, 4, 0, dummy, dummy, 1 << 20, nullptr) == TG_ERR_ARG);
    EXPECT(tg_spline_resample(dummy, dummy, 1, 8, dummy, 4, 3, 2, 0, dummy, dummy, 1 << 20, nullptr) == TG_ERR_ARG);
    EXPECT(tg_spline_resample(dummy, dummy, 1, 8, dummy, 4, 0, 4, 0, dummy, nullptr, 0, nullptr) == TG_ERR_ARG);
    EXPECT(tg_tukey_window(nullptr, 8, 0.5, nullptr) == TG_ERR_ARG);
    EXPECT(tg_ar1_noise(1, 0, 0.1, 1.0, 0, nullptr, dummy, 1, 8, nullptr) == TG_ERR_ARG);
    EXPECT(tg_ar1_noise(1, 0, 0.1, 3.0, 4, nullptr, dummy, 1, 8, nullptr) == TG_ERR_ARG);
    EXPECT(tg_shestakov_flux(dummy, 0.1, nullptr, 1, 4, nullptr) == TG_ERR_ARG);
    EXPECT(tg_pair_solve(dummy, dummy, dummy, dummy, dummy, dummy, dummy, dummy, dummy, dummy, dummy, nullptr, 1, 4, nullptr, nullptr) == TG_ERR_ARG);
    EXPECT(tg_pair_solve(dummy, dummy, dummy, dummy, dummy, dummy, dummy, dummy, dummy, dummy, dummy, dummy, 1, (int64_t)1 << 31, nullptr, nullptr) == TG_ERR_ARG);
    EXPECT(tg_block3_solve(dummy, dummy, dummy, dummy, nullptr, 1, 4, nullptr, nullptr) == TG_ERR_ARG);
    EXPECT(tg_status_reduce(nullptr, 1, nullptr, nullptr) == TG_ERR_ARG);
    EXPECT(tg_residual_reduce(istat, istat, dummy, 0, 1, 4, dummy, nullptr) == TG_ERR_ARG);
    EXPECT(tg_stream_probe(nullptr, nullptr, 4, nullptr) == TG_ERR_ARG);

    tg_state_1f S;
    tg_params_1f Q;
    memset(&S, 0, sizeof(S));
    memset(&Q, 0, sizeof(Q));
    EXPECT(tg_picard_1f_step(nullptr, &Q, nullptr) == TG_ERR_ARG);
    S.nb = 1; S.n = 8;
    EXPECT(tg_picard_1f_step(&S, &Q, nullptr) == TG_ERR_ARG);            // NULL state arrays
    S.u = S.uE = S.DE = S.cE = dummy; S.um1 = dummy; S.status = istat; S.iters = istat;
    EXPECT(tg_picard_1f_step(&S, &Q, nullptr) == TG_ERR_ARG);            // no UL
    Q.UL = dummy;
    EXPECT(tg_picard_1f_step(&S, &Q, nullptr) == TG_ERR_ARG);            // neither H7 nor a box source
    Q.H7 = dummy; Q.flux_model = 9;
    EXPECT(tg_picard_1f_step(&S, &Q, nullptr) == TG_ERR_ARG);
    Q.flux_model = TG_FLUX_EXTERNAL;
    EXPECT(tg_picard_1f_step(&S, &Q, nullptr) == TG_ERR_ARG);            // external flux without an array
    Q.flux = dummy; Q.iters_per_launch = 4;
    EXPECT(tg_picard_1f_step(&S, &Q, nullptr) == TG_ERR_ARG);
    Q.iters_per_launch = 1; Q.tread_lightly = 1;
    EXPECT(tg_picard_1f_step(&S, &Q, nullptr) == TG_ERR_ARG);            // tread_lightly without pseudo_time
    Q.tread_lightly = 0; Q.smooth_window = 4;
    EXPECT(tg_picard_1f_step(&S, &Q, nullptr) == TG_ERR_ARG);            // even window
    Q.smooth_window = 1; Q.flux_mult = dummy; Q.flux_mult_stride = 3;
    EXPECT(tg_picard_1f_step(&S, &Q, nullptr) == TG_ERR_ARG);
    Q.flux_mult = nullptr; Q.flux_model = TG_FLUX_POWER; Q.flux_power = -1.0;
    EXPECT(tg_picard_1f_step(&S, &Q, nullptr) == TG_ERR_ARG);
    Q.flux_model = TG_FLUX_SHESTAKOV; S.nb = 0;
    EXPECT(tg_picard_1f_step(&S, &Q, nullptr) == TG_OK);                 // an empty ensemble is a no-op
    EXPECT(tg_picard_1f_work_bytes(4, 512) == 0 && tg_picard_1f_work_bytes(4, 8192) > 0 && tg_picard_1f_work_bytes(0, 8192) == 0);
    S.nb = 2; S.n = 512;
    EXPECT(tg_picard_1f_step_static(&S, &Q, nullptr) == TG_ERR_UNSUPPORTED);      // needs no work space: not its business
    S.nb = 1024; S.n = 8192;
    EXPECT(tg_picard_1f_step_static(&S, &Q, nullptr) == TG_ERR_ARG);              // beyond the built-in work space
    EXPECT(strstr(tg_last_error(), "TG_STATIC_WORK_BYTES") != nullptr);

    tg_state_3f S3;
    tg_params_3f Q3;
    memset(&S3, 0, sizeof(S3));
    memset(&Q3, 0, sizeof(Q3));
    S3.nb = 1; S3.n = 8;
    EXPECT(tg_picard_3f_step(&S3, &Q3, nullptr) == TG_ERR_ARG);
    EXPECT(tg_picard_3f_step(nullptr, &Q3, nullptr) == TG_ERR_ARG);

    // calls that reach the HIP runtime: TG_OK with a GPU, TG_ERR_HIP without, never anything else
    int count = -1;
    int rc = tg_device_count(&count);
    EXPECT(rc == TG_OK || (rc == TG_ERR_HIP && count == 0));
    void *p = nullptr;
    rc = tg_malloc(&p, 64);
    EXPECT(rc == TG_OK || rc == TG_ERR_HIP);
    if (rc == TG_OK) EXPECT(tg_free(p) == TG_OK);
    if (failures == 0) printf("host_check ok (%d device(s))\n", count);
    return failures ? 1 : 0;
}
