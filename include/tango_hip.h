/* tango_hip.h -- C ABI of libtango_hip.so: batched fp64 HIP kernels (gfx950) for Tango's
 * implicit transport inner loop.
 *
 * The reference (sundials-codes/tango) is pure Python and has no FFI on this path: its boundary is
 * Python call signatures on 1-D float64 NumPy arrays.  Each entry point below names the reference
 * function(s) it replaces; the Python package tango_amd/ binds them with ctypes and keeps the
 * reference's signatures.
 *
 * Conventions
 *  - every array argument is a DEVICE pointer to float64 unless the name ends in _host;
 *  - nb = number of independent profiles, n = number of radial grid points;
 *  - layout TG_POINT_FASTEST stores profile b, point j at X[b*n + j] (a C-contiguous (nb, n) array);
 *    TG_BATCH_FASTEST stores it at X[j*nb + b] (profiles interleaved so that a lane-per-profile
 *    sweep is coalesced).  Entry points without a layout argument are point-fastest;
 *  - "shared or per-profile" coefficient arrays come with a stride in elements: 0 = one (n,) array
 *    used by every profile, n = an (nb, n) array;
 *  - nothing here allocates or frees device memory except tg_malloc/tg_free; work space is passed in;
 *  - every call is asynchronous on `stream` (a hipStream_t; NULL = the default stream);
 *  - return value: TG_OK or an error code; tg_last_error() gives the message of the calling thread.
 */
#ifndef TANGO_HIP_H
#define TANGO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *tg_stream;

enum { TG_OK = 0, TG_ERR_ARG = 1, TG_ERR_HIP = 2, TG_ERR_UNSUPPORTED = 3 };
enum { TG_POINT_FASTEST = 0, TG_BATCH_FASTEST = 1, TG_BATCH_FASTEST_TILED = 2 /* tg_tridiag_solve only: see there */ };

/* per-profile status bits written by the solvers */
enum {
    TG_ST_CONVERGED = 1, /* residual dropped below tol (tango/solver.py:472-475) */
    TG_ST_INVALID = 2,   /* NaN, Inf or negative value in a new profile (tango/solver.py:420-439) */
    TG_ST_SINGULAR = 4,  /* zero pivot met (scipy raises LinAlgError on it) */
    TG_ST_SEEDED = 8,    /* the EWMA memories hold a previous iterate (tango/lodestro_method.py:706-708) */
    TG_ST_TREAD_LIMIT = 16 /* pseudo-timestep control gave up halving after TG_TREAD_MAX_HALVINGS (see tg_params_1f) */
};

/* ---- library / device / memory plumbing (no reference counterpart) ------------------------- */
int tg_version(void);
const char *tg_last_error(void);
int tg_device_count(int *count);
int tg_set_device(int device);
int tg_malloc(void **ptr, size_t bytes);
int tg_free(void *ptr);
/* page-locked host memory: copies between it and the device are asynchronous and run at the full PCIe rate */
int tg_host_alloc(void **ptr, size_t bytes);
int tg_host_free(void *ptr);
int tg_memset(void *ptr, int value, size_t bytes, tg_stream stream);
int tg_copy_h2d(void *dst, const void *src_host, size_t bytes, tg_stream stream);
int tg_copy_d2h(void *dst_host, const void *src, size_t bytes, tg_stream stream);
int tg_copy_d2d(void *dst, const void *src, size_t bytes, tg_stream stream);
/* rows x width_bytes from one pitched device array into another (e.g. the written columns of a history, packed
 * before they cross PCIe) */
int tg_copy_d2d_2d(void *dst, size_t dst_pitch, const void *src, size_t src_pitch, size_t width_bytes, size_t rows,
                   tg_stream stream);
int tg_stream_create(tg_stream *stream);
int tg_stream_destroy(tg_stream stream);
int tg_stream_sync(tg_stream stream);

/* Streaming probe (no reference counterpart): out[k][i] = in[k][i] + in[k+1][i] over `count` doubles,
 * five input and four output streams as in one fused one-field Picard step.  Used by the benchmark to
 * measure the achievable HBM rate for that traffic shape on the device at hand. */
int tg_stream_probe(const double *const in[5], double *const out[4], int64_t count, tg_stream stream);

/* ---- tridiagonal solve: tango/tridiag.py:12-31 (solve), :88-97 (solve_with_scipy -> dgtsv) ---
 * C_j u_{j-1} + B_j u_j + A_j u_{j+1} = D_j, j = 0..n-1; A[n-1] and C[0] are ignored (LAPACK's band
 * storage never reads them).  u may alias D.
 *  - point-fastest: one wavefront (or one workgroup when n > 512) per profile, chunked elimination +
 *    parallel cyclic reduction across lanes; no work space needed (work may be NULL); n <= 16384; no pivoting
 *    (within rounding of dgtsv for the diagonally dominant matrices of the transport problem);
 *  - batch-fastest: one lane per profile performing LAPACK dgtsv's sweeps operation by operation, row
 *    interchanges included, without floating-point contraction: bit-identical to the reference's result.  Needs
 *    tg_tridiag_work_bytes() of work space (pivots, modified super-diagonal, second super-diagonal).  u may
 *    alias D but not A, B or C;
 *  - TG_BATCH_FASTEST_TILED: the same interleaved arrays X[j*nb + b], solved by the point-fastest algorithm (a
 *    workgroup moves 16 neighbouring profiles through LDS, a wavefront each): the results of TG_POINT_FASTEST
 *    on the transposed arrays (bit for bit for 256 < n <= 512, where both give a profile a whole wavefront), 40 bytes of memory traffic per point instead of 80; n <= 512; no work
 *    space; u may alias any input.
 * status: optional (nb,) int32, TG_ST_SINGULAR is OR-ed in for a profile that met a zero pivot. */
size_t tg_tridiag_work_bytes(int64_t nb, int64_t n, int layout);
int tg_tridiag_solve(const double *A, const double *B, const double *C, const double *D, double *u,
                     int64_t nb, int64_t n, int layout, int32_t *status, void *work,
                     size_t work_bytes, tg_stream stream);

/* ---- finite-difference assembly: tango/HToMatrixFD.py:53-117 (H_to_matrix), :268-299 (_fbeta);
 *      coupling column tango/JK_to_matrix_fd.py:119-130 (_H8_contribution) ------------------------
 * H1 is required; H2,H3,H4,H6,H7,H8 may be NULL (= absent).  hs[k] is the stride of Hk as above
 * (index 0..7 for H1..H8, slot of the non-existent H5 unused).  UL is (nb,); dt_b is an optional
 * (nb,) per-profile timestep overriding dt.  G (optional output) = -H8 with 0 on the last row. */
typedef struct {
    const double *H1, *H2, *H3, *H4, *H6, *H7, *H8;
    int64_t s1, s2, s3, s4, s6, s7, s8;
} tg_hcoeffs;
int tg_h_to_matrix(double dt, const double *dt_b, double dx, const double *UL,
                   const double *U_mminus1, const tg_hcoeffs *H, double *A, double *B, double *C,
                   double *D, double *G, int64_t nb, int64_t n, tg_stream stream);

/* ---- residual: tango/fieldgroups.py:122-144, :226-256; rms of tango/solver.py:464-470 ----------
 * r = (A u_{j+1} + B u_j + C u_{j-1} [+ G w_j] - f) / max_j|f_j|.  resid (nb,n) and sumsq (nb,) are
 * optional outputs; sumsq[b] = sum_j r_j^2 (the caller forms sqrt(sum over fields / total length)). */
int tg_residual(const double *A, const double *B, const double *C, const double *f, const double *u,
                const double *G, const double *w, double *resid, double *sumsq, int64_t nb,
                int64_t n, tg_stream stream);

/* ---- LoDestro pieces: tango/lodestro_method.py:682-728 (EWMA), :563-679 (FluxSplit),
 *      tango/derivatives.py:9-41, tango/smoother.py:13-48,
 *      tango/extras/shestakov_nonlinear_diffusion.py:51-61 ------------------------------------- */
int tg_ewma(const double *y, double *y_ewma_inout, double param, int seeded, int64_t count,
            tg_stream stream);
/* The EWMA of the profiles of an ensemble as the Picard loop advances it (tango/lodestro_method.py:345, :693-711 with
 * the loop control of tango/solver.py:150-154): for every profile that still iterates (status has neither
 * TG_ST_CONVERGED nor TG_ST_INVALID) y_ewma = a y + (1 - a) y_ewma if its memory is seeded (TG_ST_SEEDED), else
 * a y + (1 - a) y; profiles that have stopped keep their memory.  y, y_ewma: (nb, n); status: (nb,). */
int tg_ewma_profiles(const double *y, double *y_ewma_inout, double param, const int32_t *status, int64_t nb, int64_t n,
                     tg_stream stream);
/* gxx/gradx: optional (n,) shared arrays (NULL = 1).  DHat, cHat, theta: optional outputs. */
int tg_flux_split(const double *flux, const double *p, double dx, double Dmin, double Dmax,
                  double dpdx_thr, const double *gxx, const double *gradx, double *D, double *c,
                  double *DHat, double *cHat, double *theta, int64_t nb, int64_t n,
                  tg_stream stream);
/* edge_order 1: two-point one-sided ends; 2: three-point second-order ends */
int tg_ddx(const double *u, double dx, int edge_order, double *out, int64_t nb, int64_t n,
           tg_stream stream);
/* window must be odd (callers add one to even windows as the reference's Smoother does) */
int tg_moving_average(const double *x, int window, double *out, int64_t nb, int64_t n,
                      tg_stream stream);
/* ---- resampling between the transport grid and a turbulence grid ----------------------------------
 * tango/interfacegrids_gene.py:51-57 (map_profile_onto_turb_grid) and :446-484 (extend_with_zeros_both_sides) call
 * scipy.interpolate.InterpolatedUnivariateSpline: the cubic spline through all points with not-a-knot ends.  The
 * same curve is built here for every row of yin (nb, nin) over the increasing grid xin (nin,) and evaluated at
 * xout[out_lo:out_hi]; the other entries of yout (nb, nout) are set to zero (a coefficient extended with zeros onto
 * a larger grid); nonneg != 0 replaces negative values by zero (enforcePositive).  Points of xout outside xin
 * continue the end cubics, as the reference's spline does. */
/* out[b, out_lo + r] = bias[r] + sum_{z < zone} M[r * zone + z] * in[b, in_lo + z] for r < rows (bias may be NULL; clipped at
 * zero from below when nonneg), the rest of out untouched: the polynomial extrapolation of the transport coefficients beyond
 * the trusted part of the turbulence grid (np.polyfit over the points of a zone + np.polyval beyond it,
 * tango/interfacegrids_gene.py:288-292, :349-360) as the fixed matrix it is, M = polyval(polyfit(x_zone, I)) built once on the
 * host from the two grids; likewise the straight-line continuations of a profile beyond the transport grid (:545-576 fitted
 * slope: a matrix over the last ten points; :636-662 imposed slope: the end point plus bias = slope * (x - x_end)). */
int tg_zone_apply(const double *in, int64_t nb, int64_t nin, int64_t in_lo, int64_t zone, const double *M, const double *bias, double *out,
                  int64_t nout, int64_t out_lo, int64_t rows, int nonneg, tg_stream stream);

size_t tg_spline_work_bytes(int64_t nb, int64_t nin);
int tg_spline_resample(const double *xin, const double *yin, int64_t nb, int64_t nin, const double *xout, int64_t nout,
                       int64_t out_lo, int64_t out_hi, int nonneg, double *yout, void *work, size_t work_bytes,
                       tg_stream stream);
/* Symmetric Tukey (tapered cosine) window of n points with cosine fraction alpha, the taper scipy.signal.tukey gives
 * tango/extras/bufferzone.py:51: w (n,) device array. */
int tg_tukey_window(double *w, int64_t n, double alpha, tg_stream stream);

/* Multiplicative flux noise of tango/extras/noisyflux.py:43-62,106-162 generated on the device:
 * out[b][j] = (1 + d_j amplitude y_bj) * (taper ? taper[j] : 1), y_b a unit-variance AR(1) series along the grid with
 * lag-one correlation (tac-1)/(tac+1) (tac = correlation length in grid points, > 1), d the linear ramp of
 * dampen_sides (0 at both ends, 1 after damp_points points).  The white noise comes from a counter-based generator
 * (Philox 4x32-10, Box-Muller) keyed by (seed, sequence, profile, point): the same arguments give the same numbers on
 * any launch shape; use a new sequence number for every Picard iteration.  The series starts in its stationary
 * distribution (the reference reaches it by discarding a burn-in).  The reference draws from NumPy's global Mersenne
 * Twister, which no device generator reproduces: statistically equivalent noise, not the same numbers. */
int tg_ar1_noise(uint64_t seed, uint64_t sequence, double amplitude, double tac, int32_t damp_points, const double *taper,
                 double *out, int64_t nb, int64_t n, tg_stream stream);

/* Gamma = -(dn/dx)^3 / n^2 */
int tg_shestakov_flux(const double *nprof, double dx, double *flux, int64_t nb, int64_t n,
                      tg_stream stream);

/* ---- pair-coupled solve: tango/JK_to_matrix_fd.py:132-166 (interleaved 5-band dgbsv) -----------
 * C^U U_{j-1} + B^U U_j + A^U U_{j+1} + G^W W_j = D^U and the same with U<->W; solved as a
 * block-tridiagonal system with 2x2 blocks [[BU,GW],[GU,BW]] and diagonal off-diagonal blocks. */
int tg_pair_solve(const double *AU, const double *BU, const double *CU, const double *DU,
                  const double *GW, const double *AW, const double *BW, const double *CW,
                  const double *DW, const double *GU, double *U, double *W, int64_t nb, int64_t n,
                  int32_t *status, tg_stream stream);

/* ---- 3x3 block-tridiagonal solve (superset of the reference's n + (pi,pe) structure) -----------
 * Row j: Cb_j x_{j-1} + Bb_j x_j + Ab_j x_{j+1} = d_j with 3x3 blocks.  Blocks are stored
 * component-major: Bb[(r*3+c)*nb*n + b*n + j]; d and x: [r*nb*n + b*n + j]. */
int tg_block3_solve(const double *Cb, const double *Bb, const double *Ab, const double *d, double *x,
                    int64_t nb, int64_t n, int32_t *status, tg_stream stream);

/* ---- fused Picard iteration, one field --------------------------------------------------------
 * One call = one pass (or, with iters_per_launch = k, up to k passes) of tango/solver.py:188-272 (compute_next_iteration) for every active profile of
 * the ensemble with the analytic flux built in: EWMA(profile) -> flux -> [smoothing] -> FluxSplit ->
 * EWMA(D), EWMA(c) -> H2,H3 -> H_to_matrix -> residual of the current profile -> tridiagonal solve ->
 * validity check.  State arrays are (nb, n) point-fastest and stay resident on the device.
 * The state and parameter blocks are plain structs owned by the caller (there is no create/destroy/get/set: the
 * caller allocates the arrays, fills in the pointers and may read or replace any array between calls). */
typedef struct {
    int64_t nb, n;
    double *u;         /* current profile, overwritten with the new iterate */
    const double *um1; /* profile of the previous timestep (tango/multifield.py:18) */
    double *uE;        /* EWMA memory of the profile */
    double *DE, *cE;   /* EWMA memories of D and c */
    int32_t *status;   /* (nb,) TG_ST_* bits */
    int32_t *iters;    /* (nb,) iterations performed so far (column of err to write next) */
    double *err;       /* (nb, err_stride) rms residual history, or NULL */
    int64_t err_stride;
} tg_state_1f;

/* TG_FLUX_SHESTAKOV: Gamma = -(dp/dx)^3 / p^2 (tango/extras/shestakov_nonlinear_diffusion.py:51-61).
 * TG_FLUX_EXTERNAL: the flux array of tg_params_1f.flux.  TG_FLUX_POWER: Gamma = -|(dp/dx)/p|^q dp/dx with q =
 * tg_params_1f.flux_power (the FluxModel of examples/stripped_down_tango_example.py:55-65; q = 2 is the Shestakov model). */
enum { TG_FLUX_SHESTAKOV = 0, TG_FLUX_EXTERNAL = 1, TG_FLUX_POWER = 2 };

typedef struct {
    double dt, dx;
    double a_turb, a_prof;       /* EWMA parameters (tango/lodestro_method.py:417) */
    double Dmin, Dmax, dpdx_thr; /* thetaParams (tango/lodestro_method.py:664-666) */
    double tol;
    const double *UL;    /* (nb,) right boundary values */
    const double *dt_b;  /* optional (nb,) per-profile pseudo-timestep */
    const double *H1;    /* optional (n,) shared; NULL = 1 */
    const double *H7;    /* optional source array, stride h7_stride; NULL = box source below
                          * (a call that gives H1 or H7 as arrays takes the general form of the kernel, like one with
                          * geometry arrays or an external flux; with both NULL and the built-in Cartesian flux it takes
                          * the leaner built-in form) */
    int64_t h7_stride;
    const double *S0;        /* (nb,) box source used when H7 == NULL: H7_j = S0[b] for j < src_cut[b], else 0 */
    const int32_t *src_cut;  /* (nb,) number of grid points with x < delta (the grid is increasing) */
    const double *Vprime, *gxx, *gradx;    /* optional (n,) geometry on the transport grid */
    const double *gxx_turb, *gradx_turb;   /* optional (n,) geometry used by the flux split */
    const double *flux;  /* (nb,n) flux when flux_model == TG_FLUX_EXTERNAL (profile EWMA already advanced) */
    int32_t flux_model;
    int32_t smooth_window; /* centred moving average of the flux before the split (tango/smoother.py:13-48 at
                              tango/lodestro_method.py:359-362): odd window in points, <= 1 = no smoothing; n <= 4096 */
    int32_t fixed_count;   /* nonzero: step every profile regardless of its status, never mark converged */
    int32_t frozen_turb;   /* nonzero: reuse DE,cE as they are (inner iteration, tango/solver.py:274-309) */
    int32_t iters_per_launch; /* <= 1: one iteration per call; k > 1: up to k iterations per call with the profile
                                 held in registers in between (each profile still stops on its own iteration).  In the
                                 segmented form (see work) the k iterations are a chain of kernels inside the call and
                                 the results are those of k calls, with one exception: a profile that turns invalid
                                 (TG_ST_INVALID) at an iteration before the last has its EWMA memories advanced once
                                 more in the segments that did not see the invalid values */
    /* Pseudo-timestep control (tango/solver.py:198-216,233-235,311-372).  pseudo_time: the previous iterate takes
     * the u_{m-1} slot every iteration (useTreadLightly or useTreadInitially).  tread_initial_steps > 0: the first
     * that many iterations of a profile use tread_initial_dtau instead of dt.  tread_lightly: a new iterate whose
     * first tread_inner_count = int(innerFraction*n) points changed by a fraction >= tread_max_change is redone from
     * the previous iterate with the same D, c and dtau = dt/2, dt/4, ... until acceptable (the reference loops
     * without bound; here a profile stops after TG_TREAD_MAX_HALVINGS and gets TG_ST_TREAD_LIMIT). */
    int32_t pseudo_time;
    int32_t tread_lightly;
    int32_t tread_inner_count;
    int32_t tread_initial_steps;
    int32_t reserved_;     /* set to 0; used by the library */
    double tread_max_change;
    double tread_initial_dtau;
    /* Optional work space of tg_picard_1f_work_bytes(nb, n) bytes.  With it, profiles of n = 4096, 8192, 16384 or 32768
     * points (built-in flux, Cartesian) are stepped in segments of 512 points, one
     * wavefront each, in three kernels: turbulence update + assembly + elimination per segment; the reduced system
     * of the segment-lane unknowns per profile; back substitution per segment (with iters_per_launch > 1 the back
     * substitution of one iteration and the first kernel of the next are one kernel; n = 4096 with iters_per_launch > 1
     * takes the workgroup kernel, which then keeps the profile in registers and is faster).  Without it (NULL) such
     * profiles take the one-workgroup-per-profile kernel (n <= 16384), which is 2-4 times slower at those sizes. */
    void *work;
    size_t work_bytes;
    /* Optional history of the iterates (profilesAllIterations of tango/solver.py:225-227): the profile that
     * iteration number i of profile b produces (i = the profile's iteration count before the step) is also written to
     * iterates[(i * nb + b) * n ..], for i < iterates_rows.  Works with any iters_per_launch; the segmented form is
     * not used while it is set. */
    double *iterates;
    int64_t iterates_rows;
    /* Optional multiplier of the flux, applied before smoothing and the split: flux_j *= flux_mult[b*flux_mult_stride + j]
     * (stride 0: one (n,) array for all profiles, n: an (nb, n) array).  This is where the flux-model decorators of
     * tango/extras/bufferzone.py:39-54 (a Tukey taper, tg_tukey_window) and tango/extras/noisyflux.py:43-62 (1 + noise,
     * tg_ar1_noise) act when the flux model is built in; n <= 4096. */
    const double *flux_mult;
    int64_t flux_mult_stride;
    double flux_power;     /* exponent q of TG_FLUX_POWER */
} tg_params_1f;

enum { TG_TREAD_MAX_HALVINGS = 64 };

/* bytes of tg_params_1f.work that make the segmented form available for (nb, n); 0 where it does not apply
 * (tango/solver.py:188-272 has no counterpart of a work space: NumPy allocates its temporaries itself) */
size_t tg_picard_1f_work_bytes(int64_t nb, int64_t n);
/* the same for steps under pseudo-timestep control (pseudo_time set): room for the previous iterate on top.  With a work
 * space of this size such profiles keep the segmented form: one iteration at a time (iters_per_launch > 1: that many in a
 * row), tread_initial_* by a per-profile pseudo-timestep, tread_lightly by further attempts of the marked profiles with
 * dtau/2, dtau/4, ... and the same D and c (tango/solver.py:361-371); the host reads one counter per attempt, so a call
 * with tread_lightly returns with the stream drained.  With a smaller work space they take the workgroup kernel. */
size_t tg_picard_1f_work_bytes_tread(int64_t nb, int64_t n);
int tg_picard_1f_step(const tg_state_1f *state, const tg_params_1f *params, tg_stream stream);

/* The same step for callers that cannot allocate device memory (code under HIP graph capture, serving loops,
 * harnesses that forbid hipMalloc).  No counterpart in the reference.  It is tg_picard_1f_step for the profile
 * lengths whose step can use a work space (n = 4096, 8192, 16384, 32768; built-in flux, Cartesian),
 * with params->work ignored: the work space is a TG_STATIC_WORK_BYTES array inside the library, one per GPU, held
 * until the process ends.  tg_picard_1f_work_bytes(nb, n) (with pseudo_time: tg_picard_1f_work_bytes_tread) must not exceed
 * TG_STATIC_WORK_BYTES (about 4.7 million
 * grid points in all, e.g. 512 x 8192); larger ensembles get TG_ERR_ARG, other profile lengths TG_ERR_UNSUPPORTED
 * (they need no work space: tg_picard_1f_step allocates nothing for them either).  A call with n = 4096 and
 * iters_per_launch > 1 goes where tg_picard_1f_step sends it, to the workgroup kernel that keeps the profile in registers
 * and uses no work space.  The stream must belong to the device that is current (TG_ERR_ARG otherwise: every GPU has its
 * own copy of the array).  The call allocates and frees
 * nothing; the first call on a device looks up the array's address (make it outside graph capture).  All calls
 * on one GPU share that array: calls that overlap in time on one GPU (different streams or threads) race; calls
 * ordered on one stream are safe, every call clears what it needs cleared.  With tread_lightly the call waits for the
 * stream after every attempt (see tg_picard_1f_work_bytes_tread), which graph capture does not allow. */
enum { TG_STATIC_WORK_BYTES = 24 * 1024 * 1024 };
int tg_picard_1f_step_static(const tg_state_1f *state, const tg_params_1f *params, tg_stream stream);

/* counts[0..3] (device int64, zeroed by the call): profiles still active, converged, invalid, singular
 * (tango/solver.py:404-418: the `ok` property of a whole ensemble). */
int tg_status_reduce(const int32_t *status, int64_t nb, int64_t *counts, tg_stream stream);

/* status[b] |= bits for every profile (e.g. TG_ST_SEEDED after the first pass of a pipeline that keeps the EWMA memories
 * outside the fused step: tango/lodestro_method.py:706-708). */
int tg_status_or(int32_t *status, int64_t nb, int32_t bits, tg_stream stream);

/* Residual summary of an ensemble, the quantity tango/solver.py:464-475 (check_convergence) forms for one run, taken
 * over every profile that still iterates: out[0..4] (device doubles, zeroed by the call) = sum of squared normalised
 * residuals (n x rms^2 of each such profile's latest history entry), number of residual entries behind that sum,
 * number of profiles still iterating, largest latest rms of any profile, 1 if any profile is invalid.  Entries 0-2
 * add up over the shards of a distributed ensemble, entries 3-4 take the maximum; sqrt(out[0]/out[1]) is the rms
 * residual of everything that is still iterating.  err / err_stride / iters are those of tg_state_1f or tg_state_3f. */
int tg_residual_reduce(const int32_t *status, const int32_t *iters, const double *err, int64_t err_stride, int64_t nb,
                       int64_t n, double *out, tg_stream stream);

/* ---- fused Picard iteration, three fields n, pi<->pe -------------------------------------------
 * examples/shestakov_threefield_nonlinearnu.py:52-149 on the same call stack: shared D = (dpi/dx)^2/pi^2
 * for the three fluxes, sources S_n = 8[x<.3], S_i = [x<.1], S_e = 3x[x<.4], collisional coupling
 * nu = nu0/((pi+pe)/n)^{3/2} as H6 = -nu, H8 = +nu; n solved as a scalar tridiagonal system and
 * (pi,pe) as a 2x2 block-tridiagonal one (tango/fieldgroups.py:275-331). */
typedef struct {
    int64_t nb, n;
    double *u[3];          /* n, pi, pe */
    const double *um1[3];
    double *uE[3], *DE[3], *cE[3];
    int32_t *status, *iters;
    double *err;
    int64_t err_stride;
} tg_state_3f;

typedef struct {
    double dt, dx, a_turb, a_prof, Dmin, Dmax, dpdx_thr, tol;
    const double *UL[3]; /* (nb,) right boundary values of n, pi, pe */
    const double *H7[3]; /* (n,) source of each field, shared by all profiles */
    const double *nu0;   /* (nb,) */
    int32_t fixed_count; /* nonzero: step every profile regardless of its status, never mark converged */
    int32_t frozen_turb; /* nonzero: one pass of the inner iteration (tango/solver.py:274-309): D, c frozen, no EWMA
                            advance, no residual history entry, no convergence decision */
    int32_t outer_count; /* frozen passes only: process the profiles whose iteration count equals this value (those
                            that took part in the outer iteration just made, converged by it or not) */
    int32_t reserved_;
    /* Pseudo-timestep control, as in tg_params_1f (tango/solver.py:198-216, :233-235, :311-372); a step is redone with a
     * halved pseudo-timestep when ANY of the three profiles changed too much.  Not together with frozen_turb. */
    int32_t pseudo_time;
    int32_t tread_lightly;
    int32_t tread_inner_count;
    int32_t tread_initial_steps;
    double tread_max_change;
    double tread_initial_dtau;
    /* Inputs from outside the kernel: flux_model == TG_FLUX_EXTERNAL replaces the built-in closure by what the reference's
     * Python objects supply to the same stack.  flux[f]: (nb, n), field f's turbulent flux as fluxModel.get_flux returned
     * it for the EWMA of the profiles (tango/lodestro_method.py:342-350; the caller advances uE[f] with tg_ewma_profiles
     * before it evaluates the model, the kernel reads uE and does not write it), split per field as
     * tango/lodestro_method.py:365-369 does.  H6[f], H7[f], H8[f]: what the fields' compute_all_H callbacks returned
     * (tango/multifield.py:202, e.g. H6 = -nu, H8 = +nu of examples/shestakov_threefield_nonlinearnu.py:108-149), each
     * NULL (absent: zero) or an array with stride 0 ((n,), one for all profiles) or n ((nb, n)); H8[0] must be NULL
     * (field 0 is solved on its own, fields 1 and 2 as a pair: tango/fieldgroups.py:275-331, with H8[1] the coefficient
     * of field 2's profile in field 1's equation and the other way round, tango/JK_to_matrix_fd.py:97-112).  H1 is 1 and
     * nu0 is not used.  Not together with pseudo_time.  With frozen_turb (an inner iteration) the fluxes are not read. */
    int32_t flux_model;
    int32_t reserved2_;
    const double *flux[3];
    const double *H6[3];
    const double *H8[3];
    int64_t h6_stride[3], h7_stride[3], h8_stride[3];
} tg_params_3f;

int tg_picard_3f_step(const tg_state_3f *state, const tg_params_3f *params, tg_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* TANGO_HIP_H */
