/* C restatement of one Picard iteration of Tango's one-field transport solve (TEST INFRASTRUCTURE ONLY).
 *
 * Used (a) as a second, independent checker of the HIP kernels and (b) as the CPU baseline that bench.py
 * times beside the GPU on the same host ("port" of the reference path; OpenMP over profiles).  It is never
 * linked into or called by the product library.  Arithmetic follows the reference's expressions in order:
 *   tango/lodestro_method.py:342-375 (EWMA, flux split), tango/extras/shestakov_nonlinear_diffusion.py:51-61
 *   (flux), tango/derivatives.py:9-24, tango/HToMatrixFD.py:87-117,268-299 (assembly, upwinding),
 *   tango/fieldgroups.py:139-140 + tango/solver.py:464-470 (residual), tango/tridiag.py:33-63 (Thomas sweep;
 *   the reference's default LAPACK dgtsv performs the same elimination when no row exchange occurs, which is
 *   the case for these diagonally dominant matrices).
 * Pinned against the golden vectors in tests/test_oracle_c.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
    double dt, dx, a_turb, a_prof, Dmin, Dmax, thr, tol;
} orc_params;

static void ddx_edge1(const double *u, int n, double dx, double *out) {
    out[0] = (u[1] - u[0]) / dx;
    for (int j = 1; j < n - 1; ++j) out[j] = (u[j + 1] - u[j - 1]) / (2 * dx);
    out[n - 1] = (u[n - 1] - u[n - 2]) / dx;
}

static double fbeta(double H2h, double H3h, double dx) {
    const double gamma = 0.9;
    double c = -H3h, beta = 0.0;
    if (c > 0) {
        double star = 1 - H2h / (c * dx);
        if (star <= 0.5) beta = 0.5;
        else if (star > 0.5) beta = gamma * star + (1 - gamma);
    } else if (c < 0) {
        double star = -H2h / (c * dx);
        if (star >= 0.5) beta = 0.5;
        else if (star < 0.5) beta = gamma * star;
    }
    return beta;
}

/* One iteration of one profile.  w: work space of 12*n doubles.  Returns the rms residual evaluated before
 * the solve; *bad is set when the new profile has a non-finite or negative entry. */
static double step_profile(int n, double *u, const double *um1, double *uE, double *DE, double *cE, int seeded,
                           double UL, double S0, double delta, const orc_params *q, double *w, int *bad) {
    double *dp = w, *A = w + n, *B = w + 2 * n, *C = w + 3 * n, *f = w + 4 * n, *H2 = w + 5 * n, *H3 = w + 6 * n;
    double *h2h = w + 7 * n, *h3h = w + 8 * n, *bet = w + 9 * n, *cp = w + 10 * n, *dd = w + 11 * n;
    const double dx = q->dx, dt = q->dt;
    for (int j = 0; j < n; ++j) uE[j] = q->a_prof * u[j] + (1 - q->a_prof) * (seeded ? uE[j] : u[j]);
    ddx_edge1(uE, n, dx, dp);
    for (int j = 0; j < n; ++j) {
        double flux = -pow(dp[j], 3) / (uE[j] * uE[j]);
        double DHat = -flux / (1 * dp[j]);
        if (dp[j] == 0) DHat = 0;
        double cHat = flux / (1 * uE[j]);
        double theta = 1;
        if (DHat < q->Dmin) theta = 0;
        if (fabs(dp[j]) < q->thr && DHat >= q->Dmin && DHat <= q->Dmax) theta = (1 + (q->Dmax - DHat) / (q->Dmax - q->Dmin)) / 2;
        if (fabs(dp[j]) < q->thr && DHat > q->Dmax) theta = 0.5;
        double D = theta * DHat, c = (1 - theta) * cHat;
        DE[j] = q->a_turb * D + (1 - q->a_turb) * (seeded ? DE[j] : D);
        cE[j] = q->a_turb * c + (1 - q->a_turb) * (seeded ? cE[j] : c);
        H2[j] = DE[j];
        H3[j] = -cE[j];
    }
    for (int j = 0; j < n - 1; ++j) {
        h2h[j] = (H2[j] + H2[j + 1]) / 2;
        h3h[j] = (H3[j] + H3[j + 1]) / 2;
        bet[j] = fbeta(h2h[j], h3h[j], dx);
    }
    const double dx2 = dx * dx;
    double mx = 0;
    for (int j = 0; j < n; ++j) {
        double a = 0, b = 0, c = 0, g = 0;
        if (j < n - 1) {
            double x = (double)j * dx;
            double h7 = (x < delta) ? S0 : 0.0;
            b += 1.0 / dt;
            g += -1.0 * um1[j] / dt;
            if (j == 0) {
                a += -h2h[0] / dx2;
                b += h2h[0] / dx2;
                a += -h3h[0] * (1 - bet[0]) / dx;
                b += -h3h[0] * bet[0] / dx;
            } else {
                a += -h2h[j] / dx2;
                b += (h2h[j - 1] + h2h[j]) / dx2;
                c += -h2h[j - 1] / dx2;
                a += -h3h[j] * (1 - bet[j]) / dx;
                b += (-h3h[j] * bet[j] + h3h[j - 1] * (1 - bet[j - 1])) / dx;
                c += h3h[j - 1] * bet[j - 1] / dx;
            }
            g += -h7;
        } else {
            b = 1;
            g = -UL;
        }
        A[j] = a; B[j] = b; C[j] = c; f[j] = -g;
        if (fabs(f[j]) > mx) mx = fabs(f[j]);
    }
    double ss = 0;
    for (int j = 0; j < n; ++j) {
        double up = j < n - 1 ? u[j + 1] : 0.0, um = j > 0 ? u[j - 1] : 0.0;
        double r = A[j] * up + B[j] * u[j] + C[j] * um - f[j];
        r = r / mx;
        ss += r * r;
    }
    /* Thomas sweep */
    cp[0] = A[0] / B[0];
    dd[0] = f[0] / B[0];
    for (int j = 1; j < n; ++j) {
        double den = B[j] - C[j] * cp[j - 1];
        cp[j] = A[j] / den;
        dd[j] = (f[j] - C[j] * dd[j - 1]) / den;
    }
    u[n - 1] = dd[n - 1];
    for (int j = n - 2; j >= 0; --j) u[j] = dd[j] - cp[j] * u[j + 1];
    int b_ = 0;
    for (int j = 0; j < n; ++j)
        if (!(isfinite(u[j]) && u[j] >= 0)) b_ = 1;
    *bad = b_;
    return sqrt(1.0 / n * ss);
}

/* One iteration of every active profile of an ensemble ((nb, n) row-major arrays).  status bits as in the
 * HIP library: 1 converged, 2 invalid, 8 seeded.  Returns the number of threads used. */
int orc_picard_1f_step(int64_t nb, int n, double *u, const double *um1, double *uE, double *DE, double *cE,
                       int32_t *status, int32_t *iters, double *err, int64_t err_stride, const double *UL,
                       const double *S0, const double *delta, const orc_params *q, int fixed_count, int nthreads) {
    int used = 1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    used = omp_get_max_threads();
#endif
#pragma omp parallel
    {
        double *w = (double *)malloc(sizeof(double) * 12 * (size_t)n);
#pragma omp for schedule(static)
        for (int64_t b = 0; b < nb; ++b) {
            int st = status[b];
            if (!fixed_count && (st & 3)) continue;
            int bad = 0;
            double rms = step_profile(n, u + b * n, um1 + b * n, uE + b * n, DE + b * n, cE + b * n, (st & 8) != 0, UL[b],
                                      S0[b], delta[b], q, w, &bad);
            int it = iters[b];
            if (err && it < err_stride) err[b * err_stride + it] = rms;
            iters[b] = it + 1;
            st |= 8;
            if (!fixed_count && rms < q->tol) st |= 1;
            if (bad) st |= 2;
            status[b] = st;
        }
        free(w);
    }
    return used;
}
