/* Driver for the sanitizer build of the C restatement (test infrastructure): steps a small, ragged ensemble a few
 * iterations with several thread counts and checks that the results are finite. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef struct { double dt, dx, a_turb, a_prof, Dmin, Dmax, thr, tol; } orc_params;
int orc_picard_1f_step(int64_t nb, int n, double *u, const double *um1, double *uE, double *DE, double *cE, int32_t *status,
                       int32_t *iters, double *err, int64_t err_stride, const double *UL, const double *S0, const double *delta,
                       const orc_params *q, int fixed_count, int nthreads);

int main(void) {
    const int sizes[] = {3, 8, 37, 512};
    for (int s = 0; s < 4; ++s) {
        const int n = sizes[s];
        const int64_t nb = 5;
        const int64_t maxit = 6;
        double *u = malloc(sizeof(double) * nb * n), *um1 = malloc(sizeof(double) * nb * n);
        double *uE = calloc(nb * n, sizeof(double)), *DE = calloc(nb * n, sizeof(double)), *cE = calloc(nb * n, sizeof(double));
        int32_t *status = calloc(nb, sizeof(int32_t)), *iters = calloc(nb, sizeof(int32_t));
        double *err = calloc(nb * maxit, sizeof(double));
        double UL[5], S0[5], delta[5];
        const double dx = 1.0 / (n - 1);
        for (int64_t b = 0; b < nb; ++b) {
            UL[b] = 0.01 + 0.002 * b; S0[b] = 1.0 + 0.1 * b; delta[b] = 0.1 + 0.02 * b;
            for (int j = 0; j < n; ++j) u[b * n + j] = um1[b * n + j] = 1.0 - 0.5 * j * dx;
        }
        const orc_params q = {1e4, dx, 0.3, 0.3, 1e-5, 1e13, 10.0, 1e-11};
        for (int it = 0; it < maxit; ++it)
            orc_picard_1f_step(nb, n, u, um1, uE, DE, cE, status, iters, err, maxit, UL, S0, delta, &q, it % 2, 1 + it % 3);
        for (int64_t k = 0; k < nb * n; ++k)
            if (!isfinite(u[k]) || !isfinite(DE[k]) || !isfinite(cE[k])) { fprintf(stderr, "non-finite result at n=%d\n", n); return 1; }
        free(u); free(um1); free(uE); free(DE); free(cE); free(status); free(iters); free(err);
    }
    printf("oracle sanitizer check ok\n");
    return 0;
}
