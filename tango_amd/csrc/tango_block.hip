// Coupled-field entry points (placeholder bodies until the block solvers land).
#include "../../include/tango_hip.h"
extern "C" {
int tg_pair_solve(const double *, const double *, const double *, const double *, const double *, const double *,
                  const double *, const double *, const double *, const double *, double *, double *, int64_t, int64_t,
                  int32_t *, tg_stream) { return TG_ERR_UNSUPPORTED; }
int tg_block3_solve(const double *, const double *, const double *, const double *, double *, int64_t, int64_t, int32_t *,
                    tg_stream) { return TG_ERR_UNSUPPORTED; }
int tg_picard_3f_step(const tg_state_3f *, const tg_params_3f *, tg_stream) { return TG_ERR_UNSUPPORTED; }
}
