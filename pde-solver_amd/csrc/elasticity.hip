// elasticity.hip — placeholder entry points (implemented after the heat path is parity-green).
#include "common.h"

extern "C" int pdegpu_elasticity_solve(pdegpu_ctx* ctx, const pdegpu_grid*, const pdegpu_elasticity*, double*, double*,
                                       pdegpu_stats*) {
    return pdegpu_set_err(ctx, PDEGPU_EUNSUPPORTED, "elasticity path not built yet");
}
extern "C" int pdegpu_elasticity_apply(pdegpu_ctx* ctx, const pdegpu_grid*, const pdegpu_elasticity*, const double*, double*, int) {
    return pdegpu_set_err(ctx, PDEGPU_EUNSUPPORTED, "elasticity path not built yet");
}
