// rrtmg_lw.hip -- longwave kernels and launch sequence (placeholder until the LW path lands)
#include "rrtmg_ctx.h"
namespace rrtmg {
void free_lw_desc(rrtmg_ctx *ctx) { (void)ctx; }
int lw_init_impl(rrtmg_ctx *ctx, double, const char *) { return ctx->fail(RRTMG_ERR_UNSUPPORTED, "longwave not built yet"); }
int lw_fluxes_impl(rrtmg_ctx *ctx, const rrtmg_lw_args *) { return ctx->fail(RRTMG_ERR_UNSUPPORTED, "longwave not built yet"); }
}  // namespace rrtmg
