// placeholder: CRF pipeline (filled in next)
#include "rvseg_pipeline.h"
namespace rvseg {
void crf_state_free(Pipeline*) {}
rvseg_status crf_frames(rvseg_ctx* ctx, Pipeline*, int, const uint8_t*, const float*, float*, int8_t*, hipStream_t) {
    ctx->err = "dense CRF not implemented yet";
    return RVSEG_ERR_INVALID_ARG;
}
}  // namespace rvseg
extern "C" {
#define NOT_YET(ctx) do { if (ctx) (ctx)->err = "not implemented yet"; return RVSEG_ERR_INVALID_ARG; } while (0)
rvseg_status rvseg_crf_infer(rvseg_ctx* ctx, int32_t, int32_t, int32_t, const float*, const float*, float, int32_t, float*, int8_t*, int32_t, int32_t) { NOT_YET(ctx); }
rvseg_status rvseg_crf_infer_multi(rvseg_ctx* ctx, int32_t, int32_t, int32_t, const int32_t*, const float* const*, const float*, const float*, int32_t, float*, int8_t*, int32_t, int32_t) { NOT_YET(ctx); }
rvseg_status rvseg_lattice_build(rvseg_ctx* ctx, const float*, int32_t, int32_t, int32_t*, float*, int16_t*, int32_t, int32_t*) { NOT_YET(ctx); }
rvseg_status rvseg_lattice_filter(rvseg_ctx* ctx, const float*, int32_t, float*) { NOT_YET(ctx); }
}
