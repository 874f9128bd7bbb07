// window.hip -- grid-cell windowed matching (a5, a6, a11).  PLACEHOLDER: replaced by the real kernels next.
#include "orbm_internal.h"
extern "C" int orbm_search_proj_points_batch(orbm_ctx *ctx, int, const int32_t *, const int32_t *, float, float,
                                             const uint8_t *, int32_t, int32_t *, int32_t *, int) {
    return orbm_fail(ctx, ORBM_EINVAL, "search_proj_points_batch: not built yet");
}
extern "C" int orbm_search_window_batch(orbm_ctx *ctx, int, int, const int32_t *, const int32_t *, int, int,
                                        const uint8_t *, int32_t, int32_t *, int32_t *, int) {
    return orbm_fail(ctx, ORBM_EINVAL, "search_window_batch: not built yet");
}
extern "C" int orbm_search_init(orbm_ctx *ctx, int32_t, int32_t, float *, int, float, int, int32_t *) {
    return orbm_fail(ctx, ORBM_EINVAL, "search_init: not built yet");
}
