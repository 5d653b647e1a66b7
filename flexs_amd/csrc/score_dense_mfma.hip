// K2 score_mlp / score_ge on f32 MFMA -- placeholder until the kernels land:
// returns FX_EUNSUPPORTED so fx_score falls back to the shape-agnostic kernels.
#include "fx_common.h"

int fx_launch_score_dense_mfma(fx_engine*, fx_model* const*, int, const uint8_t*, int64_t, float*, int, int) {
    return FX_EUNSUPPORTED;
}
