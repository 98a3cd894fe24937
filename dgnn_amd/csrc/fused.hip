// Fused inference layer -- placeholder until the persistent kernel lands (returns UNSUPPORTED so
// callers use the aggregate + linear pair).
#include "common.h"

extern "C" int dgnn_sage_layer_fused_fwd(const int32_t*, const int32_t*, int64_t, const float*, int64_t, int, const float*,
                                         int64_t, int, const float*, const float*, const float*, const float*, const float*,
                                         const float*, const float*, int, int, float*, int64_t, void*) {
    dgnn_set_error("sage_layer_fused_fwd: not built in this version");
    return DGNN_E_UNSUPPORTED;
}
