// placeholder entry points; replaced by the fused kernels
#include "common.h"
extern "C" size_t prism_learner_workspace_bytes(const prism_model_dims *, int32_t) { return 0; }
extern "C" int prism_learner_supported(const prism_model_dims *, int32_t) { return PRISM_ERR_UNSUPPORTED; }
extern "C" int prism_learner_fwd_bwd(const prism_learner_desc *, prism_stream_t) {
    prism::set_error("not built yet");
    return PRISM_ERR_UNSUPPORTED;
}
extern "C" int prism_learner_clip_adam(const prism_learner_desc *, prism_stream_t) {
    prism::set_error("not built yet");
    return PRISM_ERR_UNSUPPORTED;
}
