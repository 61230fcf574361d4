// Backward (input-gradient) chain -- placeholder until the dgrad kernels land.
#include "common.h"

extern "C" int grip_vit_backward_prefix(grip_tower*, const float*, const float*, float*, void*, size_t, void*) {
    grip_set_error("vit_backward_prefix: not implemented yet");
    return GRIP_ERR_STATE;
}
extern "C" int grip_text_backward_prefix(grip_tower*, const float*, float*, void*, size_t, void*) {
    grip_set_error("text_backward_prefix: not implemented yet");
    return GRIP_ERR_STATE;
}
extern "C" int grip_cosine_head_backward(const float*, const float*, float, int, int, int, const float*, float*, float*, void*) {
    grip_set_error("cosine_head_backward: not implemented yet");
    return GRIP_ERR_STATE;
}
extern "C" int grip_weighted_ce(const float*, const int32_t*, const float*, int, int, float*, float*, void*) {
    grip_set_error("weighted_ce: not implemented yet");
    return GRIP_ERR_STATE;
}
