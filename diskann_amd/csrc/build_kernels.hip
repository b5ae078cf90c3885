// build_kernels.hip -- placeholder until the RobustPrune / multi_insert kernels land.
#include "dann_internal.h"
using namespace dann;
extern "C" {
int32_t dann_prune_batch(dann_index*, const dann_build_config*, const uint32_t*, uint32_t, const uint32_t*,
                         const float*, const uint64_t*, int32_t, uint32_t*) {
    set_error("dann_prune_batch: not built yet");
    return DANN_EUNSUPPORTED;
}
int32_t dann_insert_batch(dann_index*, const dann_build_config*, const uint32_t*, uint32_t) {
    set_error("dann_insert_batch: not built yet");
    return DANN_EUNSUPPORTED;
}
int32_t dann_build(dann_index*, const dann_build_config*, uint32_t, uint32_t, float, uint32_t) {
    set_error("dann_build: not built yet");
    return DANN_EUNSUPPORTED;
}
}
