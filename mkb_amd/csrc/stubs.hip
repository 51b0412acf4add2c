// Entry points declared in include/mkb_hip.h that are not implemented yet in this round.
#include "common.h"

extern "C" int mkb_rank(const mkb_tables_t *, const int64_t *, int64_t, int, const int64_t *, int64_t, int64_t *, void *,
                        int64_t, void *) {
    return mkb::set_error(MKB_ERR_UNSUPPORTED, "mkb_rank is not implemented yet");
}
extern "C" int64_t mkb_pool_step_workspace_bytes(const mkb_tables_t *, int64_t, int64_t) { return 0; }
extern "C" int mkb_pool_step(const mkb_tables_t *, const mkb_grads_t *, const int64_t *, const float *, const int64_t *,
                             const uint16_t *, int64_t, int64_t, int, float, float *, float *, float *, void *, void *) {
    return mkb::set_error(MKB_ERR_UNSUPPORTED, "mkb_pool_step is not implemented yet");
}
extern "C" int mkb_pool_score_fwd(const mkb_tables_t *, const int64_t *, const int64_t *, const uint16_t *, int64_t, int64_t,
                                  int, float *, void *, void *) {
    return mkb::set_error(MKB_ERR_UNSUPPORTED, "mkb_pool_score_fwd is not implemented yet");
}
