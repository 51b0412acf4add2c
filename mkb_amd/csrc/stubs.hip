// Entry points declared in include/mkb_hip.h that are not implemented yet in this round.
#include "common.h"

extern "C" int mkb_rank(const mkb_tables_t *, const int64_t *, int64_t, int, const int64_t *, int64_t, int64_t *, void *,
                        int64_t, void *) {
    return mkb::set_error(MKB_ERR_UNSUPPORTED, "mkb_rank is not implemented yet");
}
