// Instantiates the pooled tile kernels (score_pool_kernels.h) for one model; one translation unit per model so the
// five families compile in parallel.
#include "score_pool_kernels.h"
#include "score_pool_tile.h"

namespace mkb {
MKB_DEFINE_POOL_LAUNCH(pool_launch_protate, MKB_PROTATE)
}  // namespace mkb
