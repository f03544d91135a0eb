// avs_dist.hip -- multi-GPU layer: RCCL halo exchange + all-reduce.  Placeholder
// until the partitioned solver lands; every entry point reports AVS_ESTATE.
#include "avs_internal.hpp"

namespace avs {
struct PcgDist {};
avs_status dist_halo_exchange(PcgDist *, double *, hipStream_t) { set_error("multi-GPU layer not initialised"); return AVS_ESTATE; }
avs_status dist_allreduce(PcgDist *, double *, int, hipStream_t) { set_error("multi-GPU layer not initialised"); return AVS_ESTATE; }
void dist_release(avs_ctx *) {}
} // namespace avs

extern "C" {
avs_status avs_dist_get_unique_id(uint8_t *) { avs::set_error("multi-GPU layer not built yet"); return AVS_ESTATE; }
avs_status avs_dist_init(avs_ctx *, const uint8_t *, int32_t, int32_t) { avs::set_error("multi-GPU layer not built yet"); return AVS_ESTATE; }
avs_status avs_dist_partition(avs_ctx *) { avs::set_error("multi-GPU layer not built yet"); return AVS_ESTATE; }
avs_status avs_dist_solve(avs_ctx *, double, int32_t, avs_solve_info *) { avs::set_error("multi-GPU layer not built yet"); return AVS_ESTATE; }
avs_status avs_dist_get_solution(avs_ctx *, double *, int64_t, avs_memspace) { avs::set_error("multi-GPU layer not built yet"); return AVS_ESTATE; }
}
