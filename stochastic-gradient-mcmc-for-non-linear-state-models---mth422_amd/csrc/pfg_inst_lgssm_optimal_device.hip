// Instantiation unit: every particle-filter kernel of (PFG_MODEL_LGSSM, PFG_KERNEL_OPTIMAL, PFG_RNG_DEVICE).
#include "pfg_launch.hpp"

namespace pfg_host {
template int launch_mkr<PFG_MODEL_LGSSM, PFG_KERNEL_OPTIMAL, PFG_RNG_DEVICE>(pfg_ctx *, int, int, int, int, const pfg_dev_problem *, hipStream_t, bool);
template int launch_grid_mkr<PFG_MODEL_LGSSM, PFG_KERNEL_OPTIMAL, PFG_RNG_DEVICE>(pfg_ctx *, int, int, int, int, const pfg_dev_problem *, hipStream_t, int);
}  // namespace pfg_host
