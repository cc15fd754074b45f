// Instantiation unit: every particle-filter kernel of (PFG_MODEL_LGSSM, PFG_KERNEL_PRIOR).
#include "pfg_launch.hpp"

namespace pfg_host {
template int launch_mk<PFG_MODEL_LGSSM, PFG_KERNEL_PRIOR>(pfg_ctx *, int, int, int, int, int, const pfg_dev_problem *, hipStream_t);
}  // namespace pfg_host
