// Instantiation unit: every particle-filter kernel of (PFG_MODEL_SVM, PFG_KERNEL_PRIOR, PFG_RNG_REPLAY).
#include "pfg_launch.hpp"

namespace pfg_host {
template int launch_mkr<PFG_MODEL_SVM, PFG_KERNEL_PRIOR, PFG_RNG_REPLAY>(pfg_ctx *, int, int, int, int, const pfg_dev_problem *, hipStream_t, bool);
template int launch_grid_mkr<PFG_MODEL_SVM, PFG_KERNEL_PRIOR, PFG_RNG_REPLAY>(pfg_ctx *, int, int, int, int, const pfg_dev_problem *, hipStream_t, int);
}  // namespace pfg_host
