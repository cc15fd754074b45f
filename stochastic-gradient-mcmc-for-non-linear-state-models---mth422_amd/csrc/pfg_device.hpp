// Device code of libpfgrad: one persistent workgroup runs one whole buffered particle-filter
// window (the T-loop of particle_filters/buffered_smoother.py:93-133) in a single launch.
// gfx950 only (wave64, DPP row_bcast, 160 KiB LDS).
//   pf_reg_kernel   N <= 1024: particles / statistics / CDF in LDS, log-weights in registers
//   pf_mem_kernel   N <= 16384: CDF in LDS, particle records in an L2-resident HBM scratch
//   pf_big_kernel   N <= 16384, device generator: the fast form of pf_mem_kernel
//   pfg_grid_*      N <= 2^22: the particle axis of ONE window over all CUs, one launch per timestep
//
// The REPLAY instantiation units are compiled with -ffp-contract=off: the f64 code follows the
// reference's NumPy expression order operation by operation, so REPLAY runs differ from the
// reference only by the rounding of exp / log (LDS-table forms, <= 2 ulp), of the shift used
// by log_normalize (f32-rounded maximum, mathematically immaterial) and of the parallel weight
// sum / prefix scan.  Where the code wants a fused multiply-add it says fma().  The
// device-generator units (-ffp-contract=fast -DPFG_FAST_ALGEBRA) have no operation-order
// parity to keep.
#pragma once
#include "pfg_math.hpp"          // wave primitives, generators, table math
#include "pfg_models.hpp"        // model constants and the per-particle step
#include "pfg_reg_kernel.hpp"    // N <= 1024: LDS-resident kernel (+ PaRIS / systematic / O(N^2))
#include "pfg_mem_kernel.hpp"    // N <= 16384: general large-N kernel
#include "pfg_big_kernel.hpp"    // N <= 16384: device-generator fast path
#include "pfg_grid_kernel.hpp"   // N <= 2^22: one window over the whole GPU, one launch per timestep
#include "pfg_grid_dev_kernel.hpp"   // ... its device-generator timestep (the throughput path)
#include "pfg_grid_cdf.hpp"      // ... its REPLAY CDF: NumPy's sequential cumsum, bit for bit
