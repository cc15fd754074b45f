// libpfgrad.so: host side of the C ABI declared in include/pfgrad.h + kernel dispatch.
// The particle-filter kernels are instantiated in pfg_inst_*.hip (one unit per model x proposal
// kernel, compiled in parallel by sgmcmc_ssm_amd/_build.py); this unit holds the dispatcher, the
// small update / window / KSD kernels and the extern "C" entry points.
#include <mutex>
#include <unordered_map>
#include "pfg_host.hpp"
#include "pfg_device.hpp"
#include "pfg_elementwise.hpp"

using namespace pfg_host;

namespace pfg_host {
extern template int launch_mkr<PFG_MODEL_SVM, PFG_KERNEL_PRIOR, PFG_RNG_REPLAY>(pfg_ctx *, int, int, int, int, const pfg_dev_problem *, hipStream_t, bool);
extern template int launch_mkr<PFG_MODEL_SVM, PFG_KERNEL_PRIOR, PFG_RNG_DEVICE>(pfg_ctx *, int, int, int, int, const pfg_dev_problem *, hipStream_t, bool);
extern template int launch_mkr<PFG_MODEL_GARCH, PFG_KERNEL_PRIOR, PFG_RNG_REPLAY>(pfg_ctx *, int, int, int, int, const pfg_dev_problem *, hipStream_t, bool);
extern template int launch_mkr<PFG_MODEL_GARCH, PFG_KERNEL_PRIOR, PFG_RNG_DEVICE>(pfg_ctx *, int, int, int, int, const pfg_dev_problem *, hipStream_t, bool);
extern template int launch_mkr<PFG_MODEL_GARCH, PFG_KERNEL_OPTIMAL, PFG_RNG_REPLAY>(pfg_ctx *, int, int, int, int, const pfg_dev_problem *, hipStream_t, bool);
extern template int launch_mkr<PFG_MODEL_GARCH, PFG_KERNEL_OPTIMAL, PFG_RNG_DEVICE>(pfg_ctx *, int, int, int, int, const pfg_dev_problem *, hipStream_t, bool);
extern template int launch_mkr<PFG_MODEL_LGSSM, PFG_KERNEL_PRIOR, PFG_RNG_REPLAY>(pfg_ctx *, int, int, int, int, const pfg_dev_problem *, hipStream_t, bool);
extern template int launch_mkr<PFG_MODEL_LGSSM, PFG_KERNEL_PRIOR, PFG_RNG_DEVICE>(pfg_ctx *, int, int, int, int, const pfg_dev_problem *, hipStream_t, bool);
extern template int launch_mkr<PFG_MODEL_LGSSM, PFG_KERNEL_OPTIMAL, PFG_RNG_REPLAY>(pfg_ctx *, int, int, int, int, const pfg_dev_problem *, hipStream_t, bool);
extern template int launch_mkr<PFG_MODEL_LGSSM, PFG_KERNEL_OPTIMAL, PFG_RNG_DEVICE>(pfg_ctx *, int, int, int, int, const pfg_dev_problem *, hipStream_t, bool);

extern template int launch_grid_mkr<PFG_MODEL_SVM, PFG_KERNEL_PRIOR, PFG_RNG_REPLAY>(pfg_ctx *, int, int, int, int, const pfg_dev_problem *, hipStream_t, int);
extern template int launch_grid_mkr<PFG_MODEL_SVM, PFG_KERNEL_PRIOR, PFG_RNG_DEVICE>(pfg_ctx *, int, int, int, int, const pfg_dev_problem *, hipStream_t, int);
extern template int launch_grid_mkr<PFG_MODEL_GARCH, PFG_KERNEL_PRIOR, PFG_RNG_REPLAY>(pfg_ctx *, int, int, int, int, const pfg_dev_problem *, hipStream_t, int);
extern template int launch_grid_mkr<PFG_MODEL_GARCH, PFG_KERNEL_PRIOR, PFG_RNG_DEVICE>(pfg_ctx *, int, int, int, int, const pfg_dev_problem *, hipStream_t, int);
extern template int launch_grid_mkr<PFG_MODEL_GARCH, PFG_KERNEL_OPTIMAL, PFG_RNG_REPLAY>(pfg_ctx *, int, int, int, int, const pfg_dev_problem *, hipStream_t, int);
extern template int launch_grid_mkr<PFG_MODEL_GARCH, PFG_KERNEL_OPTIMAL, PFG_RNG_DEVICE>(pfg_ctx *, int, int, int, int, const pfg_dev_problem *, hipStream_t, int);
extern template int launch_grid_mkr<PFG_MODEL_LGSSM, PFG_KERNEL_PRIOR, PFG_RNG_REPLAY>(pfg_ctx *, int, int, int, int, const pfg_dev_problem *, hipStream_t, int);
extern template int launch_grid_mkr<PFG_MODEL_LGSSM, PFG_KERNEL_PRIOR, PFG_RNG_DEVICE>(pfg_ctx *, int, int, int, int, const pfg_dev_problem *, hipStream_t, int);
extern template int launch_grid_mkr<PFG_MODEL_LGSSM, PFG_KERNEL_OPTIMAL, PFG_RNG_REPLAY>(pfg_ctx *, int, int, int, int, const pfg_dev_problem *, hipStream_t, int);
extern template int launch_grid_mkr<PFG_MODEL_LGSSM, PFG_KERNEL_OPTIMAL, PFG_RNG_DEVICE>(pfg_ctx *, int, int, int, int, const pfg_dev_problem *, hipStream_t, int);

template <int MODEL, int KERNEL>
int launch_grid_mk(pfg_ctx *ctx, int dtype, int rng, int n_max, int t_max, int B, const pfg_dev_problem *dp, hipStream_t st, int phase) {
    if (rng == PFG_RNG_REPLAY) return launch_grid_mkr<MODEL, KERNEL, PFG_RNG_REPLAY>(ctx, dtype, n_max, t_max, B, dp, st, phase);
    return launch_grid_mkr<MODEL, KERNEL, PFG_RNG_DEVICE>(ctx, dtype, n_max, t_max, B, dp, st, phase);
}

template <int MODEL, int KERNEL>
int launch_mk(pfg_ctx *ctx, int dtype, int rng, int v, int n_max, int B, const pfg_dev_problem *dp, hipStream_t st, bool traced) {
    if (rng == PFG_RNG_REPLAY) return launch_mkr<MODEL, KERNEL, PFG_RNG_REPLAY>(ctx, dtype, v, n_max, B, dp, st, traced);
    return launch_mkr<MODEL, KERNEL, PFG_RNG_DEVICE>(ctx, dtype, v, n_max, B, dp, st, traced);
}
}  // namespace pfg_host

namespace {

// ---- kernel variants ----------------------------------------------------------------
// pp = ping-pong LDS state buffers (3 barriers/step); single buffer fits larger N (4 barriers).
struct Variant { int NT, PPT; bool pp; const char *tag; };
const Variant kVariants[] = { {256, 1, true, "wg256x1"}, {256, 4, true, "wg256x4"}, {256, 4, false, "wg256x4s"},
                              // latency variant: one particle per thread, 16 waves on one CU; picked for
                              // small batches (fewer windows than a quarter of the CUs), never by order
                              {1024, 1, true, "wg1024x1"},
                              // 1024 < N <= 4096 with the device generator when the state fits LDS
                              // (32-bit CDF): SVM fp64, every model in f32
                              {1024, 4, false, "wg1024x4s"},
                              // N <= 128: one wave per window (barriers and cross-wave reductions degenerate)
                              {64, 2, true, "wg64x2"},
                              // GARCH fp64: six LDS arrays allow two workgroups per CU; eight waves each
                              // put four waves on a SIMD (256x4: two)
                              {512, 2, false, "wg512x2s"},
                              // 128 < N <= 256, many windows: still one wave per window, four particles per lane
                              {64, 4, true, "wg64x4"},
                              // one wave per window, single state buffer
                              {64, 2, false, "wg64x2s"}, {64, 4, false, "wg64x4s"} };
// Measured and not kept (round 3, BASELINE config 4, 512 chains, ms per launch): the 4096 LDS slots of N <= 4096 in
// fewer, wider threads -- 512 x 8 (2 waves per SIMD, 251 VGPRs, no spills) 12.99, 256 x 16 (1 wave per SIMD, 256 VGPRs +
// 176 AGPRs) 16.14, against 12.73 for 1024 x 4 at its 128-VGPR cap (16 spilled VGPRs): LDS holds ONE such workgroup per
// CU, so its own 16 waves are all the latency hiding a CU has, and they are worth more than the registers.
constexpr int kLds4096Variant = 4, kTinyVariant = 5, kGarchVariant = 6, kTiny4Variant = 7;
// device generator: the one-wave variants on a single state buffer (half the LDS per window: 18 instead of 11 windows
// per CU for LGSSM N = 100; 16384 windows of BASELINE config 1 in 1.86 instead of 2.25 ms)
constexpr int kTinySingleVariant = 8, kTiny4SingleVariant = 9;
constexpr int kLatencyVariant = 3, kLatencyBatch = 64;
constexpr int kNumVariants = (int)(sizeof(kVariants) / sizeof(kVariants[0]));

int state_dim(int model) { return model == PFG_MODEL_GARCH ? 2 : 1; }
int stat_dim(int model) { return model == PFG_MODEL_SVM ? 3 : 4; }
int theta_dim(int model) { return model == PFG_MODEL_SVM ? 3 : 4; }

size_t lds_bytes(int model, int dtype, int rng, const Variant &v, int N) {
    size_t rs = dtype == PFG_F64 ? 8 : 4;
    const bool fast = pfg::fast_layout(v.NT, v.pp);
    size_t NL = fast ? (size_t)v.NT * v.PPT : (size_t)(N + 63) / 64 * 64;
    size_t NC = fast ? (size_t)v.NT * v.PPT + (size_t)v.NT * v.PPT / 32 : NL;
    size_t red = (size_t)v.PPT * (v.NT / 64) + (v.NT / 64) + (size_t)PFG_MAX_STAT * (v.NT / 64) + 8;
    size_t tab = (fast && dtype == PFG_F64)
                     ? 8 * (size_t)(pfg::tab_doubles_exp(rng == PFG_RNG_DEVICE) + (rng == PFG_RNG_DEVICE ? pfg::TAB_DOUBLES_RNG : 0)) : 0;
    const bool blk = fast && rng == PFG_RNG_DEVICE;        // 32-bit fixed-point CDF (pfg::pf_reg_kernel)
    const size_t NLS = NL + ((fast && rng == PFG_RNG_DEVICE) ? (size_t)(PFG_OPT_PADSTATE ? 8 / rs : 0) : 0);      // pfg::state_pad
    return (NC * (blk ? 4 : 8) + 15) / 16 * 16 + (v.pp ? 2 : 1) * NLS * (state_dim(model) + stat_dim(model)) * rs + red * 8 + tab;
}

// index into kVariants, or -1 when no LDS-resident variant fits.  PFGRAD_VARIANT=<tag> forces a
// variant (tuning / tests) when it can hold n_max.
int pick_variant(int model, int dtype, int rng, int n_max, int batch = 1 << 30) {
    if (const char *force = std::getenv("PFGRAD_VARIANT")) {
        if (!std::strcmp(force, "mem1024") && n_max <= pfg::MEM_MAX_N) return kVariantMem;
        // "big": the large-N kernels also where an LDS-resident variant would fit (A/B timing)
        if (!std::strcmp(force, "big") && rng == PFG_RNG_DEVICE && n_max > 1024 && n_max <= pfg::MEM_MAX_N) return kVariantMem;
        for (int v = 0; v < kNumVariants; ++v)
            if (!std::strcmp(force, kVariants[v].tag) && n_max <= kVariants[v].NT * kVariants[v].PPT &&
                lds_bytes(model, dtype, rng, kVariants[v], n_max) <= kLdsLimit)
                return v;
    }
    // preference order: fp64 N<=1024 runs best on the single-buffer 256x4 variant at 3
    // workgroups per CU; f32 on ping-pong.  N > 1024 goes to the large-N kernel: 1024-thread
    // register-resident variants spill at the 128-VGPR cap and measured 3-5x slower than it.
    // N <= 128, many windows: one wave per window (2048 LGSSM N=100 T=200 chains: 1.07 -> 0.57 ms);
    // a lone window is quicker on the four waves of wg256x1 (0.34 vs 0.38 ms)
    if (n_max <= 128 && batch > kLatencyBatch &&
        lds_bytes(model, dtype, rng, kVariants[kTinyVariant], n_max) <= kLdsLimit)
        return rng == PFG_RNG_DEVICE ? kTinySingleVariant : kTinyVariant;
    if (n_max > 128 && n_max <= 256 && batch > kLatencyBatch && rng == PFG_RNG_DEVICE &&
        lds_bytes(model, dtype, rng, kVariants[kTiny4Variant], n_max) <= kLdsLimit)
        return kTiny4SingleVariant;
    if (batch <= kLatencyBatch && n_max > 256 && n_max <= 1024 &&
        lds_bytes(model, dtype, rng, kVariants[kLatencyVariant], n_max) <= kLdsLimit)
        return kLatencyVariant;
    // GARCH fp64, device generator, 256 < N <= 1024: LDS holds two workgroups per CU either way; 512 threads x 2
    // particles put four waves on a SIMD instead of two (8192 windows of config 3: 1.99 -> 1.87 ms)
    if (model == PFG_MODEL_GARCH && dtype == PFG_F64 && rng == PFG_RNG_DEVICE && n_max > 256 && n_max <= 1024 &&
        lds_bytes(model, dtype, rng, kVariants[kGarchVariant], n_max) <= kLdsLimit)
        return kGarchVariant;
    const int order_f64[] = {0, 2, 1}, order_f32[] = {0, 1, 2};
    const int *order = dtype == PFG_F64 ? order_f64 : order_f32;
    for (int oi = 0; oi < 3; ++oi) {
        const int v = order[oi];
        if (n_max <= kVariants[v].NT * kVariants[v].PPT && lds_bytes(model, dtype, rng, kVariants[v], n_max) <= kLdsLimit)
            return v;
    }
    if (rng == PFG_RNG_DEVICE && n_max <= 4096 &&
        lds_bytes(model, dtype, rng, kVariants[kLds4096Variant], n_max) <= kLdsLimit)
        return kLds4096Variant;
    if (n_max <= pfg::MEM_MAX_N) return kVariantMem;
    if (n_max <= pfg::GRID_MAX_N) return kVariantGrid;     // one window over the whole GPU (pfg_grid_kernel.hpp)
    return -1;
}

size_t grid_scratch_bytes(int model, int dtype, int rng, int N) {
    const bool rp = rng == PFG_RNG_REPLAY;
    if (dtype == PFG_F64)
        return model == PFG_MODEL_SVM ? pfg::grid_layout<PFG_MODEL_SVM, double>(N, rp).bytes
               : model == PFG_MODEL_GARCH ? pfg::grid_layout<PFG_MODEL_GARCH, double>(N, rp).bytes
                                          : pfg::grid_layout<PFG_MODEL_LGSSM, double>(N, rp).bytes;
    return model == PFG_MODEL_SVM ? pfg::grid_layout<PFG_MODEL_SVM, float>(N, rp).bytes
           : model == PFG_MODEL_GARCH ? pfg::grid_layout<PFG_MODEL_GARCH, float>(N, rp).bytes
                                      : pfg::grid_layout<PFG_MODEL_LGSSM, float>(N, rp).bytes;
}

size_t scratch_bytes(int model, int dtype, int N, bool paris = false) {
    const size_t rs = dtype == PFG_F64 ? 8 : 4, per = 16 / rs;
    const size_t rec = (state_dim(model) + stat_dim(model) + per - 1) / per * per;   // pfg::mem_rec_len
    return (size_t)N * rs * (1 + 2 * rec) + 16 + (paris ? (size_t)N * (2 * rs + 8) + 16 + 2 * (size_t)((N + 1023) / 1024 * 1024) * 4 : 0);   // pfg::mem_kernel_scratch_bytes
}

int check_combo(pfg_ctx *ctx, int model, int kernel, int dtype, int rng) {
    if (model < 0 || model > 2) return fail(ctx, PFG_ERR_INVALID, "Unrecognized model id");
    if (kernel != PFG_KERNEL_PRIOR && kernel != PFG_KERNEL_OPTIMAL)
        return fail(ctx, PFG_ERR_INVALID, "Unrecoginized kernel id");
    if (model == PFG_MODEL_SVM && kernel == PFG_KERNEL_OPTIMAL)
        return fail(ctx, PFG_ERR_UNSUPPORTED, "SVM optimal kernel not analytic");   // svm/helper.py:62
    if (dtype != PFG_F64 && dtype != PFG_F32) return fail(ctx, PFG_ERR_INVALID, "bad dtype");
    if (rng != PFG_RNG_REPLAY && rng != PFG_RNG_DEVICE) return fail(ctx, PFG_ERR_INVALID, "bad rng mode");
    return PFG_OK;
}

// traced: the descriptors may carry trace_* / rec_* buffers (the plain LDS-resident kernels, device generator and
// REPLAY, exist as a production twin that ignores them, see pfg_reg_kernel.hpp; every other kernel always honours them)
int dispatch(pfg_ctx *ctx, int model, int kernel, int dtype, int rng, int n_max, int B,
             const pfg_dev_problem *dp, hipStream_t st, int smoother = PFG_SMOOTHER_NEMETH,
             bool force_mem = false, bool traced = true) {
    int rc = check_combo(ctx, model, kernel, dtype, rng);
    if (rc) return rc;
    if (B <= 0) return PFG_OK;
    if (n_max < 1) return fail(ctx, PFG_ERR_INVALID, "N must be >= 1");
    // PFG_SMOOTHER_POYIADJIS_N: the caller states that every descriptor is (NEMETH, lambduh = 1, score) -- same kernels as
    // NEMETH, except where a unit has a twin specialised to that estimator (launch_one)
    ctx->score1 = (smoother == PFG_SMOOTHER_POYIADJIS_N) && !traced;
    if (ctx->score1) {
        const char *off = std::getenv("PFGRAD_NO_SCORE1");      // A/B timing: the general kernel for these launches too
        if (off && off[0] == '1') ctx->score1 = false;
    }
    if (smoother == PFG_SMOOTHER_POYIADJIS_N) smoother = PFG_SMOOTHER_NEMETH;
    int v = smoother == PFG_SMOOTHER_PARIS ? kVariantParis
            : smoother == PFG_SMOOTHER_NEMETH_SYSTEMATIC ? kVariantSystematic
            : smoother == PFG_SMOOTHER_POYIADJIS_N2 ? kVariantN2
            : (force_mem && n_max <= pfg::MEM_MAX_N) ? kVariantMem
            : pick_variant(model, dtype, rng, n_max, B);
    if (v == -1)
        return fail(ctx, PFG_ERR_UNSUPPORTED,
                    "N = " + std::to_string(n_max) + " exceeds the supported maximum of " + std::to_string(pfg::GRID_MAX_N));
    if (v == kVariantGrid)
        return fail(ctx, PFG_ERR_UNSUPPORTED,
                    "N = " + std::to_string(n_max) + " > " + std::to_string(pfg::MEM_MAX_N) +
                    " runs as a whole-GPU window, one launch per timestep: use pfg_launch_device_grid (it needs T_max)");
    // N > 1024 with the device generator: the fast large-N kernel, unless the statistic needs the
    // general one (predictive) or PFGRAD_VARIANT=mem1024 asks for it (A/B timing, tests)
    if (v == kVariantMem && rng == PFG_RNG_DEVICE && !force_mem) {
        const char *force = std::getenv("PFGRAD_VARIANT");
        if (!(force && !std::strcmp(force, "mem1024"))) v = kVariantBig;
    }
    // the large-N kernel with the log-weights in registers: N <= 4096, and the host knows that no window asks for the
    // predictive statistic (pfg_run_batch routes those with force_mem; resident descriptors never carry it)
    if (v == kVariantMem && !force_mem && n_max <= 4096) v = kVariantMemLw4;
    ctx->last_variant = v >= 0 ? kVariants[v].tag
                        : (v == kVariantMem || v == kVariantMemLw4) ? "mem1024"
                        : v == kVariantBig ? (n_max <= 4096 ? "big4096" : "big16384")
                        : v == kVariantParis ? (n_max <= 256 ? "paris256x1" : n_max <= 1024 ? "paris256x4" : "paris_mem1024")
                        : v == kVariantSystematic ? "systematic256x4"
                        : (n_max <= 256 ? "n2_256x1" : n_max <= 1024 ? "n2_256x4" : "n2_mem1024");
    ctx->last_traced = traced || v < 0;
    if (model == PFG_MODEL_SVM) return launch_mk<PFG_MODEL_SVM, PFG_KERNEL_PRIOR>(ctx, dtype, rng, v, n_max, B, dp, st, traced);
    if (model == PFG_MODEL_GARCH) {
        if (kernel == PFG_KERNEL_PRIOR) return launch_mk<PFG_MODEL_GARCH, PFG_KERNEL_PRIOR>(ctx, dtype, rng, v, n_max, B, dp, st, traced);
        return launch_mk<PFG_MODEL_GARCH, PFG_KERNEL_OPTIMAL>(ctx, dtype, rng, v, n_max, B, dp, st, traced);
    }
    if (kernel == PFG_KERNEL_PRIOR) return launch_mk<PFG_MODEL_LGSSM, PFG_KERNEL_PRIOR>(ctx, dtype, rng, v, n_max, B, dp, st, traced);
    return launch_mk<PFG_MODEL_LGSSM, PFG_KERNEL_OPTIMAL>(ctx, dtype, rng, v, n_max, B, dp, st, traced);
}

// whole-GPU windows (N above the one-workgroup kernels' maximum, or forced): NEMETH / FILTER with the score,
// sufficient or no statistic
int dispatch_grid(pfg_ctx *ctx, int model, int kernel, int dtype, int rng, int n_max, int t_max, int B,
                  const pfg_dev_problem *dp, hipStream_t st, int phase = -1, int smoother = PFG_SMOOTHER_NEMETH) {
    int rc = check_combo(ctx, model, kernel, dtype, rng);
    if (rc) return rc;
    if (B <= 0) return PFG_OK;
    if (n_max < 1) return fail(ctx, PFG_ERR_INVALID, "N must be >= 1");
    if (t_max < 0) return fail(ctx, PFG_ERR_INVALID, "T_max must be >= 0");
    if (B > 65535) return fail(ctx, PFG_ERR_INVALID, "at most 65535 whole-GPU windows per launch");
    ctx->score1 = smoother == PFG_SMOOTHER_POYIADJIS_N && rng == PFG_RNG_DEVICE;
    if (ctx->score1) {
        const char *off = std::getenv("PFGRAD_NO_SCORE1");      // A/B timing: the general kernel for these launches too
        if (off && off[0] == '1') ctx->score1 = false;
    }
    ctx->last_variant = pfg::grid_ppt(n_max) == 8 ? (ctx->score1 ? "grid2048_score1" : "grid2048") : (ctx->score1 ? "grid1024_score1" : "grid1024");
    ctx->last_traced = true;
    if (model == PFG_MODEL_SVM) return launch_grid_mk<PFG_MODEL_SVM, PFG_KERNEL_PRIOR>(ctx, dtype, rng, n_max, t_max, B, dp, st, phase);
    if (model == PFG_MODEL_GARCH) {
        if (kernel == PFG_KERNEL_PRIOR) return launch_grid_mk<PFG_MODEL_GARCH, PFG_KERNEL_PRIOR>(ctx, dtype, rng, n_max, t_max, B, dp, st, phase);
        return launch_grid_mk<PFG_MODEL_GARCH, PFG_KERNEL_OPTIMAL>(ctx, dtype, rng, n_max, t_max, B, dp, st, phase);
    }
    if (kernel == PFG_KERNEL_PRIOR) return launch_grid_mk<PFG_MODEL_LGSSM, PFG_KERNEL_PRIOR>(ctx, dtype, rng, n_max, t_max, B, dp, st, phase);
    return launch_grid_mk<PFG_MODEL_LGSSM, PFG_KERNEL_OPTIMAL>(ctx, dtype, rng, n_max, t_max, B, dp, st, phase);
}

// ---- SGLD update for resident chains ---------------------------------------------------
__device__ __forceinline__ double reflect_chol(double L) { return L < 0.0 ? sqrt(L * L + 1e-16) : L; }

// momentum == nullptr: SGLD.  Otherwise SGHMC with friction alpha: the increment d of each
// variable becomes v <- (1 - alpha) v + drift + sqrt(alpha) * noise (noise ~ N(0, 2 eps / T)).
__global__ void sgld_update_kernel(int model, int B, double *__restrict__ theta,
                                   const double *__restrict__ outs, pfg_prior_hyper hy, double eps,
                                   double Tscale, uint64_t seed, uint64_t chain_offset,
                                   const uint64_t *step_ctr, double *__restrict__ momentum, double alpha) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double *th = theta + (size_t)b * PFG_MAX_THETA;
    const double *g = outs + (size_t)b * PFG_OUT_DOUBLES;
    const uint64_t step = step_ctr ? *step_ctr : 0ull;
    const uint64_t gid = chain_offset + (uint64_t)b;
    const uint32_t c1 = (uint32_t)step, c2 = (uint32_t)(step >> 32) ^ (uint32_t)(gid >> 32);
    pfg::u32x4 r0 = pfg::philox4x32_10({(uint32_t)gid, c1, c2, 0x5A11u}, (uint32_t)seed, (uint32_t)(seed >> 32));
    pfg::u32x4 r1 = pfg::philox4x32_10({(uint32_t)gid, c1, c2, 0x5A12u}, (uint32_t)seed, (uint32_t)(seed >> 32));
    const double nsd = sqrt(1.0 / Tscale) * sqrt(2.0 * eps) * (momentum ? sqrt(alpha) : 1.0);
    double *mv = momentum ? momentum + (size_t)b * PFG_MAX_THETA : nullptr;
    // one variable's increment: SGLD drift + noise, or the SGHMC momentum recursion
    auto incr = [&](int slot, double drift, double noise) {
        const double d = drift + noise;
        if (!mv) return d;
        const double v = (1.0 - alpha) * mv[slot] + d;
        mv[slot] = v;
        return v;
    };
    double nz[4];
    const pfg::Math<double, false> mth = {};
    mth.normal_pair(r0.x, r0.y, nz[0], nz[1]);
    mth.normal_pair(r1.x, r1.y, nz[2], nz[3]);
    if (model == PFG_MODEL_SVM || model == PFG_MODEL_LGSSM) {
        const bool lg = model == PFG_MODEL_LGSSM;
        double A = th[0], C = lg ? th[1] : 1.0, LQ = th[lg ? 2 : 1], LR = th[lg ? 3 : 2];
        double Qinv = LQ * LQ + 1e-16, Rinv = LR * LR + 1e-16;
        // score columns: SVM [LR, LQ, A]; LGSSM [LR, LQ, C, A]
        double gLR = g[0], gLQ = g[1], gC = lg ? g[2] : 0.0, gA = g[lg ? 3 : 2];
        // grad_logprior: covariance.py:272-284 (n = 1), matrices.py:597-607
        double pLQ = (hy.df_Qinv - 2.0) / LQ - LQ / hy.scale_Qinv;
        double pLR = (hy.df_Rinv - 2.0) / LR - LR / hy.scale_Rinv;
        double pA = -1.0 * (Qinv * (A - hy.mean_A)) / hy.var_col_A;
        double pC = -1.0 * (Rinv * (C - hy.mean_C)) / hy.var_col_C;
        int j = 0;
        A += incr(0, eps * ((pA + gA) / Tscale), nsd * nz[j]); ++j;
        if (lg) { C += incr(1, eps * ((pC + gC) / Tscale), nsd * nz[j]); ++j; }
        LQ += incr(lg ? 2 : 1, eps * ((pLQ + gLQ) / Tscale), nsd * nz[j]); ++j;
        LR += incr(lg ? 3 : 2, eps * ((pLR + gLR) / Tscale), nsd * nz[j]); ++j;
        // project_parameters: _utils.py:165-170, covariance.py:68-80, lgssm/parameters.py:39-42
        double aa = fabs(A);
        if (aa > 0.9999) A *= 0.9999 / aa;
        if (lg) C = 1.0;
        LQ = reflect_chol(LQ); LR = reflect_chol(LR);
        th[0] = A;
        if (lg) { th[1] = C; th[2] = LQ; th[3] = LR; } else { th[1] = LQ; th[2] = LR; }
    } else {
        double lmu = th[0], lphi = th[1], llam = th[2], LR = th[3];
        double mu = exp(lmu), phi = 1.0 / (1.0 + exp(-lphi)), lam = 1.0 / (1.0 + exp(-llam));
        // garch_var.py:152-165
        double p0 = -hy.shape_mu - 1.0 + hy.scale_mu / mu;
        double p1 = ((hy.alpha_phi - 1.0) / (1.0 + phi) - (hy.beta_phi - 1.0) / (1.0 - phi)) * phi * (1.0 - phi);
        double p2 = ((hy.alpha_lambduh - 1.0) / (1.0 + lam) - (hy.beta_lambduh - 1.0) / (1.0 - lam)) * lam * (1.0 - lam);
        double pLR = (hy.df_Rinv - 2.0) / LR - LR / hy.scale_Rinv;
        // score columns [LR, log_mu, logit_phi, logit_lambduh]
        lmu += incr(0, eps * ((p0 + g[1]) / Tscale), nsd * nz[0]);
        lphi += incr(1, eps * ((p1 + g[2]) / Tscale), nsd * nz[1]);
        llam += incr(2, eps * ((p2 + g[3]) / Tscale), nsd * nz[2]);
        LR += incr(3, eps * ((pLR + g[0]) / Tscale), nsd * nz[3]);
        th[0] = lmu; th[1] = lphi; th[2] = llam; th[3] = reflect_chol(LR);
    }
}

__global__ void bump_counter_kernel(uint64_t *ctr) { *ctr += 1; }

// window starts for resident chains (see pfg_sample_windows_device)
__global__ void sample_windows_kernel(int B, pfg_dev_problem *__restrict__ probs, const double *__restrict__ y,
                                      const double *__restrict__ wtab, int T, int S, int buffer, int strict,
                                      uint64_t seed, uint64_t chain_offset, const uint64_t *__restrict__ step_ctr) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const uint64_t chain = chain_offset + (uint64_t)b, ctr = step_ctr ? *step_ctr : 0ull;
    const pfg::u32x4 r = pfg::philox4x32_10({(uint32_t)chain, (uint32_t)(chain >> 32), (uint32_t)ctr, (uint32_t)(ctr >> 32)},
                                            (uint32_t)seed ^ 0x57494E44u, (uint32_t)(seed >> 32));   // "WIND"
    const uint32_t range = strict ? (uint32_t)(T / S) : (uint32_t)(T - S + 1);
    // 64 random bits times the range, high part: bias < range / 2^64
    const uint64_t bits = ((uint64_t)r.x << 32) | r.y;
    const int idx = (int)__umul64hi(bits, (uint64_t)range);
    const int start = strict ? idx * S : idx;
    const int left = start - buffer > 0 ? start - buffer : 0;
    const int right = start + S + buffer < T ? start + S + buffer : T;
    pfg_dev_problem &P = probs[b];
    P.y = y + left;
    P.T = right - left;
    P.t1 = start - left;
    P.tL = start + S - left;
    P.weights = wtab ? wtab + (size_t)start * S : nullptr;
}

// ---- IMQ kernel Stein discrepancy: all K^2 pairs, row i per workgroup-stride, f64 ----------
constexpr int KSD_MAX_D = 8;
__global__ __launch_bounds__(256) void imq_ksd_kernel(int K, int d, const double *__restrict__ x,
                                                      const double *__restrict__ g, double c2, double beta,
                                                      double *__restrict__ partial) {
    __shared__ double red[4];
    double acc = 0.0;
    for (int i = blockIdx.x; i < K; i += gridDim.x) {
        double xi[KSD_MAX_D], gi[KSD_MAX_D];
        for (int k = 0; k < d; ++k) { xi[k] = x[(size_t)i * d + k]; gi[k] = g[(size_t)i * d + k]; }
        for (int j = threadIdx.x; j < K; j += blockDim.x) {
            double diff2 = 0.0, gg = 0.0, g0d = 0.0, g1d = 0.0;
            for (int k = 0; k < d; ++k) {
                const double df = xi[k] - x[(size_t)j * d + k];
                const double gj = g[(size_t)j * d + k];
                diff2 += df * df; gg += gi[k] * gj; g0d += gi[k] * -df; g1d += gj * df;
            }
            const double base = diff2 + c2;
            const double bb = pow(base, -beta);
            const double coeff = -2.0 * beta * (bb / base);
            acc += gg * bb + g0d * coeff + g1d * coeff + (-(double)d + 2.0 * (beta + 1.0) * diff2 / base) * coeff;
        }
    }
    acc = pfg::wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ---- elementwise sufficient statistics: second pass over the recorded trajectory (pfg_elementwise.hpp) ----
int elementwise_pass(pfg_ctx *ctx, const pfg_problem &q, const double *theta_dev, const double *tx, const double *tlw, const int32_t *par, int Nt,
                     size_t Wd, double *S0, double *S1, double *Sbar, double *w, double *mean, double *stats) {
    const int N = q.N, T = q.T, NS = state_dim(q.model);
    const int tL = q.tL < q.T ? q.tL : q.T;
    const double lam = q.smoother == PFG_SMOOTHER_NEMETH ? q.lambduh : 1.0;
    hipStream_t st = ctx->stream;
    PFG_HIP(ctx, hipMemsetAsync(S0, 0, (size_t)N * Wd * 8, st));
    double *cur = S0, *nxt = S1;
    const dim3 cgrid((unsigned)((Wd + 255) / 256)), sgrid((unsigned)((Wd + 255) / 256), (unsigned)N);
    for (int t = 0; t < T; ++t) {
        if (lam != 1.0) {
            hipLaunchKernelGGL(pfg::ews_softmax_kernel, dim3(1), dim3(1024), 0, st, N, tlw + (size_t)t * N, w);
            hipLaunchKernelGGL(pfg::ews_colsum_kernel, cgrid, dim3(256), 0, st, N, (int)Wd, cur, w, Sbar);
        }
        const bool inside = t >= q.t1 && t < tL;
        const int col0 = inside ? 3 * (t - q.t1) : -1;
        const double wt = (inside && q.weights) ? q.weights[t - q.t1] : 1.0;
        const int32_t *pt = par + (size_t)t * Nt * N;
        const double *xt = tx + (size_t)t * N * NS, *xn = tx + (size_t)(t + 1) * N * NS;
        if (q.smoother == PFG_SMOOTHER_POYIADJIS_N2) {
            const double *lwt = tlw + (size_t)t * N;
            if (q.model == PFG_MODEL_GARCH)
                hipLaunchKernelGGL(pfg::ews_n2_step_kernel<PFG_MODEL_GARCH>, dim3(N), dim3(256), 0, st, N, (int)Wd, wt, col0, theta_dev, xt, lwt, xn, cur, nxt);
            else if (q.model == PFG_MODEL_LGSSM)
                hipLaunchKernelGGL(pfg::ews_n2_step_kernel<PFG_MODEL_LGSSM>, dim3(N), dim3(256), 0, st, N, (int)Wd, wt, col0, theta_dev, xt, lwt, xn, cur, nxt);
            else
                hipLaunchKernelGGL(pfg::ews_n2_step_kernel<PFG_MODEL_SVM>, dim3(N), dim3(256), 0, st, N, (int)Wd, wt, col0, theta_dev, xt, lwt, xn, cur, nxt);
        } else if (q.model == PFG_MODEL_GARCH)
            hipLaunchKernelGGL(pfg::ews_step_kernel<PFG_MODEL_GARCH>, sgrid, dim3(256), 0, st, N, (int)Wd, Nt, lam, wt, col0, pt, xt, xn, Sbar, cur, nxt);
        else
            hipLaunchKernelGGL(pfg::ews_step_kernel<PFG_MODEL_SVM>, sgrid, dim3(256), 0, st, N, (int)Wd, Nt, lam, wt, col0, pt, xt, xn, Sbar, cur, nxt);
        double *tmp = cur; cur = nxt; nxt = tmp;
    }
    hipLaunchKernelGGL(pfg::ews_softmax_kernel, dim3(1), dim3(1024), 0, st, N, tlw + (size_t)T * N, w);
    hipLaunchKernelGGL(pfg::ews_colsum_kernel, cgrid, dim3(256), 0, st, N, (int)Wd, cur, w, mean);
    if (stats) PFG_HIP(ctx, hipMemcpyAsync(stats, cur, (size_t)N * Wd * 8, hipMemcpyDeviceToDevice, st));
    PFG_HIP(ctx, hipGetLastError());
    return PFG_OK;
}

}  // namespace

// ======================================================================================
// C ABI
// ======================================================================================
// ---- caller-registered pinned host ranges (pfg_host_register) -------------------------------------
namespace {
constexpr size_t kDirectMinDoubles = (size_t)1 << 16;     // inputs at least this long are staged from registered pages directly
struct HostRange { const char *lo, *hi; };
std::mutex g_host_mu;
std::vector<HostRange> g_host_ranges;
bool host_registered(const void *p, size_t bytes) {
    const char *a = static_cast<const char *>(p);
    std::lock_guard<std::mutex> lk(g_host_mu);
    for (const HostRange &r : g_host_ranges)
        if (a >= r.lo && a + bytes <= r.hi) return true;
    return false;
}
}  // namespace

extern "C" {

int pfg_version(void) { return PFG_VERSION; }

int pfg_host_register(void *ptr, size_t bytes) {
    if (!ptr || bytes == 0) return PFG_ERR_INVALID;
    {
        std::lock_guard<std::mutex> lk(g_host_mu);
        const char *a = static_cast<const char *>(ptr);
        for (const HostRange &r : g_host_ranges)
            if (a < r.hi && a + bytes > r.lo) return PFG_ERR_INVALID;      // overlaps a registered range
    }
    if (hipHostRegister(ptr, bytes, hipHostRegisterPortable) != hipSuccess) {
        (void)hipGetLastError();
        return PFG_ERR_DEVICE;
    }
    std::lock_guard<std::mutex> lk(g_host_mu);
    g_host_ranges.push_back({static_cast<const char *>(ptr), static_cast<const char *>(ptr) + bytes});
    return PFG_OK;
}

int pfg_host_unregister(void *ptr) {
    {
        std::lock_guard<std::mutex> lk(g_host_mu);
        auto it = g_host_ranges.begin();
        for (; it != g_host_ranges.end(); ++it)
            if (it->lo == static_cast<const char *>(ptr)) break;
        if (it == g_host_ranges.end()) return PFG_ERR_INVALID;
        g_host_ranges.erase(it);
    }
    if (hipHostUnregister(ptr) != hipSuccess) {
        (void)hipGetLastError();
        return PFG_ERR_DEVICE;
    }
    return PFG_OK;
}

int pfg_struct_size(int which) {
    switch (which) {
        case 0: return (int)sizeof(pfg_problem);
        case 1: return (int)sizeof(pfg_result);
        case 2: return (int)sizeof(pfg_dev_problem);
        case 3: return (int)sizeof(pfg_prior_hyper);
    }
    return -1;
}

const char *pfg_last_error(pfg_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int pfg_create(pfg_ctx **out, int device_id) {
    if (!out) return fail(nullptr, PFG_ERR_INVALID, "pfg_create: out is NULL");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, PFG_ERR_DEVICE, std::string("no HIP device available: ") + hipGetErrorString(e));
    if (device_id < 0 || device_id >= ndev) return fail(nullptr, PFG_ERR_INVALID, "pfg_create: bad device id");
    pfg_ctx *ctx = new (std::nothrow) pfg_ctx();
    if (!ctx) return fail(nullptr, PFG_ERR_NOMEM, "pfg_create: out of host memory");
    ctx->device = device_id;
    if ((e = hipSetDevice(device_id)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) {
        std::string m = std::string("pfg_create: ") + hipGetErrorString(e);
        delete ctx;
        return fail(nullptr, PFG_ERR_DEVICE, m);
    }
    *out = ctx;
    return PFG_OK;
}

void pfg_destroy(pfg_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) { (void)hipStreamSynchronize(ctx->stream); (void)hipStreamDestroy(ctx->stream); }
    ctx->in.release(); ctx->out.release(); ctx->desc.release(); ctx->scratch.release(); ctx->work.release(); ctx->h_in.release(); ctx->h_out.release();
    delete ctx;
}

const char *pfg_last_variant(pfg_ctx *ctx) { return ctx ? ctx->last_variant : "none"; }

int pfg_last_traced(pfg_ctx *ctx) { return ctx ? (ctx->last_traced ? 1 : 0) : -1; }

void *pfg_ctx_stream(pfg_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

int pfg_synchronize(pfg_ctx *ctx) {
    if (!ctx) return PFG_ERR_INVALID;
    PFG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PFG_OK;
}

int64_t pfg_scratch_bytes(int model, int dtype, int rng, int N) {
    if (model < 0 || model > 2 || N < 1) return -1;
    const int v = pick_variant(model, dtype, rng, N);
    if (v >= 0) return 0;
    if (v == kVariantMem) return (int64_t)((scratch_bytes(model, dtype, N) + 255) / 256 * 256);
    if (v == kVariantGrid) return (int64_t)grid_scratch_bytes(model, dtype, rng, N);
    return -1;
}

const char *pfg_variant_name(int model, int kernel, int dtype, int rng, int n_max) {
    (void)kernel; (void)rng;
    int v = pick_variant(model, dtype, rng, n_max);
    if (v == kVariantMem && rng == PFG_RNG_DEVICE) {
        const char *force = std::getenv("PFGRAD_VARIANT");
        if (!(force && !std::strcmp(force, "mem1024"))) return n_max <= 4096 ? "big4096" : "big16384";
    }
    if (v == kVariantGrid) return pfg::grid_ppt(n_max) == 8 ? "grid2048" : "grid1024";
    return v == kVariantMem ? "mem1024" : (v < 0 ? "none" : kVariants[v].tag);
}

int pfg_launch_device(pfg_ctx *ctx, int model, int kernel, int dtype, int rng, int n_max, int B,
                      const pfg_dev_problem *dev_probs, void *hip_stream) {
    if (!ctx) return PFG_ERR_INVALID;
    if (!dev_probs && B > 0) return fail(ctx, PFG_ERR_INVALID, "pfg_launch_device: dev_probs is NULL");
    PFG_HIP(ctx, hipSetDevice(ctx->device));
    return dispatch(ctx, model, kernel, dtype, rng, n_max, B, dev_probs, (hipStream_t)hip_stream, PFG_SMOOTHER_NEMETH, false, false);
}

int pfg_launch_device_traced(pfg_ctx *ctx, int model, int kernel, int dtype, int rng, int smoother, int n_max, int B,
                             const pfg_dev_problem *dev_probs, void *hip_stream) {
    if (!ctx) return PFG_ERR_INVALID;
    if (!dev_probs && B > 0) return fail(ctx, PFG_ERR_INVALID, "pfg_launch_device_traced: dev_probs is NULL");
    if (smoother < PFG_SMOOTHER_NEMETH || smoother > PFG_SMOOTHER_POYIADJIS_N)
        return fail(ctx, PFG_ERR_INVALID, "Unrecognized pf (smoother id)");
    PFG_HIP(ctx, hipSetDevice(ctx->device));
    return dispatch(ctx, model, kernel, dtype, rng, n_max, B, dev_probs, (hipStream_t)hip_stream, smoother, false, true);
}

int pfg_launch_device_smoother(pfg_ctx *ctx, int model, int kernel, int dtype, int rng, int smoother, int n_max,
                               int B, const pfg_dev_problem *dev_probs, void *hip_stream) {
    if (!ctx) return PFG_ERR_INVALID;
    if (!dev_probs && B > 0) return fail(ctx, PFG_ERR_INVALID, "pfg_launch_device_smoother: dev_probs is NULL");
    if (smoother < PFG_SMOOTHER_NEMETH || smoother > PFG_SMOOTHER_POYIADJIS_N)
        return fail(ctx, PFG_ERR_INVALID, "Unrecognized pf (smoother id)");
    PFG_HIP(ctx, hipSetDevice(ctx->device));
    return dispatch(ctx, model, kernel, dtype, rng, n_max, B, dev_probs, (hipStream_t)hip_stream, smoother, false, false);
}

int pfg_launch_device_grid(pfg_ctx *ctx, int model, int kernel, int dtype, int rng, int n_max, int T_max, int B,
                           const pfg_dev_problem *dev_probs, void *hip_stream) {
    if (!ctx) return PFG_ERR_INVALID;
    if (!dev_probs && B > 0) return fail(ctx, PFG_ERR_INVALID, "pfg_launch_device_grid: dev_probs is NULL");
    PFG_HIP(ctx, hipSetDevice(ctx->device));
    return dispatch_grid(ctx, model, kernel, dtype, rng, n_max, T_max, B, dev_probs, (hipStream_t)hip_stream);
}

int pfg_launch_device_grid_smoother(pfg_ctx *ctx, int model, int kernel, int dtype, int rng, int smoother, int n_max, int T_max,
                                    int phase, int B, const pfg_dev_problem *dev_probs, void *hip_stream) {
    if (!ctx) return PFG_ERR_INVALID;
    if (!dev_probs && B > 0) return fail(ctx, PFG_ERR_INVALID, "pfg_launch_device_grid_smoother: dev_probs is NULL");
    if (smoother != PFG_SMOOTHER_NEMETH && smoother != PFG_SMOOTHER_FILTER && smoother != PFG_SMOOTHER_POYIADJIS_N)
        return fail(ctx, PFG_ERR_UNSUPPORTED, "whole-GPU windows are built for NEMETH / FILTER / POYIADJIS_N");
    if (phase < PFG_GRID_PHASE_FINISH) return fail(ctx, PFG_ERR_INVALID, "pfg_launch_device_grid_smoother: phase must be PFG_GRID_PHASE_ALL, a timestep >= 0, PFG_GRID_PHASE_INIT or PFG_GRID_PHASE_FINISH");
    PFG_HIP(ctx, hipSetDevice(ctx->device));
    return dispatch_grid(ctx, model, kernel, dtype, rng, n_max, phase == PFG_GRID_PHASE_ALL ? T_max : 0, B, dev_probs, (hipStream_t)hip_stream, phase, smoother);
}

int pfg_launch_device_grid_phase(pfg_ctx *ctx, int model, int kernel, int dtype, int rng, int n_max, int phase, int B,
                                 const pfg_dev_problem *dev_probs, void *hip_stream) {
    if (!ctx) return PFG_ERR_INVALID;
    if (!dev_probs && B > 0) return fail(ctx, PFG_ERR_INVALID, "pfg_launch_device_grid_phase: dev_probs is NULL");
    if (phase < PFG_GRID_PHASE_FINISH) return fail(ctx, PFG_ERR_INVALID, "pfg_launch_device_grid_phase: phase must be a timestep >= 0, PFG_GRID_PHASE_INIT or PFG_GRID_PHASE_FINISH");
    if (phase == -1) return fail(ctx, PFG_ERR_INVALID, "pfg_launch_device_grid_phase: use pfg_launch_device_grid for the whole window");
    PFG_HIP(ctx, hipSetDevice(ctx->device));
    return dispatch_grid(ctx, model, kernel, dtype, rng, n_max, 0, B, dev_probs, (hipStream_t)hip_stream, phase);
}

int pfg_sghmc_update_device(pfg_ctx *ctx, int model, int B, double *theta, double *momentum, const double *outs,
                            const pfg_prior_hyper *hyper, double epsilon, double alpha, double Tscale,
                            uint64_t seed, uint64_t chain_offset, uint64_t *step_ctr, void *hip_stream);

int pfg_sgld_update_device(pfg_ctx *ctx, int model, int B, double *theta, const double *outs,
                           const pfg_prior_hyper *hyper, double epsilon, double Tscale, uint64_t seed,
                           uint64_t chain_offset, uint64_t *step_ctr, void *hip_stream) {
    return pfg_sghmc_update_device(ctx, model, B, theta, nullptr, outs, hyper, epsilon, 1.0, Tscale, seed,
                                   chain_offset, step_ctr, hip_stream);
}

int pfg_sghmc_update_device(pfg_ctx *ctx, int model, int B, double *theta, double *momentum, const double *outs,
                            const pfg_prior_hyper *hyper, double epsilon, double alpha, double Tscale,
                            uint64_t seed, uint64_t chain_offset, uint64_t *step_ctr, void *hip_stream) {
    if (!ctx) return PFG_ERR_INVALID;
    if (!(alpha > 0.0 && alpha <= 1.0)) return fail(ctx, PFG_ERR_INVALID, "SGHMC friction alpha must be in (0, 1]");
    if (!theta || !outs || !hyper) return fail(ctx, PFG_ERR_INVALID, "pfg_sgld_update_device: NULL argument");
    if (model < 0 || model > 2) return fail(ctx, PFG_ERR_INVALID, "Unrecognized model id");
    if (!(epsilon > 0.0) || !(Tscale > 0.0)) return fail(ctx, PFG_ERR_INVALID, "epsilon and Tscale must be > 0");
    if (B <= 0) return PFG_OK;
    PFG_HIP(ctx, hipSetDevice(ctx->device));
    hipStream_t st = (hipStream_t)hip_stream;
    hipLaunchKernelGGL(sgld_update_kernel, dim3((B + 127) / 128), dim3(128), 0, st, model, B, theta, outs,
                       *hyper, epsilon, Tscale, seed, chain_offset, (const uint64_t *)step_ctr, momentum, alpha);
    if (step_ctr) hipLaunchKernelGGL(bump_counter_kernel, dim3(1), dim3(1), 0, st, step_ctr);
    PFG_HIP(ctx, hipGetLastError());
    return PFG_OK;
}

int pfg_sample_windows_device(pfg_ctx *ctx, int B, pfg_dev_problem *dev_probs, const double *y_dev,
                              const double *weights_table_dev, int T, int S, int buffer, int strict,
                              uint64_t seed, uint64_t chain_offset, const uint64_t *step_ctr, void *hip_stream) {
    if (!ctx) return PFG_ERR_INVALID;
    if (B <= 0) return PFG_OK;
    if (!dev_probs || !y_dev) return fail(ctx, PFG_ERR_INVALID, "pfg_sample_windows_device: NULL argument");
    if (S < 1 || S > T || buffer < 0) return fail(ctx, PFG_ERR_INVALID, "need 1 <= S <= T and buffer >= 0");
    PFG_HIP(ctx, hipSetDevice(ctx->device));
    hipLaunchKernelGGL(sample_windows_kernel, dim3((B + 127) / 128), dim3(128), 0, (hipStream_t)hip_stream, B,
                       dev_probs, y_dev, weights_table_dev, T, S, buffer, strict, seed, chain_offset, step_ctr);
    PFG_HIP(ctx, hipGetLastError());
    return PFG_OK;
}

int pfg_imq_ksd(pfg_ctx *ctx, int K, int d, const double *x, const double *g, double c, double beta,
                double *ksd_out) {
    if (!ctx) return PFG_ERR_INVALID;
    if (!x || !g || !ksd_out) return fail(ctx, PFG_ERR_INVALID, "pfg_imq_ksd: NULL argument");
    if (K < 1 || d < 1 || d > KSD_MAX_D) return fail(ctx, PFG_ERR_INVALID, "pfg_imq_ksd: need K >= 1 and 1 <= d <= 8");
    if (!(beta > 0.0 && beta < 1.0)) return fail(ctx, PFG_ERR_INVALID, "pfg_imq_ksd: beta must be in (0,1)");
    PFG_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)K * d;
    const int nblk = K < 1024 ? K : 1024;
    PFG_HIP(ctx, ctx->in.ensure(2 * n * 8));
    PFG_HIP(ctx, ctx->out.ensure((size_t)nblk * 8));
    double *dx = static_cast<double *>(ctx->in.ptr), *dg = dx + n;
    PFG_HIP(ctx, hipMemcpyAsync(dx, x, n * 8, hipMemcpyHostToDevice, ctx->stream));
    PFG_HIP(ctx, hipMemcpyAsync(dg, g, n * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(imq_ksd_kernel, dim3(nblk), dim3(256), 0, ctx->stream, K, d, dx, dg, c * c, beta,
                       static_cast<double *>(ctx->out.ptr));
    PFG_HIP(ctx, hipGetLastError());
    PFG_HIP(ctx, ctx->h_out.ensure((size_t)nblk));
    PFG_HIP(ctx, hipMemcpyAsync(ctx->h_out.data(), ctx->out.ptr, (size_t)nblk * 8, hipMemcpyDeviceToHost, ctx->stream));
    PFG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double tot = 0.0;
    for (int b = 0; b < nblk; ++b) tot += ctx->h_out[b];          // fixed order: reproducible
    *ksd_out = std::sqrt(tot) / (double)K;
    return PFG_OK;
}

int pfg_run(pfg_ctx *ctx, const pfg_problem *p, pfg_result *r) { return pfg_run_batch(ctx, 1, p, r); }

int pfg_run_batch(pfg_ctx *ctx, int B, const pfg_problem *ps, pfg_result *rs) {
    if (!ctx) return PFG_ERR_INVALID;
    if (B < 0 || (B > 0 && (!ps || !rs))) return fail(ctx, PFG_ERR_INVALID, "pfg_run_batch: NULL problems/results");
    if (B == 0) return PFG_OK;
    const int model = ps[0].model, kernel = ps[0].kernel, dtype = ps[0].dtype, rng = ps[0].rng;
    int rc = check_combo(ctx, model, kernel, dtype, rng);
    if (rc) return rc;
    const int NS = state_dim(model), H = stat_dim(model), P = theta_dim(model);

    // ---- validate + size ------------------------------------------------------------
    // n_in: doubles of the device input arena; n_host: doubles of the pinned staging arena -- inputs shared by many
    // windows of the batch (same pointer and length) count once in both, inputs that lie in caller-registered
    // pinned memory need device space but no staging space (same decisions as `put` / `put_shared` below)
    size_t n_in = 0, n_host = 0, n_out = 0, n_work = 0;
    int n_max = 0;
    std::unordered_map<const double *, size_t> sized_shared;
    auto size_in = [&](const double *src, size_t n) {
        if (!src || n == 0) return;
        n_in += n;
        if (!(n >= kDirectMinDoubles && host_registered(src, n * 8))) n_host += n;
    };
    auto size_shared = [&](const double *src, size_t n) {
        if (!src || n == 0) return;
        auto it = sized_shared.find(src);
        if (it != sized_shared.end() && it->second == n) return;
        sized_shared[src] = n;
        size_in(src, n);
    };
    for (int b = 0; b < B; ++b) {
        const pfg_problem &q = ps[b];
        std::string id = "problem " + std::to_string(b) + ": ";
        if (q.model != model || q.kernel != kernel || q.dtype != dtype || q.rng != rng)
            return fail(ctx, PFG_ERR_INVALID, id + "model/kernel/dtype/rng must match across a batch");
        if (q.N < 1) return fail(ctx, PFG_ERR_INVALID, id + "N must be >= 1");
        if (q.T < 0) return fail(ctx, PFG_ERR_INVALID, id + "T must be >= 0");
        if (q.t1 < 0 || q.tL < q.t1) return fail(ctx, PFG_ERR_INVALID, id + "need 0 <= t1 <= tL");
        if (q.smoother < PFG_SMOOTHER_NEMETH || q.smoother > PFG_SMOOTHER_POYIADJIS_N2)
            return fail(ctx, PFG_ERR_INVALID, id + "Unrecognized pf (smoother id)");
        if ((q.smoother == PFG_SMOOTHER_POYIADJIS_N2) != (ps[0].smoother == PFG_SMOOTHER_POYIADJIS_N2))
            return fail(ctx, PFG_ERR_INVALID, id + "pf = 'poyiadjis_N2' cannot share a batch with other smoothers");
        if (q.smoother == PFG_SMOOTHER_POYIADJIS_N2 && q.N > pfg::MEM_MAX_N)
            return fail(ctx, PFG_ERR_UNSUPPORTED, id + "pf = 'poyiadjis_N2' is implemented for N <= 16384");
        if (q.smoother == PFG_SMOOTHER_POYIADJIS_N2 && q.elementwise && q.N > 4096)
            return fail(ctx, PFG_ERR_UNSUPPORTED, id + "elementwise statistics with pf = 'poyiadjis_N2' are implemented for N <= 4096");
        if (q.smoother == PFG_SMOOTHER_POYIADJIS_N2 && q.stat == PFG_STAT_PREDICTIVE)
            return fail(ctx, PFG_ERR_INVALID, id + "Only can use pf = 'filter' since we are filtering");
        if ((q.smoother == PFG_SMOOTHER_PARIS) != (ps[0].smoother == PFG_SMOOTHER_PARIS))
            return fail(ctx, PFG_ERR_INVALID, id + "pf = 'paris' cannot share a batch with other smoothers");
        if (q.smoother == PFG_SMOOTHER_PARIS) {
            if (q.Ntilde < 1 || q.Ntilde > 64) return fail(ctx, PFG_ERR_INVALID, id + "Ntilde must be in [1, 64]");
            if (q.max_accept_reject < 0) return fail(ctx, PFG_ERR_INVALID, id + "max_accept_reject must be >= 0");
            if (q.paris_stream) {
                if (rng != PFG_RNG_REPLAY) return fail(ctx, PFG_ERR_INVALID, id + "paris_stream is a REPLAY input");
                if (q.paris_idx_u || q.paris_acc_u || q.paris_man_u)
                    return fail(ctx, PFG_ERR_INVALID, id + "paris_stream replaces the addressed pools paris_idx_u / acc_u / man_u");
                if (q.paris_stream_len < 0 || q.paris_manual_threshold < 0)
                    return fail(ctx, PFG_ERR_INVALID, id + "paris_stream_len and paris_manual_threshold must be >= 0");
                if (q.N > pfg::MEM_MAX_N)
                    return fail(ctx, PFG_ERR_UNSUPPORTED, id + "pf = 'paris' is implemented for N <= 16384");
                if ((q.flags & PFG_FLAG_PARIS_RAW_STREAM) && dtype != PFG_F64)
                    return fail(ctx, PFG_ERR_UNSUPPORTED, id + "PFG_FLAG_PARIS_RAW_STREAM is built for dtype f64 (it reproduces np.random's doubles)");
                if ((q.flags & PFG_FLAG_PARIS_RAW_STREAM) && (q.z0 || q.u || q.z))
                    return fail(ctx, PFG_ERR_INVALID, id + "PFG_FLAG_PARIS_RAW_STREAM draws z0 / u / z from paris_stream: they must be NULL");
                if ((q.flags & PFG_FLAG_PARIS_RAW_CARRY) && (!(q.flags & PFG_FLAG_PARIS_RAW_STREAM) || q.paris_stream_len < 1))
                    return fail(ctx, PFG_ERR_INVALID, id + "PFG_FLAG_PARIS_RAW_CARRY needs PFG_FLAG_PARIS_RAW_STREAM and the cached Gaussian in paris_stream[0]");
                if ((q.flags & PFG_FLAG_PARIS_RAW_CARRY) && q.init_x && q.T == 0)
                    return fail(ctx, PFG_ERR_INVALID, id + "PFG_FLAG_PARIS_RAW_CARRY with a warm start and T = 0 draws no normal: nothing to carry the cached Gaussian through");
            } else if (rng == PFG_RNG_REPLAY && (!q.paris_man_u || (q.max_accept_reject > 0 && (!q.paris_idx_u || !q.paris_acc_u)))) {
                return fail(ctx, PFG_ERR_INVALID, id + "REPLAY paris needs the paris_* uniform pools or paris_stream");
            }
            if ((q.flags & (PFG_FLAG_PARIS_RAW_STREAM | PFG_FLAG_PARIS_RAW_CARRY)) && !q.paris_stream)
                return fail(ctx, PFG_ERR_INVALID, id + "PFG_FLAG_PARIS_RAW_STREAM needs paris_stream");
            if ((q.flags & PFG_FLAG_PARIS_NO_ACCEPT_REJECT) && !q.paris_stream)
                return fail(ctx, PFG_ERR_UNSUPPORTED, id + "PaRIS with accept_reject = False is built for the REPLAY stream order (paris_stream)");
        } else if (q.paris_stream) {
            return fail(ctx, PFG_ERR_INVALID, id + "paris_stream needs pf = 'paris'");
        }
        if (q.stat < PFG_STAT_SCORE || q.stat > PFG_STAT_PREDICTIVE) return fail(ctx, PFG_ERR_INVALID, id + "bad stat id");
        if ((q.stat == PFG_STAT_PREDICTIVE) != (ps[0].stat == PFG_STAT_PREDICTIVE))
            return fail(ctx, PFG_ERR_INVALID, id + "the predictive statistic cannot share a batch with others");
        if (q.stat == PFG_STAT_PREDICTIVE) {
            if (q.smoother != PFG_SMOOTHER_FILTER)                                  // svm/helper.py:209-210
                return fail(ctx, PFG_ERR_INVALID, id + "Only can use pf = 'filter' since we are filtering");
            if (q.num_steps_ahead < 0 || q.num_steps_ahead >= PFG_MAX_PRED)
                return fail(ctx, PFG_ERR_INVALID, id + "num_steps_ahead must be in [0, 15]");
            if (q.N > pfg::MEM_MAX_N)
                return fail(ctx, PFG_ERR_UNSUPPORTED, id + "N exceeds the supported maximum of 16384");
            if (rng == PFG_RNG_REPLAY && model != PFG_MODEL_LGSSM && q.T > 0 && !q.pred_z)
                return fail(ctx, PFG_ERR_INVALID, id + "REPLAY predictive needs the pred_z pool");
        }
        if (!q.theta) return fail(ctx, PFG_ERR_INVALID, id + "theta is NULL");
        if (q.T > 0 && !q.y) return fail(ctx, PFG_ERR_INVALID, id + "observations are NULL");
        const bool raw_stream = q.smoother == PFG_SMOOTHER_PARIS && (q.flags & PFG_FLAG_PARIS_RAW_STREAM) != 0;
        if (rng == PFG_RNG_REPLAY && !raw_stream && !q.init_x && !q.z0) return fail(ctx, PFG_ERR_INVALID, id + "REPLAY needs z0");
        if (rng == PFG_RNG_REPLAY && !raw_stream && q.T > 0 && (!q.u || !q.z)) return fail(ctx, PFG_ERR_INVALID, id + "REPLAY needs u and z");
        if (q.init_x && !q.init_logw) return fail(ctx, PFG_ERR_INVALID, id + "init_x needs init_logw");
        if ((q.smoother == PFG_SMOOTHER_NEMETH_SYSTEMATIC) != (ps[0].smoother == PFG_SMOOTHER_NEMETH_SYSTEMATIC))
            return fail(ctx, PFG_ERR_INVALID, id + "systematic resampling cannot share a batch with other smoothers");
        if (q.smoother == PFG_SMOOTHER_NEMETH_SYSTEMATIC && (rng != PFG_RNG_DEVICE || q.N > 1024))
            return fail(ctx, PFG_ERR_UNSUPPORTED, id + "systematic resampling needs the DEVICE rng and N <= 1024");
        if (!(q.prior_var >= 0.0) && !(q.flags & PFG_FLAG_GARCH_STATIONARY_PRIOR) && !q.init_x)
            return fail(ctx, PFG_ERR_INVALID, id + "prior_var must be >= 0");
        if (model == PFG_MODEL_SVM && std::fabs(q.theta[0]) > 1.0) {
            char buf[160];
            snprintf(buf, sizeof buf, "Current AR parameter is |A| = %.17g > 1\nTry calling project_parameters?",
                     std::fabs(q.theta[0]));
            return fail(ctx, PFG_ERR_NUMERIC, buf);                                   // svm/kernels.py:6-11
        }
        n_max = q.N > n_max ? q.N : n_max;
        const int nw = q.weights ? (q.tL < q.T ? q.tL : q.T) - q.t1 : 0;
        size_shared(q.y, (size_t)q.T);
        size_shared(q.weights, nw > 0 ? (size_t)nw : 0);
        n_in += PFG_MAX_THETA + (q.step ? 1 : 0);
        n_host += PFG_MAX_THETA + (q.step ? 1 : 0);
        if (rng == PFG_RNG_REPLAY) {
            size_in(q.z0, (size_t)q.N);
            size_in(q.u, (size_t)q.T * q.N);
            size_in(q.z, (size_t)q.T * q.N);
        }
        if (q.init_x) {
            size_in(q.init_x, (size_t)q.N * NS);
            size_in(q.init_logw, (size_t)q.N);
            size_in(q.init_stats, (size_t)q.N * H);
        }
        if (q.smoother == PFG_SMOOTHER_PARIS && rng == PFG_RNG_REPLAY) {
            const size_t pool = (size_t)q.T * q.Ntilde * q.max_accept_reject * q.N;
            size_in(q.paris_idx_u, pool);
            size_in(q.paris_acc_u, pool);
            size_in(q.paris_man_u, (size_t)q.T * q.Ntilde * q.N);
            size_in(q.paris_stream, (size_t)q.paris_stream_len);
            if (q.paris_stream) n_out += 2;                                   // the consumed count, the carry-back distance
        }
        if (q.stat == PFG_STAT_PREDICTIVE && rng == PFG_RNG_REPLAY && model != PFG_MODEL_LGSSM)
            size_in(q.pred_z, (size_t)q.T * (q.num_steps_ahead + 1) * q.N);
        const pfg_result &r = rs[b];
        n_out += PFG_OUT_DOUBLES + (q.stat == PFG_STAT_PREDICTIVE ? PFG_MAX_PRED : 0);
        if (r.x_T) n_out += (size_t)q.N * NS;
        if (r.logw_T) n_out += q.N;
        if (r.stats_T) n_out += (size_t)q.N * H;
        if (r.trace_x) n_out += (size_t)(q.T + 1) * q.N * NS;
        if (r.trace_logw) n_out += (size_t)(q.T + 1) * q.N;
        if (r.trace_stats) n_out += (size_t)(q.T + 1) * q.N * H;
        if (r.trace_ll) n_out += (size_t)q.T + 1;
        if (r.trace_anc) n_out += ((size_t)q.T * q.N + 1) / 2;       /* int32 pairs in f64 slots */
        if (q.elementwise) {
            if (q.smoother != PFG_SMOOTHER_NEMETH && q.smoother != PFG_SMOOTHER_PARIS && q.smoother != PFG_SMOOTHER_POYIADJIS_N2)
                return fail(ctx, PFG_ERR_UNSUPPORTED, id + "elementwise statistics are built for pf = 'poyiadjis_N' | 'nemeth' | 'paris' | 'poyiadjis_N2'");
            if (q.stat == PFG_STAT_PREDICTIVE) return fail(ctx, PFG_ERR_INVALID, id + "elementwise statistics do not combine with the predictive statistic");
            if (!r.ew_mean) return fail(ctx, PFG_ERR_INVALID, id + "elementwise needs ew_mean");
            if (r.trace_x || r.trace_logw || r.trace_stats || r.trace_anc || r.rec_u || r.rec_z || r.rec_z0 || r.rec_ud)
                return fail(ctx, PFG_ERR_INVALID, id + "elementwise statistics cannot be combined with trace outputs");
            const int tLc = q.tL < q.T ? q.tL : q.T;
            if (tLc - q.t1 < 1) return fail(ctx, PFG_ERR_INVALID, id + "elementwise needs a non-empty window [t1, tL)");
            const size_t Wd = 3 * (size_t)(tLc - q.t1), Nt = q.smoother == PFG_SMOOTHER_PARIS ? (size_t)q.Ntilde : 1;
            n_out += Wd + (r.ew_stats ? (size_t)q.N * Wd : 0);
            n_work += (size_t)(q.T + 1) * q.N * (NS + 1) + ((size_t)q.T * q.N * Nt + 1) / 2 + 2 * (size_t)q.N * Wd + Wd + q.N + 8;
        } else if (r.ew_mean || r.ew_stats) {
            return fail(ctx, PFG_ERR_INVALID, id + "ew_mean / ew_stats need pfg_problem.elementwise");
        }
        if (r.rec_u || r.rec_z || r.rec_z0 || r.rec_ud) {
            if (rng != PFG_RNG_DEVICE || !r.trace_x)
                return fail(ctx, PFG_ERR_INVALID, id + "rec_u / rec_z / rec_z0 record the DEVICE generator's draws and need trace_x");
            if (r.rec_u) n_out += ((size_t)q.T * q.N + 1) / 2;
            if (r.rec_z) n_out += (size_t)q.T * q.N;
            if (r.rec_z0) n_out += q.N;
            if (r.rec_ud) n_out += (size_t)q.T * q.N;
        }
        if (r.trace_anc && !r.trace_x) return fail(ctx, PFG_ERR_INVALID, id + "trace_anc needs trace_x");
        if ((r.logw_T || r.stats_T) && !r.x_T) return fail(ctx, PFG_ERR_INVALID, id + "logw_T/stats_T need x_T");
        if ((r.trace_logw == nullptr) != (r.trace_x == nullptr))
            return fail(ctx, PFG_ERR_INVALID, id + "trace_x and trace_logw go together");
        if (r.trace_stats && !r.trace_x) return fail(ctx, PFG_ERR_INVALID, id + "trace_stats needs trace_x");
    }
    bool traced = false;
    for (int b = 0; b < B; ++b)
        traced = traced || rs[b].trace_x || rs[b].trace_ll || rs[b].rec_u || rs[b].rec_z || rs[b].rec_z0 || rs[b].rec_ud || ps[b].elementwise;
    bool score1 = !traced;          // every window the Poyiadjis O(N) score: units with a twin specialised to it run that
    for (int b = 0; b < B; ++b)
        score1 = score1 && ps[b].smoother == PFG_SMOOTHER_NEMETH && ps[b].lambduh == 1.0 && ps[b].stat == PFG_STAT_SCORE;
    const bool paris = ps[0].smoother == PFG_SMOOTHER_PARIS;
    const bool sysres = ps[0].smoother == PFG_SMOOTHER_NEMETH_SYSTEMATIC;
    const bool predictive = ps[0].stat == PFG_STAT_PREDICTIVE;   // large-N kernel only (any N)
    const bool n2 = ps[0].smoother == PFG_SMOOTHER_POYIADJIS_N2;
    int variant = paris ? kVariantParis : sysres ? kVariantSystematic : n2 ? kVariantN2
                  : predictive ? kVariantMem : pick_variant(model, dtype, rng, n_max);
    if (variant == -1)
        return fail(ctx, PFG_ERR_UNSUPPORTED,
                    "N = " + std::to_string(n_max) + " exceeds the supported maximum of " + std::to_string(pfg::GRID_MAX_N));
    // PFGRAD_VARIANT=grid: the whole-GPU window also where a one-workgroup kernel would serve (tests, A/B timing)
    if (variant != kVariantGrid && !paris && !sysres && !n2 && !predictive) {
        const char *force = std::getenv("PFGRAD_VARIANT");
        if (force && !std::strcmp(force, "grid")) variant = kVariantGrid;
    }
    int t_max = 0;
    if (variant == kVariantGrid) {
        for (int b = 0; b < B; ++b) {
            const pfg_problem &q = ps[b];
            const std::string id = "problem " + std::to_string(b) + ": ";
            if (q.smoother != PFG_SMOOTHER_NEMETH && q.smoother != PFG_SMOOTHER_FILTER)
                return fail(ctx, PFG_ERR_UNSUPPORTED, id + "N > " + std::to_string(pfg::MEM_MAX_N) + " is built for pf = 'poyiadjis_N' | 'nemeth' | 'filter'");
            if (q.elementwise) return fail(ctx, PFG_ERR_UNSUPPORTED, id + "elementwise statistics are built for N <= " + std::to_string(pfg::MEM_MAX_N));
            if (pfg::grid_ppt(q.N) != pfg::grid_ppt(n_max))
                return fail(ctx, PFG_ERR_INVALID, id + "whole-GPU windows of one batch must all have N <= 524288 or all N > 524288");
            t_max = q.T > t_max ? q.T : t_max;
        }
    }
    size_t n_scratch = 0;                  // bytes; every window of the batch gets n_max-sized state
    const size_t pred_each = predictive ? ((size_t)n_max * PFG_MAX_PRED * (dtype == PFG_F64 ? 8 : 4) + 255) / 256 * 256 : 0;
    const bool paris_mem = (paris || n2) && n_max > 1024;       // the large-N kernel's PaRIS instantiation (also its O(N^2) sweep)
    const size_t scratch_each = (scratch_bytes(model, dtype, n_max, paris_mem) + 255) / 256 * 256 + pred_each;
    if (variant == kVariantMem || paris_mem) n_scratch = scratch_each * (size_t)B;
    const size_t grid_each = variant == kVariantGrid ? grid_scratch_bytes(model, dtype, rng, n_max) : 0;
    if (variant == kVariantGrid) n_scratch = grid_each * (size_t)B;

    PFG_HIP(ctx, hipSetDevice(ctx->device));
    PFG_HIP(ctx, ctx->in.ensure(n_in * 8));
    PFG_HIP(ctx, ctx->out.ensure(n_out * 8));
    PFG_HIP(ctx, ctx->desc.ensure((size_t)B * sizeof(pfg_dev_problem)));
    if (n_scratch) PFG_HIP(ctx, ctx->scratch.ensure(n_scratch));
    if (n_work) PFG_HIP(ctx, ctx->work.ensure(n_work * 8));
    double *hin = nullptr;
    try {
        if (ctx->h_in.ensure(n_host) == hipSuccess) {
            hin = ctx->h_in.data();
        } else {
            // the runtime refuses to page-lock that much: stage from pageable memory (slower copies, same result)
            (void)hipGetLastError();
            ctx->h_in_pageable.resize(n_host);
            hin = ctx->h_in_pageable.data();
        }
        if (ctx->h_out.ensure(n_out) != hipSuccess) throw std::bad_alloc();
        ctx->h_desc.assign(B, pfg_dev_problem{});
    } catch (const std::bad_alloc &) {
        return fail(ctx, PFG_ERR_NOMEM, "pfg_run_batch: out of host memory");
    }

    // ---- pack ---------------------------------------------------------------------------
    const double *din = static_cast<const double *>(ctx->in.ptr);
    double *dout = static_cast<double *>(ctx->out.ptr);
    size_t oi = 0, oh = 0, oo = 0;       // offsets into the device input arena, the host staging arena, the output arena
    // big inputs that lie in caller-registered pinned memory (pfg_host_register) go to the device straight
    // from there; everything else is packed into the library's staging arena.  `copies` = the H2D transfers:
    // runs of packed pieces (contiguous on both sides) and the direct pieces.
    struct Copy { size_t at; const double *src; size_t n; bool packed; };
    std::vector<Copy> copies;
    bool overflow = false;               // a piece that would not fit what the sizing pass reserved (it never copies)
    auto put = [&](const double *src, size_t n) -> const double * {
        if (!src || n == 0) return nullptr;
        const bool direct = n >= kDirectMinDoubles && host_registered(src, n * 8);
        if (oi + n > n_in || (!direct && oh + n > n_host)) { overflow = true; return nullptr; }
        if (direct) {
            copies.push_back({oi, src, n, false});
        } else {
            std::memcpy(hin + oh, src, n * 8);
            if (!copies.empty() && copies.back().packed && copies.back().src + copies.back().n == hin + oh && copies.back().at + copies.back().n == oi)
                copies.back().n += n;                     // extends the current packed run
            else
                copies.push_back({oi, hin + oh, n, true});
            oh += n;
        }
        const double *d = din + oi;
        oi += n;
        return d;
    };
    // observations / window weights that many windows of the batch share (same host pointer and length) are
    // staged once: 12288 chains on one series would otherwise carry 98 MB of copies of the same 8 KB
    std::unordered_map<const double *, std::pair<size_t, const double *>> shared;
    auto put_shared = [&](const double *src, size_t n) -> const double * {
        if (!src || n == 0) return nullptr;
        auto it = shared.find(src);
        if (it != shared.end() && it->second.first == n) return it->second.second;
        const double *d = put(src, n);
        shared[src] = std::make_pair(n, d);
        return d;
    };
    auto take = [&](bool want, size_t n) -> double * {
        if (!want) return nullptr;
        double *d = dout + oo;
        oo += n;
        return d;
    };
    // elementwise pass: device-only buffers (traces the filter records, the statistic matrices)
    struct EwPlan { double *tx, *tlw, *S0, *S1, *Sbar, *w, *mean, *stats; int32_t *par; size_t Wd; int Nt; const double *theta; };
    std::vector<EwPlan> ew(B, EwPlan{});
    double *dwork = static_cast<double *>(ctx->work.ptr);
    size_t ow = 0;
    auto work = [&](size_t n) -> double * { double *d = dwork + ow; ow += n; return d; };
    for (int b = 0; b < B; ++b) {
        const pfg_problem &q = ps[b];
        const pfg_result &r = rs[b];
        pfg_dev_problem &d = ctx->h_desc[b];
        const int tL = q.tL < q.T ? q.tL : q.T;
        const int nw = q.weights ? tL - q.t1 : 0;
        d.y = put_shared(q.y, q.T);
        d.weights = put_shared(q.weights, nw > 0 ? nw : 0);
        {
            double th[PFG_MAX_THETA] = {0, 0, 0, 0};
            for (int j = 0; j < P; ++j) th[j] = q.theta[j];
            d.theta = put(th, PFG_MAX_THETA);
        }
        if (rng == PFG_RNG_REPLAY) {
            d.z0 = put(q.z0, q.z0 ? q.N : 0);
            d.u = put(q.u, (size_t)q.T * q.N);
            d.z = put(q.z, (size_t)q.T * q.N);
        }
        if (q.init_x) {
            d.init_x = put(q.init_x, (size_t)q.N * NS);
            d.init_logw = put(q.init_logw, q.N);
            d.init_stats = put(q.init_stats, q.init_stats ? (size_t)q.N * H : 0);
        }
        if (paris) {
            d.Ntilde = q.Ntilde; d.max_accept_reject = q.max_accept_reject;
            if (rng == PFG_RNG_REPLAY) {
                const size_t pool = (size_t)q.T * q.Ntilde * q.max_accept_reject * q.N;
                d.paris_idx_u = put(q.paris_idx_u, pool);
                d.paris_acc_u = put(q.paris_acc_u, pool);
                d.paris_man_u = put(q.paris_man_u, (size_t)q.T * q.Ntilde * q.N);
                if (q.paris_stream) {
                    d.paris_stream = put(q.paris_stream, (size_t)q.paris_stream_len);
                    if (!d.paris_stream) d.paris_stream = din;        // an empty stream is still "stream order" (non-NULL)
                    d.paris_stream_len = q.paris_stream_len;
                    d.paris_manual_threshold = q.paris_manual_threshold;
                    d.paris_consumed = reinterpret_cast<int64_t *>(take(true, 2));
                }
            }
        }
        if (predictive) {
            d.num_steps_ahead = q.num_steps_ahead;
            if (rng == PFG_RNG_REPLAY && model != PFG_MODEL_LGSSM)
                d.pred_z = put(q.pred_z, (size_t)q.T * (q.num_steps_ahead + 1) * q.N);
            d.pred_out = take(true, PFG_MAX_PRED);
        }
        d.out = take(true, PFG_OUT_DOUBLES);
        d.final_x = take(r.x_T != nullptr, (size_t)q.N * NS);
        d.final_logw = take(r.logw_T != nullptr, q.N);
        d.final_stats = take(r.stats_T != nullptr, (size_t)q.N * H);
        d.trace_x = take(r.trace_x != nullptr, (size_t)(q.T + 1) * q.N * NS);
        d.trace_logw = take(r.trace_logw != nullptr, (size_t)(q.T + 1) * q.N);
        d.trace_stats = take(r.trace_stats != nullptr, (size_t)(q.T + 1) * q.N * H);
        d.trace_ll = take(r.trace_ll != nullptr, (size_t)q.T + 1);
        d.trace_anc = reinterpret_cast<int32_t *>(take(r.trace_anc != nullptr, ((size_t)q.T * q.N + 1) / 2));
        if (q.elementwise) {
            EwPlan &e = ew[b];
            e.Wd = 3 * (size_t)(tL - q.t1);
            e.Nt = q.smoother == PFG_SMOOTHER_PARIS ? q.Ntilde : 1;
            e.tx = work((size_t)(q.T + 1) * q.N * NS);
            e.tlw = work((size_t)(q.T + 1) * q.N);
            e.par = reinterpret_cast<int32_t *>(work(((size_t)q.T * q.N * e.Nt + 1) / 2));
            e.S0 = work((size_t)q.N * e.Wd); e.S1 = work((size_t)q.N * e.Wd);
            e.Sbar = work(e.Wd); e.w = work(q.N);
            e.mean = take(true, e.Wd);
            e.stats = take(r.ew_stats != nullptr, (size_t)q.N * e.Wd);
            d.trace_x = e.tx; d.trace_logw = e.tlw;
            e.theta = d.theta;
            if (q.smoother == PFG_SMOOTHER_PARIS) d.trace_paris_J = e.par; else d.trace_anc = e.par;
        }
        d.rec_u = reinterpret_cast<uint32_t *>(take(r.rec_u != nullptr, ((size_t)q.T * q.N + 1) / 2));
        d.rec_z = take(r.rec_z != nullptr, (size_t)q.T * q.N);
        d.rec_z0 = take(r.rec_z0 != nullptr, q.N);
        d.rec_ud = take(r.rec_ud != nullptr, (size_t)q.T * q.N);
        d.step_ctr = nullptr;
        if (q.step) {       // the step counter of this window, as a resident chain would read it from HBM
            double slot;
            std::memcpy(&slot, &q.step, sizeof slot);
            d.step_ctr = reinterpret_cast<const uint64_t *>(put(&slot, 1));
        }
        d.scratch = n_scratch ? static_cast<void *>(static_cast<char *>(ctx->scratch.ptr) + (variant == kVariantGrid ? grid_each : scratch_each) * (size_t)b)
                              : nullptr;
        if (predictive) d.pred_scratch = static_cast<char *>(d.scratch) + (scratch_each - pred_each);
        d.prior_mean = q.prior_mean; d.prior_var = q.prior_var; d.lambduh = q.lambduh;
        d.seed = q.seed; d.stream = q.stream;
        d.T = q.T; d.t1 = q.t1; d.tL = tL; d.N = q.N;
        d.smoother = q.smoother; d.stat = q.stat; d.flags = q.flags;
    }

    // ---- stage, launch, fetch -----------------------------------------------------------
    if (overflow || oi > n_in || oh > n_host)
        return fail(ctx, PFG_ERR_INVALID, "pfg_run_batch: internal sizing error (an input changed its registration state during the call?)");
    for (const Copy &cp : copies)
        PFG_HIP(ctx, hipMemcpyAsync(static_cast<double *>(ctx->in.ptr) + cp.at, cp.src, cp.n * 8, hipMemcpyHostToDevice, ctx->stream));
    PFG_HIP(ctx, hipMemcpyAsync(ctx->desc.ptr, ctx->h_desc.data(), (size_t)B * sizeof(pfg_dev_problem),
                                hipMemcpyHostToDevice, ctx->stream));
    PFG_HIP(ctx, hipMemsetAsync(ctx->out.ptr, 0, oo * 8, ctx->stream));
    if (variant == kVariantGrid)
        rc = dispatch_grid(ctx, model, kernel, dtype, rng, n_max, t_max, B, static_cast<const pfg_dev_problem *>(ctx->desc.ptr), ctx->stream,
                           -1, score1 ? PFG_SMOOTHER_POYIADJIS_N : PFG_SMOOTHER_NEMETH);
    else
        rc = dispatch(ctx, model, kernel, dtype, rng, n_max, B, static_cast<const pfg_dev_problem *>(ctx->desc.ptr),
                      ctx->stream, paris ? PFG_SMOOTHER_PARIS : sysres ? PFG_SMOOTHER_NEMETH_SYSTEMATIC
                                   : n2 ? PFG_SMOOTHER_POYIADJIS_N2 : score1 ? PFG_SMOOTHER_POYIADJIS_N : PFG_SMOOTHER_NEMETH,
                      predictive, traced);
    if (rc) return rc;
    for (int b = 0; b < B; ++b) {
        if (!ps[b].elementwise) continue;
        rc = elementwise_pass(ctx, ps[b], ew[b].theta, ew[b].tx, ew[b].tlw, ew[b].par, ew[b].Nt, ew[b].Wd, ew[b].S0, ew[b].S1,
                              ew[b].Sbar, ew[b].w, ew[b].mean, ew[b].stats);
        if (rc) return rc;
    }
    PFG_HIP(ctx, hipMemcpyAsync(ctx->h_out.data(), ctx->out.ptr, oo * 8, hipMemcpyDeviceToHost, ctx->stream));
    PFG_HIP(ctx, hipStreamSynchronize(ctx->stream));

    const double *hout = ctx->h_out.data();
    for (int b = 0; b < B; ++b) {
        const pfg_problem &q = ps[b];
        pfg_result &r = rs[b];
        const pfg_dev_problem &d = ctx->h_desc[b];
        auto host_of = [&](const double *dev) { return hout + (dev - dout); };
        const double *o = host_of(d.out);
        for (int h = 0; h < PFG_MAX_STAT; ++h) r.mean_stat[h] = o[h];
        r.loglik = o[4];
        if (predictive) {
            const double *pp = host_of(d.pred_out);
            for (int k = 0; k < PFG_MAX_PRED; ++k) r.pred[k] = pp[k];
        }
        auto fetch = [&](double *dst, const double *dev, size_t n) {
            if (dst && dev) std::memcpy(dst, host_of(dev), n * 8);
        };
        fetch(r.x_T, d.final_x, (size_t)q.N * NS);
        fetch(r.logw_T, d.final_logw, q.N);
        fetch(r.stats_T, d.final_stats, (size_t)q.N * H);
        fetch(r.trace_x, d.trace_x, (size_t)(q.T + 1) * q.N * NS);
        fetch(r.trace_logw, d.trace_logw, (size_t)(q.T + 1) * q.N);
        fetch(r.trace_stats, d.trace_stats, (size_t)(q.T + 1) * q.N * H);
        fetch(r.trace_ll, d.trace_ll, (size_t)q.T + 1);
        if (r.trace_anc && d.trace_anc)
            std::memcpy(r.trace_anc, host_of(reinterpret_cast<const double *>(d.trace_anc)), (size_t)q.T * q.N * 4);
        if (r.rec_u && d.rec_u)
            std::memcpy(r.rec_u, host_of(reinterpret_cast<const double *>(d.rec_u)), (size_t)q.T * q.N * 4);
        fetch(r.rec_z, d.rec_z, (size_t)q.T * q.N);
        fetch(r.rec_z0, d.rec_z0, q.N);
        fetch(r.rec_ud, d.rec_ud, (size_t)q.T * q.N);
        r.paris_consumed = 0;
        r.paris_carry_back = 0;
        if (d.paris_consumed) {
            int64_t two[2];
            std::memcpy(two, host_of(reinterpret_cast<const double *>(d.paris_consumed)), 16);
            r.paris_consumed = two[0];
            r.paris_carry_back = (int32_t)two[1];
        }
        if (q.elementwise) {
            fetch(r.ew_mean, ew[b].mean, ew[b].Wd);
            fetch(r.ew_stats, ew[b].stats, (size_t)q.N * ew[b].Wd);
        }
        r.status = PFG_OK;
    }
    return PFG_OK;
}

}  // extern "C"
