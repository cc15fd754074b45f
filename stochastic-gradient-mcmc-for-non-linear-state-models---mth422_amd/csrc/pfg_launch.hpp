// Kernel launch templates of libpfgrad.so.  Included ONLY by the instantiation units
// (pfg_inst_*.hip), each of which instantiates launch_mk for one (model, proposal kernel).
#pragma once
#include "pfg_host.hpp"
#include "pfg_device.hpp"

namespace pfg_host {

// Dynamic LDS above 64 KB needs hipFuncAttributeMaxDynamicSharedMemorySize; set it once per kernel and
// CONTEXT (again only if a larger size is asked for), not on every launch.  The attribute belongs to the
// (function, device) pair and a context is bound to one device; a context is used by one thread at a time.
#define PFG_ENSURE_LDS(ctx, kern, lds)                                                                    \
    do {                                                                                                  \
        if ((lds) > 64 * 1024) {                                                                          \
            size_t &pfg_lds_set_ = (ctx)->lds_set[reinterpret_cast<const void *>(kern)];                  \
            if ((lds) > pfg_lds_set_) {                                                                   \
                PFG_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern),                    \
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds))); \
                pfg_lds_set_ = (lds);                                                                     \
            }                                                                                             \
        }                                                                                                 \
    } while (0)

// traced = the descriptors may carry trace_* / rec_* buffers: the TRACE = true instantiation; otherwise the twin with
// the trace instrumentation compiled out (device generator: what bench.py times; REPLAY: the drop-in Sampler's launch).
template <int MODEL, int KERNEL, typename REAL, int NT, int PPT, int RNG, bool PP, bool TRACE, bool SCORE1 = false>
int launch_one_t(pfg_ctx *ctx, int n_max, int B, const pfg_dev_problem *dp, hipStream_t st) {
    auto kern = pfg::pf_reg_kernel<MODEL, KERNEL, REAL, NT, PPT, RNG, PP, pfg::MODE_PLAIN, TRACE, SCORE1>;
    size_t lds = pfg::reg_kernel_lds_bytes<MODEL, REAL, NT, PPT, RNG, PP>(n_max);
    PFG_ENSURE_LDS(ctx, kern, lds);
    hipLaunchKernelGGL(kern, dim3(B), dim3(NT), lds, st, dp);
    PFG_HIP(ctx, hipGetLastError());
    return PFG_OK;
}
// ctx->score1: the caller launched with PFG_SMOOTHER_POYIADJIS_N (every descriptor: NEMETH, lambduh = 1, score) -- the
// 1024 x 4 and the one-wave x 2 fp64 device units have a twin specialised to that (see SCORE1 in pfg_reg_kernel.hpp); every
// other unit runs its general kernel
template <int MODEL, int KERNEL, typename REAL, int NT, int PPT, int RNG, bool PP>
int launch_one(pfg_ctx *ctx, int n_max, int B, const pfg_dev_problem *dp, hipStream_t st, bool traced) {
    // (measured per unit, whole library built with -DPFG_EXP_PLAIN=1 -- device generator: 1024 x 4 -4.7 %, one wave x 2
    // -2.8 %, 256 x 4 -0.6 %, 512 x 2 +2.1 %, large-N kernel 0; REPLAY arithmetic legs: SVM 256 x 4 -5.8 %, LGSSM one wave
    // -6.5 %, GARCH 256 x 4 +8 %: profiles/r04_ab_score1_twin.txt)
    // ... and the 1024 x 1 latency variant (one window alone, SVM T = N = 1000): device 1.957 -> 1.814 ms, REPLAY 2.768 -> 2.623)
    constexpr bool lat_unit = MODEL != PFG_MODEL_GARCH && NT == 1024 && PPT == 1 && PP;
    constexpr bool dev_unit = RNG == PFG_RNG_DEVICE && !PP && ((NT == 1024 && PPT == 4) || (NT == 64 && PPT == 2));
    constexpr bool rep_unit = RNG == PFG_RNG_REPLAY && MODEL != PFG_MODEL_GARCH && ((NT == 256 && PPT == 4) || (NT == 64 && PPT == 2));
    if constexpr (sizeof(REAL) == 8 && (dev_unit || rep_unit || lat_unit)) {
        if (!traced && ctx->score1) {
            ctx->last_variant = NT == 1024 ? (PPT == 4 ? "wg1024x4s_score1" : "wg1024x1_score1")
                                : NT == 256 ? (PP ? "wg256x4_score1" : "wg256x4s_score1") : (PP ? "wg64x2_score1" : "wg64x2s_score1");
            return launch_one_t<MODEL, KERNEL, REAL, NT, PPT, RNG, PP, false, true>(ctx, n_max, B, dp, st);
        }
    }
    if (!traced) return launch_one_t<MODEL, KERNEL, REAL, NT, PPT, RNG, PP, false>(ctx, n_max, B, dp, st);
    return launch_one_t<MODEL, KERNEL, REAL, NT, PPT, RNG, PP, true>(ctx, n_max, B, dp, st);
}
template <int MODEL, int KERNEL, typename REAL, int RNG>
int launch_v(pfg_ctx *ctx, int v, int n_max, int B, const pfg_dev_problem *dp, hipStream_t st, bool traced) {
    switch (v) {
        case 0: return launch_one<MODEL, KERNEL, REAL, 256, 1, RNG, true>(ctx, n_max, B, dp, st, traced);
        case 1: return launch_one<MODEL, KERNEL, REAL, 256, 4, RNG, true>(ctx, n_max, B, dp, st, traced);
        case 2: return launch_one<MODEL, KERNEL, REAL, 256, 4, RNG, false>(ctx, n_max, B, dp, st, traced);
        case 3: return launch_one<MODEL, KERNEL, REAL, 1024, 1, RNG, true>(ctx, n_max, B, dp, st, traced);
        case 5: return launch_one<MODEL, KERNEL, REAL, 64, 2, RNG, true>(ctx, n_max, B, dp, st, traced);
        case 4:
            if constexpr (RNG == PFG_RNG_DEVICE) return launch_one<MODEL, KERNEL, REAL, 1024, 4, RNG, false>(ctx, n_max, B, dp, st, traced);
            break;
        case 7:
            if constexpr (RNG == PFG_RNG_DEVICE) return launch_one<MODEL, KERNEL, REAL, 64, 4, RNG, true>(ctx, n_max, B, dp, st, traced);
            break;
        case 6:
            if constexpr (RNG == PFG_RNG_DEVICE && MODEL == PFG_MODEL_GARCH && sizeof(REAL) == 8)
                return launch_one<MODEL, KERNEL, REAL, 512, 2, RNG, false>(ctx, n_max, B, dp, st, traced);
            break;
        // one wave per window on ONE state buffer (a wave's LDS accesses execute in order: the gather of a step is
        // over before its stores are issued, the fourth "barrier" is free) -- half the LDS per window
        case 8:
            if constexpr (RNG == PFG_RNG_DEVICE) return launch_one<MODEL, KERNEL, REAL, 64, 2, RNG, false>(ctx, n_max, B, dp, st, traced);
            break;
        case 9:
            if constexpr (RNG == PFG_RNG_DEVICE) return launch_one<MODEL, KERNEL, REAL, 64, 4, RNG, false>(ctx, n_max, B, dp, st, traced);
            break;
    }
    return fail(ctx, PFG_ERR_UNSUPPORTED, "no kernel variant");
}

template <int MODEL, int KERNEL, typename REAL, int NT, int PPT, int RNG>
int launch_paris_one(pfg_ctx *ctx, int n_max, int B, const pfg_dev_problem *dp, hipStream_t st) {
    auto kern = pfg::pf_reg_kernel<MODEL, KERNEL, REAL, NT, PPT, RNG, true, pfg::MODE_PARIS>;
    size_t lds = pfg::reg_kernel_lds_bytes<MODEL, REAL, NT, PPT, RNG, true, pfg::MODE_PARIS>(n_max);
    if (lds > kLdsLimit)
        return fail(ctx, PFG_ERR_UNSUPPORTED, "pf = 'paris': N = " + std::to_string(n_max) + " does not fit the LDS-resident variant");
    PFG_ENSURE_LDS(ctx, kern, lds);
    hipLaunchKernelGGL(kern, dim3(B), dim3(NT), lds, st, dp);
    PFG_HIP(ctx, hipGetLastError());
    return PFG_OK;
}

template <int MODEL, int KERNEL, typename REAL, int RNG>
int launch_paris(pfg_ctx *ctx, int n_max, int B, const pfg_dev_problem *dp, hipStream_t st) {
    if (n_max <= 256) return launch_paris_one<MODEL, KERNEL, REAL, 256, 1, RNG>(ctx, n_max, B, dp, st);
    if (n_max <= 1024) return launch_paris_one<MODEL, KERNEL, REAL, 256, 4, RNG>(ctx, n_max, B, dp, st);
    if (n_max > pfg::MEM_MAX_N)
        return fail(ctx, PFG_ERR_UNSUPPORTED, "pf = 'paris' is implemented for N <= 16384 (N = " + std::to_string(n_max) + ")");
    // large-N kernel, PaRIS instantiation (state in the HBM scratch; descriptors must carry one)
    auto kern = pfg::pf_mem_kernel<MODEL, KERNEL, REAL, RNG, true>;
    size_t lds = pfg::mem_kernel_lds_bytes<REAL, RNG>(n_max);
    PFG_ENSURE_LDS(ctx, kern, lds);
    hipLaunchKernelGGL(kern, dim3(B), dim3(pfg::MEM_NT), lds, st, dp);
    PFG_HIP(ctx, hipGetLastError());
    return PFG_OK;
}


// large-N kernel, device-RNG fast path (thread-major CDF, unrolled search, two chunks in flight)
template <int MODEL, int KERNEL, typename REAL, int NP2>
int launch_big_one(pfg_ctx *ctx, int B, const pfg_dev_problem *dp, hipStream_t st) {
    auto kern = pfg::pf_big_kernel<MODEL, KERNEL, REAL, NP2>;
    size_t lds = pfg::big_kernel_lds_bytes<REAL>(NP2);
    PFG_ENSURE_LDS(ctx, kern, lds);
    hipLaunchKernelGGL(kern, dim3(B), dim3(pfg::MEM_NT), lds, st, dp);
    PFG_HIP(ctx, hipGetLastError());
    return PFG_OK;
}

template <int MODEL, int KERNEL, typename REAL>
int launch_big(pfg_ctx *ctx, int n_max, int B, const pfg_dev_problem *dp, hipStream_t st) {
    if (n_max <= 4096) return launch_big_one<MODEL, KERNEL, REAL, 4096>(ctx, B, dp, st);
    return launch_big_one<MODEL, KERNEL, REAL, 16384>(ctx, B, dp, st);
}

// O(N^2) Poyiadjis smoother instantiations (ping-pong variants, parents' log-weights in LDS)
template <int MODEL, int KERNEL, typename REAL, int NT, int PPT, int RNG>
int launch_n2_one(pfg_ctx *ctx, int n_max, int B, const pfg_dev_problem *dp, hipStream_t st) {
    auto kern = pfg::pf_reg_kernel<MODEL, KERNEL, REAL, NT, PPT, RNG, true, pfg::MODE_N2>;
    size_t lds = pfg::reg_kernel_lds_bytes<MODEL, REAL, NT, PPT, RNG, true, pfg::MODE_N2>(n_max);
    if (lds > kLdsLimit)
        return fail(ctx, PFG_ERR_UNSUPPORTED, "pf = 'poyiadjis_N2': N = " + std::to_string(n_max) + " does not fit the LDS-resident variant");
    PFG_ENSURE_LDS(ctx, kern, lds);
    hipLaunchKernelGGL(kern, dim3(B), dim3(NT), lds, st, dp);
    PFG_HIP(ctx, hipGetLastError());
    return PFG_OK;
}

template <int MODEL, int KERNEL, typename REAL, int RNG>
int launch_n2(pfg_ctx *ctx, int n_max, int B, const pfg_dev_problem *dp, hipStream_t st) {
    if (n_max <= 256) return launch_n2_one<MODEL, KERNEL, REAL, 256, 1, RNG>(ctx, n_max, B, dp, st);
    if (n_max <= 1024) return launch_n2_one<MODEL, KERNEL, REAL, 256, 4, RNG>(ctx, n_max, B, dp, st);
    if (n_max <= pfg::MEM_MAX_N) {
        // large-N kernel, PaRIS instantiation (second log-weight array): its O(N^2) sweep (descriptors carry a scratch)
        auto kern = pfg::pf_mem_kernel<MODEL, KERNEL, REAL, RNG, true>;
        size_t lds = pfg::mem_kernel_lds_bytes<REAL, RNG>(n_max);
        PFG_ENSURE_LDS(ctx, kern, lds);
        hipLaunchKernelGGL(kern, dim3(B), dim3(pfg::MEM_NT), lds, st, dp);
        PFG_HIP(ctx, hipGetLastError());
        return PFG_OK;
    }
    return fail(ctx, PFG_ERR_UNSUPPORTED, "pf = 'poyiadjis_N2' is implemented for N <= 16384 (N = " + std::to_string(n_max) + ")");
}

// systematic-resampling instantiation (extension): device RNG, the fp64 / f32 default 256x4 variants
template <int MODEL, int KERNEL, typename REAL, bool PP>
int launch_systematic(pfg_ctx *ctx, int n_max, int B, const pfg_dev_problem *dp, hipStream_t st) {
    if (n_max > 1024) return fail(ctx, PFG_ERR_UNSUPPORTED, "systematic resampling is built for N <= 1024");
    auto kern = pfg::pf_reg_kernel<MODEL, KERNEL, REAL, 256, 4, PFG_RNG_DEVICE, PP, pfg::MODE_SYSTEMATIC>;
    size_t lds = pfg::reg_kernel_lds_bytes<MODEL, REAL, 256, 4, PFG_RNG_DEVICE, PP, pfg::MODE_SYSTEMATIC>(n_max);
    if (lds > kLdsLimit) return fail(ctx, PFG_ERR_UNSUPPORTED, "systematic resampling: state does not fit LDS");
    PFG_ENSURE_LDS(ctx, kern, lds);
    hipLaunchKernelGGL(kern, dim3(B), dim3(256), lds, st, dp);
    PFG_HIP(ctx, hipGetLastError());
    return PFG_OK;
}

// lw4: every window of the batch has N <= 4096 and none asks for the predictive statistic (the dispatcher knows): the
// variant that keeps the log-weights in registers
template <int MODEL, int KERNEL, typename REAL, int RNG>
int launch_mem(pfg_ctx *ctx, int n_max, int B, const pfg_dev_problem *dp, hipStream_t st, bool lw4 = false) {
    size_t lds = pfg::mem_kernel_lds_bytes<REAL, RNG>(n_max);
    if (lw4 && n_max <= 4096 && ctx->score1 && sizeof(REAL) == 8 && MODEL != PFG_MODEL_GARCH) {      // (GARCH: unmeasured here, +8 % in the LDS-resident REPLAY unit)
        auto kern = pfg::pf_mem_kernel<MODEL, KERNEL, REAL, RNG, false, true, true>;      // the score-only twin (see SCORE1)
        PFG_ENSURE_LDS(ctx, kern, lds);
        hipLaunchKernelGGL(kern, dim3(B), dim3(pfg::MEM_NT), lds, st, dp);
        ctx->last_variant = "mem1024_score1";
    } else if (lw4 && n_max <= 4096) {
        auto kern = pfg::pf_mem_kernel<MODEL, KERNEL, REAL, RNG, false, true>;
        PFG_ENSURE_LDS(ctx, kern, lds);
        hipLaunchKernelGGL(kern, dim3(B), dim3(pfg::MEM_NT), lds, st, dp);
    } else {
        auto kern = pfg::pf_mem_kernel<MODEL, KERNEL, REAL, RNG>;
        PFG_ENSURE_LDS(ctx, kern, lds);
        hipLaunchKernelGGL(kern, dim3(B), dim3(pfg::MEM_NT), lds, st, dp);
    }
    PFG_HIP(ctx, hipGetLastError());
    return PFG_OK;
}


// ---- whole-GPU window (pfg_grid_kernel.hpp): init, then per timestep [REPLAY: the reference's CDF] + the step kernel,
// then finish; T_max + 2 (REPLAY: 5 T_max + 2) launches on `st`, no host synchronisation in between.  Every window of the
// batch must fall into the same tile class (pfg::grid_ppt(N)); windows shorter than t_max leave their launches at once.
// phase: PFG_GRID_ALL = init, every timestep, finish; PFG_GRID_INIT / PFG_GRID_FINISH alone; t >= 0: timestep t alone
// (callers that put events or graph nodes between the launches)
constexpr int PFG_GRID_ALL = -1, PFG_GRID_INIT = -2, PFG_GRID_FINISH = -3;
template <int MODEL, int KERNEL, typename REAL, int RNG, int NT, int PPT, int KMAX>
int launch_grid_ppt(pfg_ctx *ctx, int n_max, int t_max, int B, const pfg_dev_problem *dp, hipStream_t st, int phase) {
    constexpr int NW = NT / pfg::WAVE;
    const pfg::GridLayout L = pfg::grid_layout<MODEL, REAL>(n_max, RNG == PFG_RNG_REPLAY);
    const dim3 grid((unsigned)L.G, (unsigned)B), blk(NT);
    const size_t red_bytes = (size_t)(4 * PPT * NW + 16 + PFG_MAX_STAT * NW) * 8;
    const size_t lds_init = red_bytes + pfg::tab_bytes<REAL, RNG, (RNG == PFG_RNG_DEVICE)>();
    const size_t lds_fin = (size_t)(pfg::GRID_MAX_TILES + 1) * 8 + red_bytes;
    auto k_init = pfg::pfg_grid_init_kernel<MODEL, KERNEL, REAL, RNG, NT, PPT>;
    auto k_fin = pfg::pfg_grid_finish_kernel<MODEL, REAL, RNG, NT, PPT>;
    const int t_lo = phase >= 0 ? phase : 0, t_hi = phase >= 0 ? phase + 1 : (phase == PFG_GRID_ALL ? t_max : 0);
    if (phase == PFG_GRID_ALL || phase == PFG_GRID_INIT) hipLaunchKernelGGL(k_init, grid, blk, lds_init, st, dp);
    if constexpr (RNG == PFG_RNG_REPLAY) {
        auto k_step = pfg::pfg_grid_step_kernel<MODEL, KERNEL, REAL, RNG, NT, PPT>;
        const size_t lds_step = pfg::grid_step_lds_bytes<NT, PPT, REAL, RNG>(L.C);
        PFG_ENSURE_LDS(ctx, k_step, lds_step);
        // the reference's CDF: four launches that spread the particle axis over the GPU (PFGRAD_CDF_SINGLE=1: the
        // lone-workgroup kernel, A/B and cross-check)
        const char *single_env = std::getenv("PFGRAD_CDF_SINGLE");
        const bool cdf_single = single_env && single_env[0] == '1';
        const unsigned n16 = (unsigned)((n_max + 16383) / 16384), n4 = (unsigned)((n_max + pfg::CDF_BLK - 1) / pfg::CDF_BLK);
        for (int t = t_lo; t < t_hi; ++t) {
            if (cdf_single) {
                hipLaunchKernelGGL((pfg::pfg_grid_cdf_kernel<MODEL, REAL>), dim3((unsigned)B), dim3(pfg::CDF_NT), 0, st, dp, t);
            } else {
                hipLaunchKernelGGL((pfg::pfg_grid_cdf_sum_kernel<MODEL, REAL>), dim3(n16, (unsigned)B), dim3(pfg::CDF_NT), 0, st, dp, t);
                hipLaunchKernelGGL((pfg::pfg_grid_cdf_class_kernel<MODEL, REAL>), dim3(n4, (unsigned)B), dim3(pfg::CDF_NT), 0, st, dp, t);
                hipLaunchKernelGGL((pfg::pfg_grid_cdf_chain_kernel<MODEL, REAL>), dim3((unsigned)B), dim3(pfg::WAVE), 0, st, dp, t);
                hipLaunchKernelGGL((pfg::pfg_grid_cdf_apply_kernel<MODEL, REAL>), dim3(n4, (unsigned)B), dim3(pfg::CDF_NT), 0, st, dp, t);
            }
            hipLaunchKernelGGL(k_step, grid, blk, lds_step, st, dp, t);
        }
    } else {
        const size_t lds_step = pfg::grid_dev_lds_doubles<NT, PPT>(L.G) * 8;
        if (ctx->score1) {                  // PFG_SMOOTHER_POYIADJIS_N: the score-only twin of the step kernel
            auto k_step = pfg::pfg_grid_step_dev_kernel<MODEL, KERNEL, REAL, NT, PPT, KMAX, true>;
            PFG_ENSURE_LDS(ctx, k_step, lds_step);
            for (int t = t_lo; t < t_hi; ++t) hipLaunchKernelGGL(k_step, grid, blk, lds_step, st, dp, t);
        } else {
            auto k_step = pfg::pfg_grid_step_dev_kernel<MODEL, KERNEL, REAL, NT, PPT, KMAX>;
            PFG_ENSURE_LDS(ctx, k_step, lds_step);
            for (int t = t_lo; t < t_hi; ++t) hipLaunchKernelGGL(k_step, grid, blk, lds_step, st, dp, t);
        }
    }
    if (phase == PFG_GRID_ALL || phase == PFG_GRID_FINISH) hipLaunchKernelGGL(k_fin, grid, blk, lds_fin, st, dp);
    PFG_HIP(ctx, hipGetLastError());
    return PFG_OK;
}

template <int MODEL, int KERNEL, int RNG>
int launch_grid_mkr(pfg_ctx *ctx, int dtype, int n_max, int t_max, int B, const pfg_dev_problem *dp, hipStream_t st, int phase) {
    if (n_max > pfg::GRID_MAX_N)
        return fail(ctx, PFG_ERR_UNSUPPORTED, "N = " + std::to_string(n_max) + " exceeds the supported maximum of " + std::to_string(pfg::GRID_MAX_N));
    const int ppt = pfg::grid_ppt(n_max), kmax = pfg::grid_kmax(n_max);
    constexpr int NT = pfg::GRID_NT;
#define PFG_GRID_CASE(REAL_)                                                                                              \
    (ppt == 4 ? launch_grid_ppt<MODEL, KERNEL, REAL_, RNG, NT, 4, 2>(ctx, n_max, t_max, B, dp, st, phase)               \
     : kmax == 2 ? launch_grid_ppt<MODEL, KERNEL, REAL_, RNG, NT, 8, 2>(ctx, n_max, t_max, B, dp, st, phase)            \
                 : launch_grid_ppt<MODEL, KERNEL, REAL_, RNG, NT, 8, 8>(ctx, n_max, t_max, B, dp, st, phase))
    if (dtype == PFG_F64) return PFG_GRID_CASE(double);
    return PFG_GRID_CASE(float);
#undef PFG_GRID_CASE
}

// every kernel of one (model, proposal kernel, generator): explicitly instantiated in
// pfg_inst_<model>_<kernel>_<rng>.hip.  The DEVICE-generator units are compiled with
// -ffp-contract=fast (no operation-order parity to keep there), the REPLAY units with
// -ffp-contract=off (the reference's NumPy operation order).
template <int MODEL, int KERNEL, int RNG>
int launch_mkr(pfg_ctx *ctx, int dtype, int v, int n_max, int B, const pfg_dev_problem *dp, hipStream_t st, bool traced) {
    if (v == kVariantSystematic) {
        if constexpr (RNG != PFG_RNG_DEVICE) {
            return fail(ctx, PFG_ERR_UNSUPPORTED, "systematic resampling needs the DEVICE rng");
        } else {
            if (dtype == PFG_F64) return launch_systematic<MODEL, KERNEL, double, false>(ctx, n_max, B, dp, st);
            return launch_systematic<MODEL, KERNEL, float, true>(ctx, n_max, B, dp, st);
        }
    }
    if (v == kVariantBig) {
        if constexpr (RNG != PFG_RNG_DEVICE) {
            return fail(ctx, PFG_ERR_UNSUPPORTED, "the large-N fast path needs the DEVICE rng");
        } else {
            if (dtype == PFG_F64) return launch_big<MODEL, KERNEL, double>(ctx, n_max, B, dp, st);
            return launch_big<MODEL, KERNEL, float>(ctx, n_max, B, dp, st);
        }
    }
    if (v == kVariantN2) {
        if (dtype == PFG_F64) return launch_n2<MODEL, KERNEL, double, RNG>(ctx, n_max, B, dp, st);
        return launch_n2<MODEL, KERNEL, float, RNG>(ctx, n_max, B, dp, st);
    }
    if (v == kVariantParis) {
        if (dtype == PFG_F64) return launch_paris<MODEL, KERNEL, double, RNG>(ctx, n_max, B, dp, st);
        return launch_paris<MODEL, KERNEL, float, RNG>(ctx, n_max, B, dp, st);
    }
    if (v == kVariantMem || v == kVariantMemLw4) {
        if (dtype == PFG_F64) return launch_mem<MODEL, KERNEL, double, RNG>(ctx, n_max, B, dp, st, v == kVariantMemLw4);
        return launch_mem<MODEL, KERNEL, float, RNG>(ctx, n_max, B, dp, st, v == kVariantMemLw4);
    }
    if (dtype == PFG_F64) return launch_v<MODEL, KERNEL, double, RNG>(ctx, v, n_max, B, dp, st, traced);
    return launch_v<MODEL, KERNEL, float, RNG>(ctx, v, n_max, B, dp, st, traced);
}

}  // namespace pfg_host
