// Device code of libpfgrad: one persistent workgroup runs one whole buffered particle-filter
// window (the T-loop of particle_filters/buffered_smoother.py:93-133) in a single launch,
// particle state resident in registers + LDS.  gfx950 only (wave64, 160 KiB LDS).
//
// Compiled with -ffp-contract=off: the f64 instantiation follows the reference's NumPy
// expression order operation by operation so that REPLAY runs agree with the reference to
// rounding of exp/log and of the (parallel) weight sum / prefix scan only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pfgrad.h"

namespace pfg {

constexpr int WAVE = 64;
constexpr double LOG_2PI = 1.8378770664093453;   // log(2*pi)
constexpr double TWO_PI = 6.283185307179586;

// ------------------------------------------------------------------------------------
// wave-level primitives (64 lanes)
// ------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        T o = __shfl_xor(v, d, WAVE);
        v = o > v ? o : v;
    }
    return v;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, WAVE);
    return v;
}

__device__ __forceinline__ double wave_incl_scan(double v, int lane) {
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
        double o = __shfl_up(v, d, WAVE);
        if (lane >= d) v += o;
    }
    return v;
}

// ------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011), counter-based: no state to carry between steps.
// ------------------------------------------------------------------------------------
struct u32x4 { uint32_t x, y, z, w; };

__device__ __forceinline__ u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x;
        uint32_t hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
        u32x4 n;
        n.x = hi1 ^ c.y ^ k0; n.y = lo1; n.z = hi0 ^ c.w ^ k1; n.w = lo0;
        c = n; k0 += W0; k1 += W1;
    }
    return c;
}

// 53-bit uniform in [0,1) from two words, the construction NumPy's random_sample uses
__device__ __forceinline__ double u01_53(uint32_t a, uint32_t b) {
    return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

// standard normal by Box-Muller from two words
template <typename REAL>
__device__ __forceinline__ REAL normal_bm(uint32_t a, uint32_t b);
template <>
__device__ __forceinline__ double normal_bm<double>(uint32_t a, uint32_t b) {
    double u1 = ((double)a + 1.0) * (1.0 / 4294967296.0);     // (0,1]
    double u2 = (double)b * (1.0 / 4294967296.0);             // [0,1)
    return sqrt(-2.0 * log(u1)) * cospi(2.0 * u2);
}
template <>
__device__ __forceinline__ float normal_bm<float>(uint32_t a, uint32_t b) {
    float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);  // (0,1], 24 bits
    float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * __logf(u1)) * cospif(2.0f * u2);
}

// ------------------------------------------------------------------------------------
// models.  Consts are derived from raw theta exactly as the reference's Parameters
// properties do (variables/covariance.py:128-157, variables/garch_var.py:69-91).
// ------------------------------------------------------------------------------------
template <int MODEL> struct ModelDims;
template <> struct ModelDims<PFG_MODEL_SVM>   { static constexpr int NS = 1, H = 3; };
template <> struct ModelDims<PFG_MODEL_GARCH> { static constexpr int NS = 2, H = 4; };
template <> struct ModelDims<PFG_MODEL_LGSSM> { static constexpr int NS = 1, H = 4; };

template <typename REAL> struct Consts {
    // common
    REAL LRinv, iLRinv, Rinv, R, logLRinv, c0;          // c0 = -0.5*log(2pi)
    // svm / lgssm
    REAL A, C, LQinv, iLQinv, Qinv;
    REAL opt_sd, opt_prec, opt_var, opt_logvar;         // lgssm optimal kernel
    // garch
    REAL mu, phi, lam, alpha, beta, gamma;
};

template <int MODEL, typename REAL>
__device__ __forceinline__ Consts<REAL> make_consts(const double *__restrict__ th) {
    Consts<double> d = {};
    d.c0 = -0.5 * LOG_2PI;
    double LRinv;
    if (MODEL == PFG_MODEL_SVM) {
        d.A = th[0]; d.LQinv = th[1]; LRinv = th[2];
    } else if (MODEL == PFG_MODEL_LGSSM) {
        d.A = th[0]; d.C = th[1]; d.LQinv = th[2]; LRinv = th[3];
    } else {
        LRinv = th[3];
        d.mu = exp(th[0]);
        d.phi = 1.0 / (1.0 + exp(-th[1]));
        d.lam = 1.0 / (1.0 + exp(-th[2]));
        d.alpha = d.mu * (1.0 - d.phi);
        d.beta = d.phi * d.lam;
        d.gamma = d.phi * (1.0 - d.lam);
    }
    d.LRinv = LRinv;
    d.iLRinv = 1.0 / LRinv;
    d.Rinv = LRinv * LRinv + 1e-16;
    d.R = 1.0 / d.Rinv;
    d.logLRinv = log(LRinv);
    if (MODEL != PFG_MODEL_GARCH) {
        d.iLQinv = 1.0 / d.LQinv;
        d.Qinv = d.LQinv * d.LQinv + 1e-16;
    }
    if (MODEL == PFG_MODEL_LGSSM) {
        d.opt_prec = d.Qinv + (d.C * d.C) * d.Rinv;
        d.opt_sd = pow(d.opt_prec, -0.5);
        d.opt_var = 1.0 / d.Qinv + 1.0 / d.Rinv;
        d.opt_logvar = log(d.opt_var);
    }
    Consts<REAL> c;
    c.LRinv = (REAL)d.LRinv; c.iLRinv = (REAL)d.iLRinv; c.Rinv = (REAL)d.Rinv; c.R = (REAL)d.R;
    c.logLRinv = (REAL)d.logLRinv; c.c0 = (REAL)d.c0;
    c.A = (REAL)d.A; c.C = (REAL)d.C; c.LQinv = (REAL)d.LQinv; c.iLQinv = (REAL)d.iLQinv;
    c.Qinv = (REAL)d.Qinv;
    c.opt_sd = (REAL)d.opt_sd; c.opt_prec = (REAL)d.opt_prec; c.opt_var = (REAL)d.opt_var;
    c.opt_logvar = (REAL)d.opt_logvar;
    c.mu = (REAL)d.mu; c.phi = (REAL)d.phi; c.lam = (REAL)d.lam;
    c.alpha = (REAL)d.alpha; c.beta = (REAL)d.beta; c.gamma = (REAL)d.gamma;
    return c;
}

__device__ __forceinline__ double exp_r(double v) { return exp(v); }
__device__ __forceinline__ float exp_r(float v) { return __expf(v); }
__device__ __forceinline__ double log_r(double v) { return log(v); }
__device__ __forceinline__ float log_r(float v) { return __logf(v); }
__device__ __forceinline__ double sqrt_r(double v) { return sqrt(v); }
__device__ __forceinline__ float sqrt_r(float v) { return sqrtf(v); }

// One particle: parent state xp -> proposal x' (Kernel.rv), log weight (Kernel.reweight) and
// additive statistic (score or sufficient statistic), all from the same registers.
// stat: PFG_STAT_*; `inside` = t in [t1,tL).  add[] is NOT yet scaled by weight_t.
template <int MODEL, int KERNEL, typename REAL>
__device__ __forceinline__ void particle_step(const Consts<REAL> &c, const REAL *xp, REAL y, REAL z,
                                              int stat, bool inside, REAL *xn, REAL &lw, REAL *add) {
    constexpr int H = ModelDims<MODEL>::H;
    const REAL half = (REAL)0.5;
#pragma unroll
    for (int h = 0; h < H; ++h) add[h] = (REAL)0;
    if (MODEL == PFG_MODEL_SVM) {
        // svm/kernels.py:34-37, :56-62; svm/helper.py:342-348
        REAL xpA = xp[0] * c.A;
        REAL x1 = c.iLQinv * z + xpA;
        REAL e = exp_r(-x1);
        REAL y2 = y * y;
        lw = ((c.c0 + ((-half * y2) * e) * c.Rinv) + c.logLRinv) + (-half * x1);
        xn[0] = x1;
        if (inside) {
            if (stat == PFG_STAT_SCORE) {
                REAL dx = x1 - c.A * xp[0];
                add[2] = (c.Qinv * dx) * xp[0];
                add[1] = c.iLQinv - (dx * dx) * c.LQinv;
                REAL dy2 = y2 * e;                       // y^2 / exp(x')
                add[0] = c.iLRinv - dy2 * c.LRinv;
            } else if (stat == PFG_STAT_SUFF) {
                add[0] = x1; add[1] = x1 * x1; add[2] = xp[0] * x1;
            }
        }
    } else if (MODEL == PFG_MODEL_LGSSM) {
        REAL x1;
        if (KERNEL == PFG_KERNEL_PRIOR) {
            // lgssm/kernels.py:30-33, :58-62
            x1 = c.iLQinv * z + xp[0] * c.A;
            REAL diff = y - c.C * x1;
            lw = (c.c0 + (-half * (diff * diff)) * c.Rinv) + c.logLRinv;
        } else {
            // lgssm/kernels.py:87-97, :117-120
            REAL mp = (xp[0] * c.A) * c.Qinv + (y * c.C) * c.Rinv;
            x1 = c.opt_sd * z + mp / c.opt_prec;
            REAL diff = y - c.A * xp[0];
            lw = ((-half * (diff * diff)) / c.opt_var - half * (REAL)LOG_2PI) - half * c.opt_logvar;
        }
        xn[0] = x1;
        if (inside) {
            if (stat == PFG_STAT_SCORE) {
                // lgssm/helper.py:1270-1277, order [LRinv, LQinv, C, A]
                REAL dx = x1 - c.A * xp[0];
                add[3] = (c.Qinv * dx) * xp[0];
                add[1] = c.iLQinv - (dx * dx) * c.LQinv;
                REAL dy = y - c.C * x1;
                add[2] = (c.Rinv * dy) * x1;
                add[0] = c.iLRinv - (dy * dy) * c.LRinv;
            } else if (stat == PFG_STAT_SUFF) {
                add[0] = x1; add[1] = x1 * x1; add[2] = xp[0] * x1;
            }
        }
    } else {
        // garch/kernels.py:60-68 / :146-156, reweight :83-88 / :172-178
        REAL xx = xp[0] * xp[0];
        REAL s2 = (c.alpha + c.beta * xx) + c.gamma * xp[1];
        REAL x1;
        if (KERNEL == PFG_KERNEL_PRIOR) {
            x1 = sqrt_r(s2) * z;
            REAL diff = y - x1;
            lw = (c.c0 + (-half * (diff * diff)) * c.Rinv) + c.logLRinv;
        } else {
            REAL var = (REAL)1 / (c.Rinv + (REAL)1 / s2);
            REAL mean = var * (y * c.Rinv);
            x1 = mean + sqrt_r(var) * z;
            REAL v2 = s2 + c.R;
            lw = (c.c0 + (-half * (y * y)) / v2) + (-half * log_r(v2));
        }
        xn[0] = x1; xn[1] = s2;
        if (inside) {
            if (stat == PFG_STAT_SCORE) {
                // garch/helper.py:350-370, order [LRinv, log_mu, logit_phi, logit_lambduh]
                REAL v = s2;
                REAL gv = (-half * (v - x1 * x1)) / (v * v);
                add[1] = (gv * ((REAL)1 - c.phi)) * c.mu;
                add[2] = ((gv * ((-c.mu + c.lam * xx) + ((REAL)1 - c.lam) * xp[1])) * ((REAL)1 - c.phi)) * c.phi;
                add[3] = (((gv * c.phi) * (xx - xp[1])) * ((REAL)1 - c.lam)) * c.lam;
                REAL dy = y - x1;
                add[0] = c.iLRinv - (dy * dy) * c.LRinv;
            } else if (stat == PFG_STAT_SUFF) {
                REAL x2 = x1 * x1;
                add[0] = x1; add[1] = x2; add[2] = x2 * x2;
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// LDS-resident kernel: N <= NT*PPT particles, particle i = k*NT + tid held by thread tid in
// register slot k.  LDS: cdf[NL] f64 | x[NS][NL] | stats[H][NL] | reduction scratch.
// 4 workgroup barriers per timestep.
// ------------------------------------------------------------------------------------
template <int NT, int PPT> struct RegLayout {
    static constexpr int NW = NT / WAVE;
    static constexpr int RED = PPT * NW + NW + PFG_MAX_STAT * NW + 8;  // doubles of scratch
};

template <int MODEL, typename REAL, int NT, int PPT>
__host__ __device__ inline size_t reg_kernel_lds_bytes(int N) {
    int NL = (N + WAVE - 1) / WAVE * WAVE;
    return (size_t)NL * 8 + (size_t)NL * (ModelDims<MODEL>::NS + ModelDims<MODEL>::H) * sizeof(REAL) +
           (size_t)RegLayout<NT, PPT>::RED * 8;
}

template <int MODEL, int KERNEL, typename REAL, int NT, int PPT, int RNG>
__global__ __launch_bounds__(NT) void pf_reg_kernel(const pfg_dev_problem *__restrict__ probs) {
    constexpr int NS = ModelDims<MODEL>::NS;
    constexpr int H = ModelDims<MODEL>::H;
    constexpr int NW = NT / WAVE;
    extern __shared__ __align__(16) unsigned char smem[];

    const pfg_dev_problem &P = probs[blockIdx.x];
    const int N = P.N, T = P.T, t1 = P.t1, tL = P.tL;
    const int NL = (N + WAVE - 1) / WAVE * WAVE;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
    const bool is_filter = (P.smoother == PFG_SMOOTHER_FILTER);
    const int stat = P.stat;
    const double lam_d = is_filter ? 0.0 : P.lambduh;
    const REAL lam = (REAL)lam_d, oml = (REAL)(1.0 - lam_d);
    const bool needS_every = is_filter || (lam_d != 1.0);

    double *cdf = reinterpret_cast<double *>(smem);
    REAL *xL = reinterpret_cast<REAL *>(cdf + NL);
    REAL *sL = xL + (size_t)NS * NL;
    double *red = reinterpret_cast<double *>(sL + (size_t)H * NL);
    double *red_scan = red;                 // [PPT*NW]
    double *red_max = red + PPT * NW;       // [NW]
    double *red_S = red_max + NW;           // [H*NW]

    const Consts<REAL> c = make_consts<MODEL, REAL>(P.theta);
    int np2 = 1;
    while (np2 < N) np2 <<= 1;

    const uint32_t k0 = (uint32_t)P.seed, k1 = (uint32_t)(P.seed >> 32);
    const uint64_t stepc = P.step_ctr ? *P.step_ctr : 0ull;
    const uint32_t ctr_z = (uint32_t)P.stream ^ (uint32_t)(stepc << 20);
    const uint32_t ctr_w = (uint32_t)(P.stream >> 32) ^ (uint32_t)(stepc >> 12);

    REAL x[PPT][NS], lw[PPT], s[PPT][H];
    // ---- x0 (kernels.py:83-100, garch/kernels.py:7-18) or warm start ------------------
    {
        double pv = P.prior_var;
        if (MODEL == PFG_MODEL_GARCH && (P.flags & PFG_FLAG_GARCH_STATIONARY_PRIOR))
            pv = (double)c.alpha / (1.0 - (double)c.beta - (double)c.gamma);
        const double sd = sqrt(pv);
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = k * NT + tid;
            const bool valid = i < N;
#pragma unroll
            for (int d = 0; d < NS; ++d) x[k][d] = (REAL)0;
#pragma unroll
            for (int h = 0; h < H; ++h) s[k][h] = (REAL)0;
            lw[k] = (REAL)0;
            if (valid) {
                if (P.init_x) {
#pragma unroll
                    for (int d = 0; d < NS; ++d) x[k][d] = (REAL)P.init_x[(size_t)i * NS + d];
                    lw[k] = (REAL)P.init_logw[i];
                    if (P.init_stats && !is_filter) {
#pragma unroll
                        for (int h = 0; h < H; ++h) s[k][h] = (REAL)P.init_stats[(size_t)i * H + h];
                    }
                } else {
                    double z0;
                    if (RNG == PFG_RNG_REPLAY) z0 = P.z0[i];
                    else {
                        u32x4 r = philox4x32_10({(uint32_t)i, 0u, ctr_z, ctr_w}, k0, k1);
                        z0 = (double)normal_bm<REAL>(r.z, r.w);
                    }
                    x[k][0] = (REAL)(P.prior_mean + sd * z0);
                }
#pragma unroll
                for (int d = 0; d < NS; ++d) xL[(size_t)d * NL + i] = x[k][d];
#pragma unroll
                for (int h = 0; h < H; ++h) sL[(size_t)h * NL + i] = s[k][h];
            }
        }
    }
    auto trace_state = [&](int row) {
        if (!P.trace_x) return;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = k * NT + tid;
            if (i < N) {
#pragma unroll
                for (int d = 0; d < NS; ++d) P.trace_x[((size_t)row * N + i) * NS + d] = (double)x[k][d];
                P.trace_logw[(size_t)row * N + i] = (double)lw[k];
                if (P.trace_stats && !is_filter) {
#pragma unroll
                    for (int h = 0; h < H; ++h)
                        P.trace_stats[((size_t)row * N + i) * H + h] = (double)s[k][h];
                }
            }
        }
    };
    trace_state(0);
    if (P.trace_ll && tid == 0) P.trace_ll[0] = 0.0;

    double ll = 0.0, wt_prev = 1.0, tie = 1.0;
    double filt[H];
#pragma unroll
    for (int h = 0; h < H; ++h) filt[h] = 0.0;
    double S[H];
    double m = 0.0, W = (double)N;

    for (int t = 0; t <= T; ++t) {
        // ---- (A) block max of the current log weights  (log_normalize, pf.py:374-377) ----
        REAL ml = -INFINITY;
#pragma unroll
        for (int k = 0; k < PPT; ++k)
            if (k * NT + tid < N) ml = lw[k] > ml ? lw[k] : ml;
        ml = wave_max(ml);
        if (lane == 0) red_max[wave] = (double)ml;
        __syncthreads();                                                        // barrier 1
        double mm = red_max[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) mm = red_max[w] > mm ? red_max[w] : mm;
        m = mm;
        // ---- (B) unnormalised weights, (C) prefix scan + weighted statistic sums --------
        const bool needS = needS_every || (t == T);
        double p[PPT], cs[PPT];
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const bool valid = (k * NT + tid) < N;
            p[k] = valid ? (double)exp_r((REAL)(lw[k] - (REAL)m)) : 0.0;
            cs[k] = wave_incl_scan(p[k], lane);
            if (lane == WAVE - 1) red_scan[k * NW + wave] = cs[k];
        }
        if (needS) {
#pragma unroll
            for (int h = 0; h < H; ++h) {
                double part = 0.0;
#pragma unroll
                for (int k = 0; k < PPT; ++k) part += (double)s[k][h] * p[k];
                part = wave_sum(part);
                if (lane == 0) red_S[h * NW + wave] = part;
            }
        }
        __syncthreads();                                                        // barrier 2
        {
            double run = 0.0;
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                double off = 0.0;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    if (w == wave) off = run;
                    run += red_scan[k * NW + w];
                }
                cs[k] += off;
            }
            W = run;
        }
#pragma unroll
        for (int h = 0; h < H; ++h) S[h] = 0.0;
        if (needS) {
#pragma unroll
            for (int h = 0; h < H; ++h) {
                double acc = 0.0;
#pragma unroll
                for (int w = 0; w < NW; ++w) acc += red_S[h * NW + w];
                S[h] = acc / W;
            }
        }
        // log-likelihood increment of the step that produced these weights
        // (buffered_smoother.py:124-126): log(mean(exp(logw))) = m + log(W/N)
        if (t > 0 && (t - 1) >= t1 && (t - 1) < tL) ll += wt_prev * (m + log(W / (double)N));
        if (P.trace_ll && tid == 0) P.trace_ll[t] = ll;
        if (is_filter && t > 0) {
#pragma unroll
            for (int h = 0; h < H; ++h) filt[h] += S[h];
        }
        if (t == T) break;

        // ---- (D) normalised CDF to LDS (RandomState.choice: cumsum, /= last) -------------
        const double y_t = P.y[t];
        const bool inside = (t >= t1) && (t < tL);
        const double wt = (inside && P.weights) ? P.weights[t - t1] : 1.0;
        double uu[PPT];
        REAL zz[PPT];
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = k * NT + tid;
            uu[k] = 0.0; zz[k] = (REAL)0;
            if (i < N) {
                if (RNG == PFG_RNG_REPLAY) {
                    uu[k] = P.u[(size_t)t * N + i];
                    zz[k] = (REAL)P.z[(size_t)t * N + i];
                } else {
                    u32x4 r = philox4x32_10({(uint32_t)i, (uint32_t)(t + 1), ctr_z, ctr_w}, k0, k1);
                    uu[k] = u01_53(r.x, r.y);
                    zz[k] = normal_bm<REAL>(r.z, r.w);
                }
                cdf[i] = cs[k] / W;
            }
        }
        __syncthreads();                                                        // barrier 3
        // ---- (E) multinomial ancestors: smallest j with cdf[j] > u  (searchsorted 'right')
        REAL xp[PPT][NS], sp[PPT][H];
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = k * NT + tid;
            int pos = 0;
            if (i < N) {
                const double u = uu[k];
                for (int step = np2 >> 1; step >= 1; step >>= 1) {
                    const int idx = pos + step - 1;
                    if (idx < N && cdf[idx] <= u) pos += step;
                }
                pos = pos < N - 1 ? pos : N - 1;
                if (RNG == PFG_RNG_REPLAY) {
                    // near-tie margin: how close u came to flipping the ancestor index
                    double hi = cdf[pos] - u;
                    double lo = pos > 0 ? u - cdf[pos - 1] : 1.0;
                    double mg = hi < lo ? hi : lo;
                    tie = mg < tie ? mg : tie;
                }
            }
            // ---- (F) gather parent state and statistics --------------------------------
#pragma unroll
            for (int d = 0; d < NS; ++d) xp[k][d] = xL[(size_t)d * NL + pos];
            if (!is_filter) {
#pragma unroll
                for (int h = 0; h < H; ++h) sp[k][h] = sL[(size_t)h * NL + pos];
            } else {
#pragma unroll
                for (int h = 0; h < H; ++h) sp[k][h] = (REAL)0;
            }
        }
        __syncthreads();                                                        // barrier 4
        // ---- (G) propose, weight, additive statistic; (H) publish to LDS -----------------
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = k * NT + tid;
            if (i < N) {
                REAL add[H];
                particle_step<MODEL, KERNEL, REAL>(c, xp[k], (REAL)y_t, zz[k], stat, inside, x[k], lw[k], add);
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    REAL a = add[h] * (REAL)wt;
                    // pf.py:175-179 / :78-80
                    s[k][h] = is_filter ? a : (lam * sp[k][h] + oml * (REAL)S[h]) + a;
                }
#pragma unroll
                for (int d = 0; d < NS; ++d) xL[(size_t)d * NL + i] = x[k][d];
#pragma unroll
                for (int h = 0; h < H; ++h) sL[(size_t)h * NL + i] = s[k][h];
            }
        }
        wt_prev = wt;
        trace_state(t + 1);
    }

    // ---- outputs --------------------------------------------------------------------
    if (RNG == PFG_RNG_REPLAY && P.out) {
        tie = -wave_max(-tie);
        if (lane == 0) red_max[wave] = tie;
        __syncthreads();
        tie = red_max[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) tie = red_max[w] < tie ? red_max[w] : tie;
    }
    if (tid == 0 && P.out) {
#pragma unroll
        for (int h = 0; h < PFG_MAX_STAT; ++h) P.out[h] = 0.0;
#pragma unroll
        for (int h = 0; h < H; ++h) P.out[h] = is_filter ? filt[h] : S[h];
        P.out[4] = ll;
        P.out[5] = W;
        P.out[6] = m;
        P.out[7] = tie;
    }
    if (P.final_x) {
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = k * NT + tid;
            if (i < N) {
#pragma unroll
                for (int d = 0; d < NS; ++d) P.final_x[(size_t)i * NS + d] = (double)x[k][d];
                if (P.final_logw) P.final_logw[i] = (double)lw[k];
                if (P.final_stats && !is_filter) {
#pragma unroll
                    for (int h = 0; h < H; ++h) P.final_stats[(size_t)i * H + h] = (double)s[k][h];
                }
            }
        }
    }
}

}  // namespace pfg
