// Device code of libpfgrad: one persistent workgroup runs one whole buffered particle-filter
// window (the T-loop of particle_filters/buffered_smoother.py:93-133) in a single launch.
// gfx950 only (wave64, DPP row_bcast, 160 KiB LDS).
//   pf_reg_kernel   N <= 1024: particles / statistics / CDF in LDS, log-weights in registers
//   pf_mem_kernel   N <= 16384: CDF in LDS, particle records in an L2-resident HBM scratch
//
// Compiled with -ffp-contract=off: the f64 instantiation follows the reference's NumPy
// expression order operation by operation, so REPLAY runs differ from the reference only by
// the rounding of exp / log (LDS-table forms, <= 2 ulp), of the shift used by log_normalize
// (f32-rounded maximum, mathematically immaterial) and of the parallel weight sum / prefix
// scan.  Where this file wants a fused multiply-add it says fma().
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "pfgrad.h"

namespace pfg {

constexpr int WAVE = 64;
// kernel instantiation modes beyond the plain filter / Nemeth path
constexpr int MODE_PLAIN = 0, MODE_PARIS = 1, MODE_SYSTEMATIC = 2, MODE_N2 = 3;
constexpr double LOG_2PI = 1.8378770664093453;   // log(2*pi)

// ------------------------------------------------------------------------------------
// wave-level primitives (64 lanes) on DPP: row_shr 1,2,4,8 inside 16-lane rows, then
// row_bcast:15 / row_bcast:31 across rows (gfx9 cross-lane modes; no LDS traffic).
// A lane whose DPP source does not exist keeps `old`, the operation's identity.
// ------------------------------------------------------------------------------------
// in-row shift with bound_ctrl: lanes without a source read 0 (no preset of the destination)
template <int CTRL>
__device__ __forceinline__ double dpp_shr0_f64(double v) {
    int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double old, double v) {
    int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f32(float old, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ double bcast_lane63(double v) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63),
                            __builtin_amdgcn_readlane(__double2loint(v), 63));
}
__device__ __forceinline__ float bcast_lane63(float v) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// inclusive prefix sum over the wave; lane 63 ends with the wave total
__device__ __forceinline__ double wave_incl_scan(double v) {
    v += dpp_shr0_f64<0x111>(v);        // row_shr:1
    v += dpp_shr0_f64<0x112>(v);        // row_shr:2
    v += dpp_shr0_f64<0x114>(v);        // row_shr:4
    v += dpp_shr0_f64<0x118>(v);        // row_shr:8
    v += dpp_f64<0x142, 0xa>(0.0, v);   // row_bcast:15 -> rows 1,3
    v += dpp_f64<0x143, 0xc>(0.0, v);   // row_bcast:31 -> rows 2,3
    return v;
}
__device__ __forceinline__ double wave_sum(double v) { return bcast_lane63(wave_incl_scan(v)); }

// a value every lane of the wave holds identically: pin it in scalar registers (2 SGPRs instead
// of 2 VGPRs for as long as it lives)
__device__ __forceinline__ double uniform_f64(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)),
                            __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

__device__ __forceinline__ double wave_max(double v) {
#define PFG_MAX_STEP(CTRL, RM) { double o = dpp_f64<CTRL, RM>(v, v); v = o > v ? o : v; }
    PFG_MAX_STEP(0x111, 0xf) PFG_MAX_STEP(0x112, 0xf) PFG_MAX_STEP(0x114, 0xf) PFG_MAX_STEP(0x118, 0xf)
    PFG_MAX_STEP(0x142, 0xa) PFG_MAX_STEP(0x143, 0xc)
#undef PFG_MAX_STEP
    return bcast_lane63(v);
}
// The shift m used by log_normalize is reduced in f32 (v_max_f32 takes DPP operands directly:
// 6 instructions instead of ~50 for f64).  It only has to be within a few ulp(f32) of the true
// maximum: exp(lw - m) / sum and m + log(W/N) are invariant to it up to rounding.
__device__ __forceinline__ float wave_max(float v) {
#define PFG_MAX_STEP(CTRL, RM) { float o = dpp_f32<CTRL, RM>(v, v); v = o > v ? o : v; }
    PFG_MAX_STEP(0x111, 0xf) PFG_MAX_STEP(0x112, 0xf) PFG_MAX_STEP(0x114, 0xf) PFG_MAX_STEP(0x118, 0xf)
    PFG_MAX_STEP(0x142, 0xa) PFG_MAX_STEP(0x143, 0xc)
#undef PFG_MAX_STEP
    return bcast_lane63(v);
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

// ------------------------------------------------------------------------------------
// Device RNG (PFG_RNG_DEVICE): one xoshiro128++ generator per lane (Blackman & Vigna 2019;
// adds / xors / rotates only -- 32-bit multiplies are quarter-rate on CDNA), its 128-bit state
// keyed by Philox4x32-10(seed; lane, stream = global chain id, step counter), so streams are
// reproducible and independent of how chains are spread over GPUs.
// ------------------------------------------------------------------------------------
struct LaneRng {
    uint32_t s0, s1, s2, s3;
    __device__ __forceinline__ uint32_t next() {
        const uint32_t sum = s0 + s3;
        const uint32_t result = ((sum << 7) | (sum >> 25)) + s0;
        const uint32_t t = s1 << 9;
        s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3;
        s2 ^= t;
        s3 = (s3 << 11) | (s3 >> 21);
        return result;
    }
};

__device__ __forceinline__ LaneRng lane_rng_init(uint64_t seed, uint64_t stream, uint64_t step, uint32_t lane) {
    u32x4 r = philox4x32_10({lane, (uint32_t)step, (uint32_t)stream,
                             (uint32_t)(stream >> 32) ^ (uint32_t)(step >> 32)},
                            (uint32_t)seed, (uint32_t)(seed >> 32));
    LaneRng g;
    g.s0 = r.x; g.s1 = r.y; g.s2 = r.z; g.s3 = r.w | 1u;   // never the all-zero state
    return g;
}

// uniform in (0,1) with 32 random bits (resampling needs resolution << 1/N only)
__device__ __forceinline__ double u01_32(uint32_t a) { return ((double)a + 0.5) * (1.0 / 4294967296.0); }

// ------------------------------------------------------------------------------------
// fp64 elementary functions on small LDS tables.  ocml's exp / log / sincospi cost 42 / 98 / 70
// VALU instructions each (half of them re-materialising polynomial coefficients); the table
// forms below need 17 / 20 / 17 and one LDS read, at <= 2 ulp -- well inside the parity
// tolerance.  Tables are filled once per workgroup with ocml.  Explicit fma(): the file is
// compiled with -ffp-contract=off.
//   e2[j] = 2^(j/128)                                   j < 128
//   lg[j] = {1/c_j, log c_j},  c_j = 1 + (j+0.5)/128    j < 128
//   sc[j] = {sin, cos}(2 pi (j+0.5)/256)                j < 256
// ------------------------------------------------------------------------------------
constexpr int TAB_E2 = 128, TAB_LG = 128, TAB_SC = 0;     // no sin/cos table: see Math<double,true>::normal_pair
constexpr int TAB_DOUBLES_EXP = TAB_E2 + 2 * TAB_LG, TAB_DOUBLES_RNG = 2 * TAB_SC;   // exp+log always; sincos with the device RNG

struct TabF64 {
    const double *e2;
    const double2 *lg;
    const double2 *sc;
};

__device__ inline void tab_fill(double *mem, bool with_rng, int tid, int nthreads) {
    double *lg = mem + TAB_E2, *sc = lg + 2 * TAB_LG;
    for (int j = tid; j < TAB_E2; j += nthreads) mem[j] = exp2((double)j * (1.0 / 128.0));
    for (int j = tid; j < TAB_LG; j += nthreads) {
        const double c = 1.0 + ((double)j + 0.5) * (1.0 / 128.0);
        lg[2 * j] = 1.0 / c; lg[2 * j + 1] = log(c);
    }
    if (with_rng) {
        for (int j = tid; j < TAB_SC; j += nthreads) {
            double sn, cs;
            sincospi(((double)j + 0.5) * (1.0 / 128.0), &sn, &cs);
            sc[2 * j] = sn; sc[2 * j + 1] = cs;
        }
    }
}

// exp(x), any x (overflow -> inf, underflow -> 0, -inf -> 0)
__device__ __forceinline__ double exp_tab(double x, const double *__restrict__ e2) {
    x = fmax(x, -1000.0);
    const double kd = rint(x * 184.6649652337873);                  // 128/ln2
    const int k = (int)kd;
#ifdef PFG_FAST_ALGEBRA
    // device-generator units: one-step reduction and a cubic for expm1 -- relative error < 3e-12
    // (|r| <= ln2/256: r^4/24 = 2e-12), far below the Monte-Carlo noise these kernels carry, and
    // 4 instructions shorter; the REPLAY units keep the <= 2 ulp form below
    const double r = fma(kd, -0.0054152123481245725, x);             // ln2/128
    const double t = e2[k & (TAB_E2 - 1)];
    double p = fma(r, 0.16666666666666666, 0.5);
    p = fma(p, r, 1.0);
    p = p * r;
    return ldexp(fma(t, p, t), k >> 7);
#else
    double r = fma(kd, -0.00541521234663378, x);                     // ln2/128, 32-bit head
    r = fma(kd, -1.4907929134926466e-12, r);                         //          tail
    const double t = e2[k & (TAB_E2 - 1)];
    double p = fma(r, 0.008333333333333333, 0.041666666666666664);   // expm1(r), |r| <= ln2/256
    p = fma(p, r, 0.16666666666666666);
    p = fma(p, r, 0.5);
    p = p * r;
    p = fma(p, r, r);
    return ldexp(fma(t, p, t), k >> 7);
#endif
}

// log(x) for finite x > 0 in the normal range
__device__ __forceinline__ double log_tab(double x, const double2 *__restrict__ lg) {
    const uint32_t hi = (uint32_t)__double2hiint(x);
    const int e = (int)(hi >> 20) - 1023;
    const double mant = __hiloint2double((int)((hi & 0x000FFFFFu) | 0x3FF00000u), __double2loint(x));
    const double2 t = lg[(hi >> 13) & (TAB_LG - 1)];
    const double r = fma(mant, t.x, -1.0);                           // |r| <= 2^-8
    double p = fma(r, 0.2, -0.25);                                   // log1p(r)
    p = fma(p, r, 0.3333333333333333);
    p = fma(p, r, -0.5);
    p = p * r;
    p = fma(p, r, r);
    return fma((double)e, 0.6931471805599453, t.y) + p;
}

// sqrt(x) for finite x > 0 (no special cases): rsq + one coupled Newton step + correction
__device__ __forceinline__ double sqrt_pos(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    const double d = fma(-g, g, x);
    return fma(d, h, g);
}

template <typename REAL, bool TAB> struct Math;
template <> struct Math<double, true> {
    TabF64 t;
    __device__ __forceinline__ double exp(double x) const { return exp_tab(x, t.e2); }
    __device__ __forceinline__ double log(double x) const { return log_tab(x, t.lg); }
    __device__ __forceinline__ double sqrt(double x) const { return ::sqrt(x); }
    // two independent standard normals from two words (Box-Muller, both branches).  The draws
    // are INPUTS of the filter, like the 32-bit uniforms: they are generated with the f32
    // transcendental units (v_log / v_sin / v_cos: ~12 issue slots per normal instead of ~25 for
    // a table-based fp64 evaluation) and widened; all arithmetic on the state stays fp64.
    // u1 keeps its full exponent range ((a + 0.5) 2^-32: |z| up to 6.7), the angle has 24 bits.
    __device__ __forceinline__ void normal_pair(uint32_t a, uint32_t b, double &z0, double &z1) const {
        const float u1 = ((float)a + 0.5f) * 2.3283064365386963e-10f;       // (0, 1]
        const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);             // [0,1): angle / 2pi
        const float r = sqrtf(-2.0f * __logf(u1));
        z0 = (double)(r * __builtin_amdgcn_cosf(u2));
        z1 = (double)(r * __builtin_amdgcn_sinf(u2));
    }
};
template <> struct Math<double, false> {
    TabF64 t;
    __device__ __forceinline__ double exp(double x) const { return ::exp(x); }
    __device__ __forceinline__ double log(double x) const { return ::log(x); }
    __device__ __forceinline__ double sqrt(double x) const { return ::sqrt(x); }
    __device__ __forceinline__ void normal_pair(uint32_t a, uint32_t b, double &z0, double &z1) const {
        const double u1 = ((double)a + 0.5) * (1.0 / 4294967296.0);
        const double r = ::sqrt(-2.0 * ::log(u1));
        double sn, cs;
        sincospi((double)b * (1.0 / 2147483648.0), &sn, &cs);
        z0 = r * cs; z1 = r * sn;
    }
};
template <bool TAB> struct Math<float, TAB> {
    TabF64 t;
    __device__ __forceinline__ float exp(float x) const { return __expf(x); }
    __device__ __forceinline__ float log(float x) const { return __logf(x); }
    __device__ __forceinline__ float sqrt(float x) const { return sqrtf(x); }
    __device__ __forceinline__ void normal_pair(uint32_t a, uint32_t b, float &z0, float &z1) const {
        const float u1 = ((float)(a >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0,1), 24 bits
        const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);           // [0,1): angle / 2pi
        const float r = sqrtf(-2.0f * __logf(u1));
        // v_sin_f32 / v_cos_f32 take their argument in revolutions
        z0 = r * __builtin_amdgcn_cosf(u2); z1 = r * __builtin_amdgcn_sinf(u2);
    }
};

// bytes of LDS math tables a kernel instantiation carries
template <typename REAL, int RNG, bool TAB>
__host__ __device__ constexpr size_t tab_bytes() {
    return (TAB && sizeof(REAL) == 8) ? (size_t)8 * (TAB_DOUBLES_EXP + (RNG == PFG_RNG_DEVICE ? TAB_DOUBLES_RNG : 0)) : 0;
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
    REAL logalpha, logLQinv;                            // PaRIS backward kernel
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
        d.logLQinv = log(d.LQinv);
    } else {
        d.logalpha = log(d.alpha);
    }
    if (MODEL == PFG_MODEL_LGSSM) {
        d.opt_prec = d.Qinv + (d.C * d.C) * d.Rinv;
        d.opt_sd = pow(d.opt_prec, -0.5);
        d.opt_var = 1.0 / d.Qinv + 1.0 / d.Rinv;
        d.opt_logvar = log(d.opt_var);
    }
    // wave-uniform by construction: pin every constant in scalar registers (frees ~2 VGPRs each)
    {
        double *f = reinterpret_cast<double *>(&d);
#pragma unroll
        for (int q = 0; q < (int)(sizeof(d) / sizeof(double)); ++q) f[q] = uniform_f64(f[q]);
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
    c.logalpha = (REAL)d.logalpha; c.logLQinv = (REAL)d.logLQinv;
    return c;
}

// One particle: parent state xp -> proposal x' (Kernel.rv), log weight (Kernel.reweight) and
// additive statistic (STAT = PFG_STAT_SCORE: complete-data score; otherwise the sufficient
// statistics), all from the same registers, straight-line.  add[] is NOT yet scaled by weight_t.
template <int MODEL, int KERNEL, int STAT, typename REAL, typename MATH>
__device__ __forceinline__ void particle_step(const Consts<REAL> &c, const MATH &mth, const REAL *xp,
                                              REAL y, REAL z, REAL *xn, REAL &lw, REAL *add) {
    constexpr int H = ModelDims<MODEL>::H;
    const REAL half = (REAL)0.5;
#pragma unroll
    for (int h = 0; h < H; ++h) add[h] = (REAL)0;
    if (MODEL == PFG_MODEL_SVM) {
        // svm/kernels.py:34-37, :56-62; svm/helper.py:342-348
        REAL xpA = xp[0] * c.A;
        REAL x1 = c.iLQinv * z + xpA;
        REAL e = mth.exp(-x1);
        REAL y2 = y * y;
#ifdef PFG_FAST_ALGEBRA
        // device-generator units (no operation-order parity to keep): the same expressions with
        // the wave-uniform factors of the step collected (they are computed once per step)
        const REAL k0 = c.c0 + c.logLRinv, ke = (-half * y2) * c.Rinv;
        lw = fma(ke, e, fma(-half, x1, k0));
#else
        lw = ((c.c0 + ((-half * y2) * e) * c.Rinv) + c.logLRinv) + (-half * x1);
#endif
        xn[0] = x1;
        if (STAT == PFG_STAT_SCORE) {
            REAL dx = x1 - c.A * xp[0];
            add[2] = (c.Qinv * dx) * xp[0];
            add[1] = c.iLQinv - (dx * dx) * c.LQinv;
#ifdef PFG_FAST_ALGEBRA
            add[0] = fma(-(y2 * c.LRinv), e, c.iLRinv);
#else
            REAL dy2 = y2 * e;                       // y^2 / exp(x')
            add[0] = c.iLRinv - dy2 * c.LRinv;
#endif
        } else {
            add[0] = x1; add[1] = x1 * x1; add[2] = xp[0] * x1;
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
        if (STAT == PFG_STAT_SCORE) {
            // lgssm/helper.py:1270-1277, order [LRinv, LQinv, C, A]
            REAL dx = x1 - c.A * xp[0];
            add[3] = (c.Qinv * dx) * xp[0];
            add[1] = c.iLQinv - (dx * dx) * c.LQinv;
            REAL dy = y - c.C * x1;
            add[2] = (c.Rinv * dy) * x1;
            add[0] = c.iLRinv - (dy * dy) * c.LRinv;
        } else {
            add[0] = x1; add[1] = x1 * x1; add[2] = xp[0] * x1;
        }
    } else {
        // garch/kernels.py:60-68 / :146-156, reweight :83-88 / :172-178
        REAL xx = xp[0] * xp[0];
        REAL s2 = (c.alpha + c.beta * xx) + c.gamma * xp[1];
        REAL x1;
        if (KERNEL == PFG_KERNEL_PRIOR) {
            x1 = mth.sqrt(s2) * z;
            REAL diff = y - x1;
            lw = (c.c0 + (-half * (diff * diff)) * c.Rinv) + c.logLRinv;
        } else {
#ifdef PFG_FAST_ALGEBRA
            // device-generator units: 1/s2 is shared with the score below, the step's uniform
            // factors are collected (3 divisions per particle instead of 4)
            const REAL rs2 = (REAL)1 / s2;
            REAL var = (REAL)1 / (c.Rinv + rs2);
            x1 = fma(mth.sqrt(var), z, var * (y * c.Rinv));
            REAL v2 = s2 + c.R;
            lw = fma(-half * (y * y), (REAL)1 / v2, c.c0) + (-half * mth.log(v2));
#else
            REAL var = (REAL)1 / (c.Rinv + (REAL)1 / s2);
            REAL mean = var * (y * c.Rinv);
            x1 = mean + mth.sqrt(var) * z;
            REAL v2 = s2 + c.R;
            lw = (c.c0 + (-half * (y * y)) / v2) + (-half * mth.log(v2));
#endif
        }
        xn[0] = x1; xn[1] = s2;
        if (STAT == PFG_STAT_SCORE) {
            // garch/helper.py:350-370, order [LRinv, log_mu, logit_phi, logit_lambduh]
            REAL v = s2;
#ifdef PFG_FAST_ALGEBRA
            const REAL rv = (KERNEL == PFG_KERNEL_PRIOR) ? (REAL)1 / v : (REAL)1 / v;   // CSE'd with rs2 above
            const REAL gv = (-half * (v - x1 * x1)) * (rv * rv);
            const REAL omp = (REAL)1 - c.phi, oml_ = (REAL)1 - c.lam;
            add[1] = gv * (omp * c.mu);
            add[2] = (gv * fma(c.lam, xx, fma(oml_, xp[1], -c.mu))) * (omp * c.phi);
            add[3] = (gv * (xx - xp[1])) * ((c.phi * oml_) * c.lam);
            REAL dy = y - x1;
            add[0] = fma(-(dy * dy), c.LRinv, c.iLRinv);
#else
            REAL gv = (-half * (v - x1 * x1)) / (v * v);
            add[1] = (gv * ((REAL)1 - c.phi)) * c.mu;
            add[2] = ((gv * ((-c.mu + c.lam * xx) + ((REAL)1 - c.lam) * xp[1])) * ((REAL)1 - c.phi)) * c.phi;
            add[3] = (((gv * c.phi) * (xx - xp[1])) * ((REAL)1 - c.lam)) * c.lam;
            REAL dy = y - x1;
            add[0] = c.iLRinv - (dy * dy) * c.LRinv;
#endif
        } else {
            REAL x2 = x1 * x1;
            add[0] = x1; add[1] = x2; add[2] = x2 * x2;
        }
    }
}

// ------------------------------------------------------------------------------------
// LDS-resident kernel: N <= NT*PPT particles; particle i = k*NT + tid belongs to thread tid,
// slot k.  Only the log-weights live in registers across timesteps; particles and statistics
// live in LDS as struct-of-arrays over the particle axis (lane i <-> particle i: conflict-free).
//   LDS: cdf[NL] f64 | buf0 {x[NS][NL], stats[H][NL]} | buf1 (PP only) | reduction scratch
// PP = ping-pong state buffers: children are written to the other buffer, so no barrier is
// needed between gathering parents and publishing children (3 barriers per timestep, and a
// slot's parent state dies as soon as its child is computed).  PP = false keeps ONE buffer
// (larger N fits in 160 KiB) at the price of a 4th barrier and of holding all gathered
// parents in registers across it.
// ------------------------------------------------------------------------------------
// Additive statistic h(parent, child) alone (PaRIS evaluates it for rewired parents): the same
// expressions as in particle_step.  `aux` carries the child's sub-expression the proposal step
// already has (SVM: exp(-x')).
template <int MODEL, int STAT, typename REAL>
__device__ __forceinline__ void additive_stat(const Consts<REAL> &c, const REAL *xp, const REAL *xn, REAL y,
                                              REAL aux, REAL *add) {
    constexpr int H = ModelDims<MODEL>::H;
    const REAL half = (REAL)0.5;
#pragma unroll
    for (int h = 0; h < H; ++h) add[h] = (REAL)0;
    if (STAT != PFG_STAT_SCORE) {
        if (MODEL == PFG_MODEL_GARCH) { REAL x2 = xn[0] * xn[0]; add[0] = xn[0]; add[1] = x2; add[2] = x2 * x2; }
        else { add[0] = xn[0]; add[1] = xn[0] * xn[0]; add[2] = xp[0] * xn[0]; }
        return;
    }
    if (MODEL == PFG_MODEL_SVM) {
        REAL dx = xn[0] - c.A * xp[0];
        add[2] = (c.Qinv * dx) * xp[0];
        add[1] = c.iLQinv - (dx * dx) * c.LQinv;
        add[0] = c.iLRinv - ((y * y) * aux) * c.LRinv;
    } else if (MODEL == PFG_MODEL_LGSSM) {
        REAL dx = xn[0] - c.A * xp[0];
        add[3] = (c.Qinv * dx) * xp[0];
        add[1] = c.iLQinv - (dx * dx) * c.LQinv;
        REAL dy = y - c.C * xn[0];
        add[2] = (c.Rinv * dy) * xn[0];
        add[0] = c.iLRinv - (dy * dy) * c.LRinv;
    } else {
        REAL xx = xp[0] * xp[0];
        REAL v = xn[1];
        REAL gv = (-half * (v - xn[0] * xn[0])) / (v * v);
        add[1] = (gv * ((REAL)1 - c.phi)) * c.mu;
        add[2] = ((gv * ((-c.mu + c.lam * xx) + ((REAL)1 - c.lam) * xp[1])) * ((REAL)1 - c.phi)) * c.phi;
        add[3] = (((gv * c.phi) * (xx - xp[1])) * ((REAL)1 - c.lam)) * c.lam;
        REAL dy = y - xn[0];
        add[0] = c.iLRinv - (dy * dy) * c.LRinv;
    }
}

// log q(child | parent) - max q: the accept-reject exponent of PaRIS backward sampling
// (Kernel.prior_log_density - get_prior_log_density_max; kernels.py:102-138, garch/kernels.py:20-47)
template <int MODEL, typename REAL, typename MATH>
__device__ __forceinline__ REAL backward_log_ratio(const Consts<REAL> &c, const MATH &mth, const REAL *xp,
                                                   const REAL *xn) {
    const REAL half = (REAL)0.5;
    if (MODEL == PFG_MODEL_GARCH) {
        REAL s2 = (c.alpha + c.beta * (xp[0] * xp[0])) + c.gamma * xp[1];
        REAL ll = ((-half * (xn[0] * xn[0])) / s2 - half * (REAL)LOG_2PI) - half * mth.log(s2);
        return ll - (-half * (REAL)LOG_2PI - half * c.logalpha);
    }
    REAL diff = xn[0] - c.A * xp[0];
    REAL ll = ((-half * (diff * diff)) * c.Qinv + -half * (REAL)LOG_2PI) + c.logLQinv;
    return ll - (-half * (REAL)LOG_2PI + c.logLQinv);
}

__host__ __device__ __forceinline__ constexpr int cdf_phys(int i) { return i + (i >> 5); }
// FAST layout = LDS math tables + sentinel-padded, bank-conflict-free cdf with an unrolled search.
// Everything except the 1024-thread single-buffer variant (which spends all LDS on particles).
__host__ __device__ constexpr bool fast_layout(int NT, bool PP) { return PP || NT <= 256; }

template <int NT, int PPT> struct RegLayout {
    static constexpr int NW = NT / WAVE;
    static constexpr int RED = PPT * NW + NW + PFG_MAX_STAT * NW + 8;  // doubles of scratch
};

// PP variants: cdf has NT*PPT entries (tail = sentinel 2.0 -> unrolled, clamp-free search) and
// the fp64 math runs on LDS tables; the single-buffer variant spends its LDS on particles.
template <int MODEL, typename REAL, int NT, int PPT, int RNG, bool PP, int MODE = 0>
__host__ __device__ inline size_t reg_kernel_lds_bytes(int N) {
    constexpr bool PARIS = (MODE == MODE_PARIS || MODE == MODE_N2);   // parents' log-weights in LDS
    constexpr bool FAST = fast_layout(NT, PP);
    // FAST layouts hold NT*PPT particle slots whatever N is: the array stride is a compile-time
    // constant and folds into the ds_read / ds_write immediates
    size_t NL = FAST ? (size_t)NT * PPT : (size_t)(N + WAVE - 1) / WAVE * WAVE;
    size_t NC = FAST ? (size_t)NT * PPT + (size_t)NT * PPT / 32 : NL;   // padded 33/32 (see cdf_phys)
    return NC * 8 + (PP ? 2 : 1) * NL * (ModelDims<MODEL>::NS + ModelDims<MODEL>::H) * sizeof(REAL) +
           (size_t)RegLayout<NT, PPT>::RED * 8 + tab_bytes<REAL, RNG, FAST>() +
           (PARIS ? NL * 8 + NL * 4 + 3 * NL * 4 : 0);   // PaRIS: parents' log-weights, fallback queue,
                                                         // two wave-queue arrays, accepted parents
}

// waves per SIMD the register allocator should aim for: what LDS lets a CU hold anyway.
// 256x4 fp64: ping-pong state is 80 KB/workgroup -> 2 workgroups (2 waves/SIMD); the single
// buffer is 50 KB -> 3, which is worth a few spilled registers (measured +15 %).
__host__ __device__ constexpr int occ_max(int NT, int PPT, size_t real, bool PP) {
    return (NT >= 512 || PPT == 1) ? 4 : ((PP && real == 8) ? 2 : 3);
}
__host__ __device__ constexpr int occ_min(int NT, int PPT, size_t real, bool PP) {
    return (NT == 256 && PPT == 4 && !PP) ? 3 : 1;
}

template <int MODEL, int KERNEL, typename REAL, int NT, int PPT, int RNG, bool PP, int MODE = 0>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(occ_min(NT, PPT, sizeof(REAL), PP), occ_max(NT, PPT, sizeof(REAL), PP)))) void pf_reg_kernel(const pfg_dev_problem *__restrict__ probs) {
    constexpr bool PARIS = (MODE == MODE_PARIS);
    constexpr bool systematic = (MODE == MODE_SYSTEMATIC);
    constexpr bool N2 = (MODE == MODE_N2);
    static_assert(!(PARIS || N2) || PP, "PaRIS / O(N^2) need the parents intact while children are built: ping-pong buffers");
    static_assert(!systematic || RNG == PFG_RNG_DEVICE, "systematic resampling draws its offset on the device");
    constexpr int NS = ModelDims<MODEL>::NS;
    constexpr int H = ModelDims<MODEL>::H;
    constexpr int NW = NT / WAVE;
    extern __shared__ __align__(16) unsigned char smem[];

    const pfg_dev_problem &P = probs[blockIdx.x];
    const int N = P.N, T = P.T, t1 = P.t1, tL = P.tL;
    const int NL = fast_layout(NT, PP) ? NT * PPT : (N + WAVE - 1) / WAVE * WAVE;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
    const bool is_filter = (P.smoother == PFG_SMOOTHER_FILTER);
    const int stat = P.stat;
    const double lam_d = is_filter ? 0.0 : ((P.smoother == PFG_SMOOTHER_PARIS || N2) ? 1.0 : P.lambduh);
    const REAL lam = (REAL)lam_d, oml = (REAL)(1.0 - lam_d);
    const bool needS_every = is_filter || (lam_d != 1.0);
    const double *__restrict__ const yv = P.y;
    const double *__restrict__ const wv = P.weights;
    const double *__restrict__ const uv = P.u;
    const double *__restrict__ const zv = P.z;

    constexpr bool FAST = fast_layout(NT, PP);
    constexpr bool TAB = FAST;
    // Device RNG only: the CDF is built in THREAD-major order (position tid*PPT + k <-> particle
    // k*NT + tid).  Multinomial resampling does not care how particles are labelled, and in this
    // order a thread's PPT weights are contiguous: one in-register prefix + ONE wave scan per
    // thread instead of PPT wave scans.  REPLAY keeps the reference's index order (parity).
    constexpr bool BLK = FAST && RNG == PFG_RNG_DEVICE && MODE == MODE_PLAIN && (PPT & (PPT - 1)) == 0;
    constexpr int LOG_PPT = PPT == 1 ? 0 : (PPT == 2 ? 1 : (PPT == 4 ? 2 : 3));
    // PP: the cdf is stored at physical index i + (i >> 5) (one pad slot per 32 entries): the
    // binary search probes at power-of-two strides, which would otherwise all hit one LDS bank
    // (measured: 720 conflict cycles per wave-timestep, i.e. all of SQ_LDS_BANK_CONFLICT).
    const int NC = FAST ? NT * PPT + NT * PPT / 32 : NL;
    double *cdf = reinterpret_cast<double *>(smem);
    REAL *buf0 = reinterpret_cast<REAL *>(cdf + NC);
    const size_t bufsz = (size_t)(NS + H) * NL;
    REAL *cur = buf0, *nxt = PP ? buf0 + bufsz : buf0;
    double *red = reinterpret_cast<double *>(buf0 + (PP ? 2 : 1) * bufsz);
    double *red_scan = red;                 // [PPT*NW]
    double *red_max = red + PPT * NW;       // [NW]
    float *red_maxf = reinterpret_cast<float *>(red_max);
    double *red_S = red_max + NW;           // [H*NW]
    double *red_W0 = red_S + PFG_MAX_STAT * NW;      // [8] spare doubles (systematic-resampling offset)
    const double invN = 1.0 / (double)N;
    double *tabmem = red + RegLayout<NT, PPT>::RED;
    REAL *lwL = reinterpret_cast<REAL *>(tabmem + tab_bytes<REAL, RNG, TAB>() / 8);    // [NL], PARIS only
    int *paris_queue = reinterpret_cast<int *>(tabmem + tab_bytes<REAL, RNG, TAB>() / 8 + NL);   // [NL], PARIS only

    Math<REAL, TAB> mth;
    mth.t.e2 = tabmem;
    mth.t.lg = reinterpret_cast<const double2 *>(tabmem + TAB_E2);
    mth.t.sc = reinterpret_cast<const double2 *>(tabmem + TAB_E2 + 2 * TAB_LG);
    if (tab_bytes<REAL, RNG, TAB>() > 0) tab_fill(tabmem, RNG == PFG_RNG_DEVICE, tid, NT);
    if (FAST) {
#pragma unroll
        for (int k = 0; k < PPT; ++k)
            if (k * NT + tid >= N) cdf[cdf_phys(k * NT + tid)] = 2.0;      // sentinel: never <= u
    }
    __syncthreads();

    const Consts<REAL> c = make_consts<MODEL, REAL>(P.theta);
    int np2 = 1;
    while (np2 < N) np2 <<= 1;

    LaneRng rng = {};
    if (RNG == PFG_RNG_DEVICE)
        rng = lane_rng_init(P.seed, P.stream, P.step_ctr ? *P.step_ctr : 0ull, (uint32_t)tid);
    // PPT standard normals for this thread's slots (device RNG)
    auto draw_normals = [&](REAL *zz) {
#pragma unroll
        for (int k = 0; k < PPT; k += 2) {
            REAL a, b;
            mth.normal_pair(rng.next(), rng.next(), a, b);
            zz[k] = a;
            if (k + 1 < PPT) zz[k + 1] = b;
        }
    };

    REAL lw[PPT];
    // ---- x0 (kernels.py:83-100, garch/kernels.py:7-18) or warm start ------------------
    {
        double pv = P.prior_var;
        if (MODEL == PFG_MODEL_GARCH && (P.flags & PFG_FLAG_GARCH_STATIONARY_PRIOR))
            pv = (double)c.alpha / (1.0 - (double)c.beta - (double)c.gamma);
        const double sd = sqrt(pv);
        REAL z0[PPT];
        if (RNG == PFG_RNG_DEVICE) draw_normals(z0);
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = k * NT + tid;
            lw[k] = (REAL)0;
            if (i < N) {
                REAL x[NS], s[H];
#pragma unroll
                for (int d = 0; d < NS; ++d) x[d] = (REAL)0;
#pragma unroll
                for (int h = 0; h < H; ++h) s[h] = (REAL)0;
                if (P.init_x) {
#pragma unroll
                    for (int d = 0; d < NS; ++d) x[d] = (REAL)P.init_x[(size_t)i * NS + d];
                    lw[k] = (REAL)P.init_logw[i];
                    if (P.init_stats && !is_filter) {
#pragma unroll
                        for (int h = 0; h < H; ++h) s[h] = (REAL)P.init_stats[(size_t)i * H + h];
                    }
                } else {
                    const double z = (RNG == PFG_RNG_REPLAY) ? P.z0[i] : (double)z0[k];
                    x[0] = (REAL)(P.prior_mean + sd * z);
                }
#pragma unroll
                for (int d = 0; d < NS; ++d) cur[(size_t)d * NL + i] = x[d];
#pragma unroll
                for (int h = 0; h < H; ++h) cur[(size_t)(NS + h) * NL + i] = s[h];
                if (P.trace_x) {
#pragma unroll
                    for (int d = 0; d < NS; ++d) P.trace_x[(size_t)i * NS + d] = (double)x[d];
                    P.trace_logw[i] = (double)lw[k];
                    if (P.trace_stats && !is_filter) {
#pragma unroll
                        for (int h = 0; h < H; ++h) P.trace_stats[(size_t)i * H + h] = (double)s[h];
                    }
                }
            }
        }
    }

    double ll = 0.0, wt_prev = 1.0, tie = 1.0;
    double filt[H], S[H];
#pragma unroll
    for (int h = 0; h < H; ++h) { filt[h] = 0.0; S[h] = 0.0; }
    double m = 0.0, W = (double)N;
    // slots beyond N: log-weight -inf (weight exactly 0, never an ancestor); their lanes run the
    // same straight-line code on clamped indices and only their stores are masked.
    bool valid[PPT];
    int own[PPT];                       // own particle index, clamped for reads
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        valid[k] = (k * NT + tid) < N;
        own[k] = valid[k] ? (k * NT + tid) : (N - 1);
        if (!valid[k]) lw[k] = -INFINITY;
    }
    const int last = N - 1;

    for (int t = 0; t <= T; ++t) {
        // ---- (A) block max of the current log weights  (log_normalize, pf.py:374-377) ----
        float ml = (float)lw[0];
#pragma unroll
        for (int k = 1; k < PPT; ++k) ml = fmaxf(ml, (float)lw[k]);
        ml = wave_max(ml);
        if (lane == 0) red_maxf[wave] = ml;
        __syncthreads();                                                        // barrier 1
        {
            float mm = red_maxf[0];
#pragma unroll
            for (int w = 1; w < NW; ++w) mm = fmaxf(mm, red_maxf[w]);
            m = uniform_f64((double)mm);      // f32-rounded max: a valid shift for log_normalize (see wave_max)
        }
        // ---- (B) unnormalised weights, (C) prefix scan + weighted statistic sums --------
        const bool needS = needS_every || (t == T);
        double cs[PPT];
#pragma unroll
        for (int k = 0; k < PPT; ++k) cs[k] = (double)mth.exp((REAL)(lw[k] - (REAL)m));   // exp(-inf) = 0
        if (needS) {
#pragma unroll
            for (int h = 0; h < H; ++h) {
                double part = 0.0;
#pragma unroll
                for (int k = 0; k < PPT; ++k) part += (double)cur[(size_t)(NS + h) * NL + own[k]] * cs[k];
                part = wave_sum(part);
                if (lane == 0) red_S[h * NW + wave] = part;
            }
        }
        if (BLK) {
#pragma unroll
            for (int k = 1; k < PPT; ++k) cs[k] += cs[k - 1];
            const double inc = wave_incl_scan(cs[PPT - 1]);
            const double exc = inc - cs[PPT - 1];
#pragma unroll
            for (int k = 0; k < PPT; ++k) cs[k] += exc;
            if (lane == WAVE - 1) red_scan[wave] = inc;
        } else {
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                cs[k] = wave_incl_scan(cs[k]);
                if (lane == WAVE - 1) red_scan[k * NW + wave] = cs[k];
            }
        }
        if (RNG != PFG_RNG_REPLAY && systematic && tid == 0) red_W0[0] = u01_32(rng.next());
        // this step's randomness: REPLAY loads are issued here so that their latency overlaps the
        // barrier; device draws happen right before their use (keeps register pressure down)
        double uu[PPT];
        REAL zz[PPT];
        if (t < T && RNG == PFG_RNG_REPLAY) {
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                uu[k] = uv[(size_t)t * N + own[k]];
                zz[k] = (REAL)zv[(size_t)t * N + own[k]];
            }
        }
        __syncthreads();                                                        // barrier 2
        if (BLK) {
            // NW wave totals: exclusive prefix by a DPP scan over the first lanes
            const double tot = (lane < NW) ? red_scan[lane] : 0.0;
            double inc = tot;
            inc += dpp_shr0_f64<0x111>(inc);
            inc += dpp_shr0_f64<0x112>(inc);
            if (NW > 4) { inc += dpp_shr0_f64<0x114>(inc); inc += dpp_shr0_f64<0x118>(inc); }
            const double exc = inc - tot;
            const double off = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(exc), wave),
                                                __builtin_amdgcn_readlane(__double2loint(exc), wave));
#pragma unroll
            for (int k = 0; k < PPT; ++k) cs[k] += off;
            W = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(inc), NW - 1),
                                 __builtin_amdgcn_readlane(__double2loint(inc), NW - 1));
        } else if (PPT * NW <= 16) {
            // lane j < PPT*NW holds total j; exclusive prefix by a 16-lane DPP scan; each thread
            // picks its PPT offsets and the grand total with v_readlane (uniform indices)
            double tot = (lane < PPT * NW) ? red_scan[lane] : 0.0;
            double inc = tot;
            inc += dpp_shr0_f64<0x111>(inc);
            inc += dpp_shr0_f64<0x112>(inc);
            inc += dpp_shr0_f64<0x114>(inc);
            inc += dpp_shr0_f64<0x118>(inc);
            const double exc = inc - tot;
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int j = k * NW + wave;
                cs[k] += __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(exc), j),
                                          __builtin_amdgcn_readlane(__double2loint(exc), j));
            }
            W = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(inc), PPT * NW - 1),
                                 __builtin_amdgcn_readlane(__double2loint(inc), PPT * NW - 1));
        } else {
            double run = 0.0;
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                double off = 0.0;
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    off = (w == wave) ? run : off;
                    run += red_scan[k * NW + w];
                }
                cs[k] += off;
            }
            W = uniform_f64(run);
        }
        const double invW = uniform_f64(1.0 / W);
        if (needS) {
#pragma unroll
            for (int h = 0; h < H; ++h) {
                double acc = 0.0;
#pragma unroll
                for (int w = 0; w < NW; ++w) acc += red_S[h * NW + w];
                S[h] = uniform_f64(acc * invW);
            }
        }
        // log-likelihood increment of the step that produced these weights
        // (buffered_smoother.py:124-126): log(mean(exp(logw))) = m + log(W/N).  Wave 0 only.
        if (wave == 0) {
            if (t > 0 && (t - 1) >= t1 && (t - 1) < tL) ll = uniform_f64(ll + wt_prev * (m + log(W / (double)N)));
            if (P.trace_ll && tid == 0) P.trace_ll[t] = ll;
        }
        if (is_filter && t > 0) {
#pragma unroll
            for (int h = 0; h < H; ++h) filt[h] = uniform_f64(filt[h] + S[h]);
        }
        if (t == T) break;

        // ---- (D) normalised CDF to LDS (RandomState.choice: cumsum, /= last) -------------
        const double y_t = yv[t];
        const bool inside = (t >= t1) && (t < tL);
        const double wt = (inside && wv) ? wv[t - t1] : 1.0;
        const bool use_stat = inside && (stat != PFG_STAT_NONE);
        const bool plain = !needS_every;                 // not filter and lambda == 1
        if (BLK) {
            // all PPT positions: slots beyond N carry weight 0 (flat CDF, never selected)
#pragma unroll
            for (int k = 0; k < PPT; ++k) cdf[cdf_phys(tid * PPT + k)] = cs[k] * invW;
        } else {
#pragma unroll
            for (int k = 0; k < PPT; ++k)
                if (valid[k]) {
                    cdf[FAST ? cdf_phys(k * NT + tid) : k * NT + tid] = cs[k] * invW;
                    if (PARIS || N2) lwL[k * NT + tid] = lw[k];
                }
        }
        __syncthreads();                                                        // barrier 3

        // ---- (E) ancestors: smallest j with cdf[j] > u (searchsorted 'right').  Branch-free:
        // probes past the end read cdf[N-1] (= 1 > u), the final clamp covers rounding.
        if (RNG != PFG_RNG_REPLAY) {
            if (systematic) {
                // extension: one uniform per timestep (drawn by thread 0 before barrier 2)
                const double u0 = red_W0[0];
#pragma unroll
                for (int k = 0; k < PPT; ++k) uu[k] = ((double)(k * NT + tid) + u0) * invN;
            } else {
#pragma unroll
                for (int k = 0; k < PPT; ++k) uu[k] = u01_32(rng.next());
            }
        }
        int anc[PPT];
#pragma unroll
        for (int k = 0; k < PPT; ++k) anc[k] = 0;
        if (FAST) {
            // sentinel-padded cdf, physical positions: log2(NT*PPT) fixed probes whose offsets
            // fold into the ds_read immediates; logical index recovered once at the end
#pragma unroll
            for (int step = (NT * PPT) >> 1; step >= 1; step >>= 1) {
                const int probe = step - 1 + (step >= 32 ? (step >> 5) - 1 : 0);
                const int adv = step + (step >> 5);
#pragma unroll
                for (int k = 0; k < PPT; ++k) anc[k] += (cdf[anc[k] + probe] <= uu[k]) ? adv : 0;
            }
#pragma unroll
            for (int k = 0; k < PPT; ++k) anc[k] -= (anc[k] * 993) >> 15;      // p - p/33 (exact for p < 8192)
            if (BLK) {
                // CDF position -> particle index
#pragma unroll
                for (int k = 0; k < PPT; ++k) anc[k] = (anc[k] & (PPT - 1)) * NT + (anc[k] >> LOG_PPT);
            }
        } else {
            for (int step = np2 >> 1; step >= 1; step >>= 1) {
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    int idx = anc[k] + step - 1;
                    idx = idx < last ? idx : last;
                    anc[k] += (cdf[idx] <= uu[k]) ? step : 0;
                }
            }
        }
#pragma unroll
        for (int k = 0; k < PPT; ++k) anc[k] = anc[k] < last ? anc[k] : last;
        if (RNG == PFG_RNG_REPLAY) {
            // near-tie margin: how close u came to flipping an ancestor index
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const double hi = cdf[FAST ? cdf_phys(anc[k]) : anc[k]] - uu[k];
                const double lo = anc[k] > 0 ? uu[k] - cdf[FAST ? cdf_phys(anc[k] - 1) : anc[k] - 1] : 1.0;
                const double mg = hi < lo ? hi : lo;
                tie = (valid[k] && mg < tie) ? mg : tie;
            }
        }
        // ---- (F) gather parents, (G) propose / weight / statistic, (H) publish children ---
        auto slots = [&](auto stat_tag) {
            constexpr int STAT = decltype(stat_tag)::value;
            REAL xp[PPT][NS], sp[PPT][H];
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
#pragma unroll
                for (int d = 0; d < NS; ++d) xp[k][d] = cur[(size_t)d * NL + anc[k]];
#pragma unroll
                for (int h = 0; h < H; ++h) sp[k][h] = cur[(size_t)(NS + h) * NL + anc[k]];
            }
            if (!PP) __syncthreads();                                           // barrier 4 (single buffer)
            if (RNG != PFG_RNG_REPLAY) draw_normals(zz);
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                REAL xn[NS], add[H], lwn;
                particle_step<MODEL, KERNEL, STAT, REAL>(c, mth, xp[k], (REAL)y_t, zz[k], xn, lwn, add);
                lw[k] = valid[k] ? lwn : (REAL)(-INFINITY);
                if (plain) {
                    // Poyiadjis O(N), lambda = 1: 1*s[a] + 0*S + w_t h = s[a] + w_t h exactly
                    if (use_stat) {
#pragma unroll
                        for (int h = 0; h < H; ++h) sp[k][h] = sp[k][h] + add[h] * (REAL)wt;
                    }
                } else {
#pragma unroll
                    for (int h = 0; h < H; ++h) {
                        const REAL a = use_stat ? add[h] * (REAL)wt : (REAL)0;
                        // pf.py:175-179 / :78-80
                        const REAL sm = (lam * sp[k][h] + oml * (REAL)S[h]) + a;
                        sp[k][h] = is_filter ? a : sm;
                    }
                }
                if (valid[k]) {
                    const int i = k * NT + tid;
#pragma unroll
                    for (int d = 0; d < NS; ++d) nxt[(size_t)d * NL + i] = xn[d];
#pragma unroll
                    for (int h = 0; h < H; ++h) nxt[(size_t)(NS + h) * NL + i] = sp[k][h];
                }
            }
        };
        // PaRIS (pf.py:183-341): children are proposed from the filter's ancestors as above, then
        // every child draws Ntilde parents from the backward kernel  w_k q(child | x_k)  by
        // accept-reject against the filter weights (exact categorical fallback after
        // max_accept_reject rounds) and averages  stats[J] + w_t h(x_J, child)  over them.
        auto paris_slots = [&](auto stat_tag) {
            constexpr int STAT = decltype(stat_tag)::value;
            const int Nt = P.Ntilde, R = P.max_accept_reject;
            const double *__restrict__ const pidx = P.paris_idx_u;
            const double *__restrict__ const pacc = P.paris_acc_u;
            const double *__restrict__ const pman = P.paris_man_u;
            int *queue = paris_queue;                               // [<= N] children left to the fallback
            int *qcount = reinterpret_cast<int *>(red_max) + NW;    // behind the NW floats of red_maxf
            // ---- 1. propose every child from its filter ancestor and publish x' -------------
            if (RNG != PFG_RNG_REPLAY) draw_normals(zz);
            REAL xn[PPT][NS], lwn[PPT], aux[PPT], sacc[PPT][H];
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                REAL xp[NS], add[H];
#pragma unroll
                for (int d = 0; d < NS; ++d) xp[d] = cur[(size_t)d * NL + anc[k]];
                particle_step<MODEL, KERNEL, STAT, REAL>(c, mth, xp, (REAL)y_t, zz[k], xn[k], lwn[k], add);
                aux[k] = (MODEL == PFG_MODEL_SVM) ? mth.exp(-xn[k][0]) : (REAL)0;
#pragma unroll
                for (int h = 0; h < H; ++h) sacc[k][h] = (REAL)0;
                if (valid[k]) {
#pragma unroll
                    for (int d = 0; d < NS; ++d) nxt[(size_t)d * NL + k * NT + tid] = xn[k][d];
                }
            }
            // wave-local work queues of pending children (this wave's NT*PPT/NW slots of two [NL]
            // arrays) and the accepted parent of every child of the current backward draw
            int *const wq0 = paris_queue + NL + wave * (PPT * WAVE);
            int *const wq1 = paris_queue + 2 * NL + wave * (PPT * WAVE);
            int *const Jres = paris_queue + 3 * NL;
            const unsigned long long ltmask = (1ull << lane) - 1ull;
            for (int j = 0; j < Nt; ++j) {
                // ---- 2. accept-reject against the filter weights, up to R rounds per child ------
                // Pending children sit compacted in a wave-local queue.  While more than half a
                // wave is pending each lane tries one candidate for one child per pass; below that
                // a child gets K = 2^k <= 64/pending CONSECUTIVE rounds in one pass (K lanes, the
                // first accepting round wins -- exactly the sequential outcome, also on replayed
                // pools), so the long tail of rounds costs a handful of passes.
                int *qa = wq0, *qb = wq1;
                int cnt = 0;
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    const unsigned long long mk = __ballot(valid[k]);
                    if (valid[k]) qa[cnt + __popcll(mk & ltmask)] = k * NT + tid;
                    cnt += __popcll(mk);
                }
                auto candidate = [&](int child, int round, bool act, int &Iout) {
                    double u1, u2;
                    if (RNG == PFG_RNG_REPLAY) {
                        const size_t at = (((size_t)t * Nt + j) * R + (act ? round : 0)) * N + child;
                        u1 = pidx[at]; u2 = pacc[at];
                    } else { u1 = u01_32(rng.next()); u2 = u01_32(rng.next()); }
                    int I = 0;
#pragma unroll
                    for (int step = (NT * PPT) >> 1; step >= 1; step >>= 1) {
                        const int probe = step - 1 + (step >= 32 ? (step >> 5) - 1 : 0);
                        I += (cdf[I + probe] <= u1) ? step + (step >> 5) : 0;
                    }
                    I -= (I * 993) >> 15;
                    I = I < last ? I : last;
                    REAL xI[NS], xc[NS];
#pragma unroll
                    for (int d = 0; d < NS; ++d) { xI[d] = cur[(size_t)d * NL + I]; xc[d] = nxt[(size_t)d * NL + child]; }
                    const double thr = (double)mth.exp(backward_log_ratio<MODEL, REAL>(c, mth, xI, xc));
                    Iout = I;
                    return act && u2 <= thr;
                };
                int r0 = 0;
                while (cnt > 0 && r0 < R) {                       // wave-uniform
                    __builtin_amdgcn_wave_barrier();
                    int ncnt = 0;
                    if (cnt > WAVE / 2) {
                        for (int e0 = 0; e0 < cnt; e0 += WAVE) {
                            const int e = e0 + lane;
                            const bool act = e < cnt;
                            const int child = qa[act ? e : 0];
                            int I;
                            const bool acc = candidate(child, r0, act, I);
                            if (acc) Jres[child] = I;
                            const bool rej = act && !acc;
                            const unsigned long long mk = __ballot(rej);
                            if (rej) qb[ncnt + __popcll(mk & ltmask)] = child;
                            ncnt += __popcll(mk);
                        }
                        r0 += 1;
                    } else {
                        int logK = 1;
                        while ((cnt << (logK + 1)) <= WAVE) ++logK;           // cnt * 2^logK <= 64
                        const int K = 1 << logK;
                        const int e = lane >> logK, o = lane & (K - 1);
                        const bool have = e < cnt;
                        const bool act = have && (r0 + o) < R;
                        const int child = qa[have ? e : 0];
                        int I;
                        const bool acc = candidate(child, r0 + o, act, I);
                        const unsigned long long am = __ballot(acc);
                        const unsigned long long segmask = (K >= 64) ? ~0ull : ((1ull << K) - 1ull);
                        const unsigned long long seg = (am >> (e << logK)) & segmask;
                        const int first = __ffsll((long long)seg) - 1;       // lowest accepting round
                        if (acc && o == first) Jres[child] = I;
                        const bool rej = have && o == 0 && seg == 0ull;
                        const unsigned long long mk = __ballot(rej);
                        if (rej) qb[__popcll(mk & ltmask)] = child;
                        ncnt = __popcll(mk);
                        r0 += K;
                    }
                    { int *tq = qa; qa = qb; qb = tq; }
                    cnt = ncnt;
                }
                // ---- 3. children still pending: exact categorical draw, one child per wave at a
                //         time over all parents -------------------------------------------------------
                if (tid == 0) *qcount = 0;
                __syncthreads();
                for (int e0 = 0; e0 < cnt; e0 += WAVE) {
                    const int e = e0 + lane;
                    if (e < cnt) {
                        const int i = qa[e];
                        queue[atomicAdd(qcount, 1)] = i;
                        // the child's fallback uniform rides in its (still unused) statistic slot
                        const double um = (RNG == PFG_RNG_REPLAY) ? pman[((size_t)t * Nt + j) * N + i]
                                                                  : u01_32(rng.next());
                        nxt[(size_t)NS * NL + i] = (REAL)um;
                    }
                }
                __syncthreads();
                const int nq = *qcount;
                // one pending child per WAVE at a time: lane handles parents lane, lane+64, ...
                // (wave-local max / total / ordered cumulative count: no workgroup barrier inside)
                constexpr int MAXC = NT * PPT / WAVE;
                const int nchunk = (N + WAVE - 1) / WAVE;
                for (int e = wave; e < nq; e += NW) {
                    const int ci = queue[e];
                    REAL xc[NS];
#pragma unroll
                    for (int d = 0; d < NS; ++d) xc[d] = nxt[(size_t)d * NL + ci];
                    const double um = (double)nxt[(size_t)NS * NL + ci];
                    if constexpr (RNG == PFG_RNG_DEVICE) {
                        // Device generator: any enumeration of the parents is a valid categorical
                        // sampler, so enumerate LANE-major (lane's parents lane, lane+64, ...): per-lane
                        // running sums, ONE wave scan over the lane totals, then the owning lane
                        // resolves its own <= MAXC entries -- no per-chunk wave reductions (they are
                        // dependent DPP chains with nothing to overlap: one wave per SIMD here).
                        // fp64 shifts by the block maximum m of the parents' log-weights (the
                        // backward ratio is <= 0, so every exponent is <= 0); f32 takes the exact max.
                        REAL mm = (REAL)m;
                        REAL lq[MAXC];
                        float mxf2 = -INFINITY;
#pragma unroll
                        for (int mI = 0; mI < MAXC; ++mI) {
                            const int q = mI * WAVE + lane;
                            const int qq = q < N ? q : last;
                            REAL xq[NS];
#pragma unroll
                            for (int d = 0; d < NS; ++d) xq[d] = cur[(size_t)d * NL + qq];
                            lq[mI] = (q < N && mI < nchunk) ? lwL[qq] + backward_log_ratio<MODEL, REAL>(c, mth, xq, xc)
                                                            : (REAL)(-INFINITY);
                            mxf2 = fmaxf(mxf2, (float)lq[mI]);
                        }
                        if (sizeof(REAL) == 4) mm = (REAL)wave_max(mxf2);
                        double evl[MAXC], tl = 0.0;
#pragma unroll
                        for (int mI = 0; mI < MAXC; ++mI) {
                            evl[mI] = (double)mth.exp((REAL)(lq[mI] - mm));       // exp(-inf) = 0
                            tl += evl[mI];
                        }
                        const double incl = wave_incl_scan(tl);
                        const double target = um * bcast_lane63(incl);
                        int Lsel = (int)wave_sum(incl <= target ? 1.0 : 0.0);
                        Lsel = __builtin_amdgcn_readfirstlane(Lsel < WAVE - 1 ? Lsel : WAVE - 1);
                        const double loc = target - (incl - tl);                  // this lane's local target
                        double run = 0.0;
                        int msel = 0;
                        bool found = false;
#pragma unroll
                        for (int mI = 0; mI < MAXC; ++mI) {
                            run += evl[mI];
                            const bool here = !found && run > loc;
                            msel = here ? mI : msel;
                            found = found || here;
                        }
                        int res = msel * WAVE + lane;
                        res = __builtin_amdgcn_readlane(res, Lsel);
                        if (lane == 0) Jres[ci] = res < last ? res : last;
                        continue;
                    }
                    REAL l[MAXC];
                    float mxf = -INFINITY;
#pragma unroll
                    for (int mI = 0; mI < MAXC; ++mI) {
                        const int q = mI * WAVE + lane;
                        const int qq = q < N ? q : last;
                        REAL xq[NS];
#pragma unroll
                        for (int d = 0; d < NS; ++d) xq[d] = cur[(size_t)d * NL + qq];
                        l[mI] = (q < N) ? lwL[qq] + backward_log_ratio<MODEL, REAL>(c, mth, xq, xc) : (REAL)(-INFINITY);
                        mxf = fmaxf(mxf, (float)l[mI]);
                        if (mI + 1 >= nchunk) break;
                    }
                    const REAL mm = (REAL)wave_max(mxf);
                    // chunk m = parents [64m, 64m+64): independent wave sums (pipelined), then the
                    // chunk holding the target is scanned once -- index order as np.random.choice
                    double ev[MAXC], csum[MAXC], tot = 0.0;
#pragma unroll
                    for (int mI = 0; mI < MAXC; ++mI) {
                        ev[mI] = (mI < nchunk) ? (double)mth.exp((REAL)(l[mI] - mm)) : 0.0;
                        csum[mI] = wave_sum(ev[mI]);
                        tot += csum[mI];
                    }
                    const double target = um * tot;
                    double before = 0.0, evsel = 0.0, run = 0.0;
                    int msel = nchunk - 1;
                    bool found = false;
#pragma unroll
                    for (int mI = 0; mI < MAXC; ++mI) {
                        const bool here = !found && mI < nchunk && (run + csum[mI] > target || mI == nchunk - 1);
                        if (here) { msel = mI; before = run; found = true; }
                        evsel = here ? ev[mI] : evsel;
                        run += csum[mI];
                    }
                    const double inc = wave_incl_scan(evsel) + before;
                    int cnt = ((msel * WAVE + lane) < N && inc <= target) ? 1 : 0;
                    cnt = msel * WAVE + (int)wave_sum((double)cnt);
                    if (lane == 0) Jres[ci] = cnt < last ? cnt : last;
                }
                __syncthreads();
                // ---- 4. rewired parent: stats[J] + w_t h(x_J, child) ----------------------------
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    const int Jk = valid[k] ? Jres[k * NT + tid] : 0;
                    REAL xJ[NS], aj[H];
#pragma unroll
                    for (int d = 0; d < NS; ++d) xJ[d] = cur[(size_t)d * NL + Jk];
                    additive_stat<MODEL, STAT, REAL>(c, xJ, xn[k], (REAL)y_t, aux[k], aj);
#pragma unroll
                    for (int h = 0; h < H; ++h) {
                        const REAL a = use_stat ? aj[h] * (REAL)wt : (REAL)0;
                        sacc[k][h] += cur[(size_t)(NS + h) * NL + Jk] + a;
                    }
                }
                __syncthreads();                // queue / statistic-slot scratch free for the next j
            }
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                lw[k] = valid[k] ? lwn[k] : (REAL)(-INFINITY);
                if (valid[k]) {
#pragma unroll
                    for (int h = 0; h < H; ++h) nxt[(size_t)(NS + h) * NL + k * NT + tid] = sacc[k][h] / (REAL)Nt;
                }
            }
        };
        // Poyiadjis O(N^2) (pf.py:84-136): children are proposed from the filter's ancestors, then
        // every child averages  stats_j + w_t h(x_j, child)  over ALL parents j with the backward
        // weights  log_normalize(logw_j + log q(child | x_j)).  Every lane walks the parents in the
        // same order (LDS broadcast reads); two passes: exact per-child maximum, then exp-sums.
        auto n2_slots = [&](auto stat_tag) {
            constexpr int STAT = decltype(stat_tag)::value;
            if (RNG != PFG_RNG_REPLAY) draw_normals(zz);
            REAL xn[PPT][NS], lwn[PPT], aux[PPT];
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                REAL xp[NS], add[H];
#pragma unroll
                for (int d = 0; d < NS; ++d) xp[d] = cur[(size_t)d * NL + anc[k]];
                particle_step<MODEL, KERNEL, STAT, REAL>(c, mth, xp, (REAL)y_t, zz[k], xn[k], lwn[k], add);
                aux[k] = (MODEL == PFG_MODEL_SVM) ? mth.exp(-xn[k][0]) : (REAL)0;
                if (valid[k]) {
#pragma unroll
                    for (int d = 0; d < NS; ++d) nxt[(size_t)d * NL + k * NT + tid] = xn[k][d];
                }
            }
            REAL mx[PPT];
#pragma unroll
            for (int k = 0; k < PPT; ++k) mx[k] = (REAL)(-INFINITY);
#pragma unroll 2
            for (int j = 0; j < N; ++j) {
                REAL xj[NS];
#pragma unroll
                for (int d = 0; d < NS; ++d) xj[d] = cur[(size_t)d * NL + j];
                const REAL lj = lwL[j];
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    const REAL v = lj + backward_log_ratio<MODEL, REAL>(c, mth, xj, xn[k]);
                    mx[k] = v > mx[k] ? v : mx[k];
                }
            }
            REAL den[PPT], num[PPT][H];
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                den[k] = (REAL)0;
#pragma unroll
                for (int h = 0; h < H; ++h) num[k][h] = (REAL)0;
            }
#pragma unroll 2
            for (int j = 0; j < N; ++j) {
                REAL xj[NS], sj[H];
#pragma unroll
                for (int d = 0; d < NS; ++d) xj[d] = cur[(size_t)d * NL + j];
#pragma unroll
                for (int h = 0; h < H; ++h) sj[h] = cur[(size_t)(NS + h) * NL + j];
                const REAL lj = lwL[j];
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    const REAL e = mth.exp((lj + backward_log_ratio<MODEL, REAL>(c, mth, xj, xn[k])) - mx[k]);
                    REAL aj[H];
                    additive_stat<MODEL, STAT, REAL>(c, xj, xn[k], (REAL)y_t, aux[k], aj);
                    den[k] += e;
#pragma unroll
                    for (int h = 0; h < H; ++h) {
                        const REAL a = use_stat ? aj[h] * (REAL)wt : (REAL)0;
                        num[k][h] += e * (sj[h] + a);
                    }
                }
            }
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                lw[k] = valid[k] ? lwn[k] : (REAL)(-INFINITY);
                if (valid[k]) {
#pragma unroll
                    for (int h = 0; h < H; ++h) nxt[(size_t)(NS + h) * NL + k * NT + tid] = num[k][h] / den[k];
                }
            }
        };
        bool did_paris = false;
        if constexpr (N2) {
            if (stat == PFG_STAT_SCORE) n2_slots(std::integral_constant<int, PFG_STAT_SCORE>{});
            else n2_slots(std::integral_constant<int, PFG_STAT_SUFF>{});
            did_paris = true;
        }
        if constexpr (MODE == MODE_PARIS) {
            if (P.smoother == PFG_SMOOTHER_PARIS) {
                if (stat == PFG_STAT_SCORE) paris_slots(std::integral_constant<int, PFG_STAT_SCORE>{});
                else paris_slots(std::integral_constant<int, PFG_STAT_SUFF>{});
                did_paris = true;
            }
        }
        if (!did_paris) {
            if (stat == PFG_STAT_SCORE) slots(std::integral_constant<int, PFG_STAT_SCORE>{});
            else slots(std::integral_constant<int, PFG_STAT_SUFF>{});
        }
        if (P.trace_x) {
            // own children back from LDS (written by this thread: no barrier needed)
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                if (valid[k]) {
                    const int i = k * NT + tid;
                    const size_t row = (size_t)(t + 1) * N + i;
                    if (P.trace_anc) P.trace_anc[(size_t)t * N + i] = anc[k];
#pragma unroll
                    for (int d = 0; d < NS; ++d) P.trace_x[row * NS + d] = (double)nxt[(size_t)d * NL + i];
                    P.trace_logw[row] = (double)lw[k];
                    if (P.trace_stats && !is_filter) {
#pragma unroll
                        for (int h = 0; h < H; ++h)
                            P.trace_stats[row * H + h] = (double)nxt[(size_t)(NS + h) * NL + i];
                    }
                }
            }
        }
        if (PP) { REAL *tmp = cur; cur = nxt; nxt = tmp; }
        wt_prev = wt;
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
        // own entries of the current buffer: written by this thread, no barrier needed
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = k * NT + tid;
            if (i < N) {
#pragma unroll
                for (int d = 0; d < NS; ++d) P.final_x[(size_t)i * NS + d] = (double)cur[(size_t)d * NL + i];
                if (P.final_logw) P.final_logw[i] = (double)lw[k];
                if (P.final_stats && !is_filter) {
#pragma unroll
                    for (int h = 0; h < H; ++h)
                        P.final_stats[(size_t)i * H + h] = (double)cur[(size_t)(NS + h) * NL + i];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// Large-N kernel: one 1024-thread workgroup per window, any N <= MEM_MAX_N.  Only the CDF
// (padded, sentinel-filled up to the next power of two) and the math tables live in LDS;
// particles, statistics and log-weights live in a per-window HBM scratch that stays
// L2-resident (N = 10000 fp64 SVM: 2 x 320 KB ping-pong + 80 KB), every access by the owning
// thread coalesced over the particle axis, only the parent gather random.  The timestep is
// the same phase sequence as pf_reg_kernel with rolled loops over chunks of 1024 particles.
//   scratch (REAL): lw[N] | buf0 [N][REC] | buf1 [N][REC],  record = {x[NS], stats[H], pad}
// (array-of-records: the parent gather is ONE 16-byte-vector access per particle instead of NS+H
// scattered 8-byte reads, each of which would pull its own cache line from L2)
// ------------------------------------------------------------------------------------
constexpr int MEM_NT = 1024;
constexpr int MEM_NW = MEM_NT / WAVE;
constexpr int MEM_MAX_N = 16384;
constexpr int MEM_MAX_CHUNKS = MEM_MAX_N / MEM_NT;

__host__ __device__ inline int mem_np2(int N) { int p = 64; while (p < N) p <<= 1; return p; }

// record length in REALs: NS + H rounded up to whole 16-byte vectors
template <int MODEL, typename REAL>
__host__ __device__ constexpr int mem_rec_len() {
    constexpr int per = 16 / (int)sizeof(REAL);
    return (ModelDims<MODEL>::NS + ModelDims<MODEL>::H + per - 1) / per * per;
}
// PaRIS adds: the children's log-weights (the parents' stay readable for the exact fallback),
// the fallback queue (child, result: int32 each) and its uniforms
template <int MODEL, typename REAL>
__host__ __device__ inline size_t mem_kernel_scratch_bytes(int N, bool paris = false) {
    return (size_t)N * sizeof(REAL) * (1 + 2 * mem_rec_len<MODEL, REAL>()) + 16 +
           (paris ? (size_t)N * (2 * sizeof(REAL) + 8) + 16 + 2 * (size_t)((N + MEM_NT - 1) / MEM_NT * MEM_NT) * 4 : 0);
}
template <int REC, typename REAL>
__device__ __forceinline__ void rec_load(REAL *dst, const REAL *src) {
    using V = float4;
#pragma unroll
    for (int v = 0; v < REC * (int)sizeof(REAL) / 16; ++v)
        reinterpret_cast<V *>(dst)[v] = reinterpret_cast<const V *>(src)[v];
}
template <int REC, typename REAL>
__device__ __forceinline__ void rec_store(REAL *dst, const REAL *src) {
    using V = float4;
#pragma unroll
    for (int v = 0; v < REC * (int)sizeof(REAL) / 16; ++v)
        reinterpret_cast<V *>(dst)[v] = reinterpret_cast<const V *>(src)[v];
}
template <typename REAL, int RNG>
__host__ __device__ inline size_t mem_kernel_lds_bytes(int N) {
    const size_t np2 = (size_t)mem_np2(N);
    return (np2 + np2 / 32) * 8 + (size_t)(2 * MEM_MAX_CHUNKS * MEM_NW + MEM_NW + PFG_MAX_STAT * MEM_NW + 8) * 8 +
           (size_t)(PFG_MAX_PRED * MEM_NW + MEM_NW + 2 * PFG_MAX_PRED) * 8 + tab_bytes<REAL, RNG, true>();
}

template <int MODEL, int KERNEL, typename REAL, int RNG, bool PARIS = false>
__global__ __launch_bounds__(MEM_NT) void pf_mem_kernel(const pfg_dev_problem *__restrict__ probs) {
    constexpr int NS = ModelDims<MODEL>::NS;
    constexpr int H = ModelDims<MODEL>::H;
    constexpr int NT = MEM_NT, NW = MEM_NW;
    extern __shared__ __align__(16) unsigned char smem[];

    const pfg_dev_problem &P = probs[blockIdx.x];
    const int N = P.N, T = P.T, t1 = P.t1, tL = P.tL;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
    const int nchunk = (N + NT - 1) / NT;
    const int np2 = mem_np2(N);
    const bool is_filter = (P.smoother == PFG_SMOOTHER_FILTER);
    const int stat = P.stat;
    const double lam_d = is_filter ? 0.0 : (PARIS ? 1.0 : P.lambduh);
    const REAL lam = (REAL)lam_d, oml = (REAL)(1.0 - lam_d);
    const bool needS_every = is_filter || (lam_d != 1.0);
    const double *__restrict__ const yv = P.y;
    const double *__restrict__ const wv = P.weights;
    const double *__restrict__ const uv = P.u;
    const double *__restrict__ const zv = P.z;

    double *cdf = reinterpret_cast<double *>(smem);                 // [np2 + np2/32] physical
    double *red_scan = cdf + (np2 + np2 / 32);                      // [MAX_CHUNKS*NW] wave totals
    double *red_off = red_scan + MEM_MAX_CHUNKS * NW;               // [MAX_CHUNKS*NW] exclusive offsets
    double *red_max = red_off + MEM_MAX_CHUNKS * NW;                // [NW]
    float *red_maxf = reinterpret_cast<float *>(red_max);
    double *red_S = red_max + NW;                                   // [H*NW]
    double *red_W = red_S + PFG_MAX_STAT * NW;                      // [1] grand total (+ spare)
    // predictive log-likelihood (PFG_STAT_PREDICTIVE): column maxima, weighted sum, accumulators
    double *red_pmax = red_W + 8;                                   // [MAX_PRED][NW]
    double *red_pt = red_pmax + PFG_MAX_PRED * NW;                  // [NW]
    double *pmaxv = red_pt + NW;                                    // [MAX_PRED] column maxima
    double *predv = pmaxv + PFG_MAX_PRED;                           // [MAX_PRED] out['statistics']
    double *tabmem = predv + PFG_MAX_PRED;

    constexpr int REC = mem_rec_len<MODEL, REAL>();
    REAL *lwg = reinterpret_cast<REAL *>(P.scratch);                // [N]
    // records start 16-byte aligned behind the log-weights
    REAL *cur = reinterpret_cast<REAL *>((reinterpret_cast<uintptr_t>(lwg + N) + 15) & ~(uintptr_t)15);   // [N][REC]
    REAL *nxt = cur + (size_t)REC * N;
    // PaRIS extras behind the two record buffers: children's log-weights, fallback queue
    REAL *lwn_g = reinterpret_cast<REAL *>((reinterpret_cast<uintptr_t>(cur + 2 * (size_t)REC * N) + 15) & ~(uintptr_t)15);
    REAL *qum = lwn_g + N;                                           // [N] fallback uniforms
    int *qchild = reinterpret_cast<int *>(qum + N);                  // [N]
    int *qres = qchild + N;                                          // [N]
    int *wq0 = qres + N;                                             // [nchunk*NT] wave-local queues (ping)
    int *wq1 = wq0 + (size_t)nchunk * MEM_NT;                        // (pong)
    int *qcount = reinterpret_cast<int *>(red_W + 1);                // LDS
    // predictive: the statistic of the newest step, [lead k][particle]; folded into predv by the
    // NEXT iteration's normalisation (its weights are log_normalize(new_logw), pf.py:72-76)
    const bool predictive = (stat == PFG_STAT_PREDICTIVE);
    const int KP = predictive ? P.num_steps_ahead + 1 : 0;
    REAL *const pa = reinterpret_cast<REAL *>(P.pred_scratch);     // [KP][N]
    int nact_prev = 0;                                              // leads with t+k < T at the last step
    if (tid < PFG_MAX_PRED) predv[tid] = 0.0;

    Math<REAL, true> mth;
    mth.t.e2 = tabmem;
    mth.t.lg = reinterpret_cast<const double2 *>(tabmem + TAB_E2);
    mth.t.sc = reinterpret_cast<const double2 *>(tabmem + TAB_E2 + 2 * TAB_LG);
    if (tab_bytes<REAL, RNG, true>() > 0) tab_fill(tabmem, RNG == PFG_RNG_DEVICE, tid, NT);
    for (int i = N + tid; i < np2; i += NT) cdf[cdf_phys(i)] = 2.0;  // sentinel: never <= u

    const Consts<REAL> c = make_consts<MODEL, REAL>(P.theta);
    LaneRng rng = {};
    if (RNG == PFG_RNG_DEVICE)
        rng = lane_rng_init(P.seed, P.stream, P.step_ctr ? *P.step_ctr : 0ull, (uint32_t)tid);

    // ---- x0 or warm start ---------------------------------------------------------------
    {
        double pv = P.prior_var;
        if (MODEL == PFG_MODEL_GARCH && (P.flags & PFG_FLAG_GARCH_STATIONARY_PRIOR))
            pv = (double)c.alpha / (1.0 - (double)c.beta - (double)c.gamma);
        const double sd = sqrt(pv);
        for (int i = tid; i < N; i += NT) {
            REAL x[NS], s[H], l0 = (REAL)0;
#pragma unroll
            for (int d = 0; d < NS; ++d) x[d] = (REAL)0;
#pragma unroll
            for (int h = 0; h < H; ++h) s[h] = (REAL)0;
            if (P.init_x) {
#pragma unroll
                for (int d = 0; d < NS; ++d) x[d] = (REAL)P.init_x[(size_t)i * NS + d];
                l0 = (REAL)P.init_logw[i];
                if (P.init_stats && !is_filter) {
#pragma unroll
                    for (int h = 0; h < H; ++h) s[h] = (REAL)P.init_stats[(size_t)i * H + h];
                }
            } else {
                double z;
                if (RNG == PFG_RNG_REPLAY) z = P.z0[i];
                else { REAL a, b; mth.normal_pair(rng.next(), rng.next(), a, b); z = (double)a; }
                x[0] = (REAL)(P.prior_mean + sd * z);
            }
            lwg[i] = l0;
            alignas(16) REAL rec[REC] = {};
#pragma unroll
            for (int d = 0; d < NS; ++d) rec[d] = x[d];
#pragma unroll
            for (int h = 0; h < H; ++h) rec[NS + h] = s[h];
            rec_store<REC, REAL>(cur + (size_t)i * REC, rec);
            if (P.trace_x) {
#pragma unroll
                for (int d = 0; d < NS; ++d) P.trace_x[(size_t)i * NS + d] = (double)x[d];
                P.trace_logw[i] = (double)l0;
                if (P.trace_stats && !is_filter) {
#pragma unroll
                    for (int h = 0; h < H; ++h) P.trace_stats[(size_t)i * H + h] = (double)s[h];
                }
            }
        }
    }
    __syncthreads();

    double ll = 0.0, wt_prev = 1.0, tie = 1.0;
    double filt[H], S[H];
#pragma unroll
    for (int h = 0; h < H; ++h) { filt[h] = 0.0; S[h] = 0.0; }
    double m = 0.0, W = (double)N;

    for (int t = 0; t <= T; ++t) {
        // ---- (A) max of the log weights (f32-rounded shift, see wave_max) -------------------
        float ml = -INFINITY;
        for (int i = tid; i < N; i += NT) ml = fmaxf(ml, (float)lwg[i]);
        ml = wave_max(ml);
        if (lane == 0) red_maxf[wave] = ml;
        const bool pred_upd = predictive && t > 0;      // fold step t-1's statistic (uniform)
        if (pred_upd) {
            for (int k = 0; k < nact_prev; ++k) {        // exact fp64 column maxima (np.max(add.T, axis=1))
                double mk = -INFINITY;
                for (int i = tid; i < N; i += NT) { const double a = (double)pa[(size_t)k * N + i]; mk = a > mk ? a : mk; }
                mk = wave_max(mk);
                if (lane == 0) red_pmax[k * NW + wave] = mk;
            }
        }
        __syncthreads();                                                        // barrier 1
        {
            float mm = red_maxf[0];
#pragma unroll
            for (int w = 1; w < NW; ++w) mm = fmaxf(mm, red_maxf[w]);
            m = (double)mm;
        }
        if (pred_upd) {
            if (tid < nact_prev) {
                double mk = red_pmax[tid * NW];
#pragma unroll
                for (int w = 1; w < NW; ++w) { const double o = red_pmax[tid * NW + w]; mk = o > mk ? o : mk; }
                pmaxv[tid] = mk;
            }
            __syncthreads();                                                    // barrier 1b
        }
        // ---- (B,C) weights, per-chunk wave scans (unnormalised, wave-local) into the CDF -----
        const bool needS = needS_every || (t == T);
        {
            double ptot = 0.0;
            double part[H];
#pragma unroll
            for (int h = 0; h < H; ++h) part[h] = 0.0;
            for (int j = 0; j < nchunk; ++j) {
                const int i = j * NT + tid;
                const bool v = i < N;
                const int ii = v ? i : N - 1;
                double p = (double)mth.exp((REAL)(lwg[ii] - (REAL)m));
                p = v ? p : 0.0;
                if (needS) {
#pragma unroll
                    for (int h = 0; h < H; ++h) part[h] += (double)cur[(size_t)ii * REC + NS + h] * p;
                }
                if (pred_upd) {
                    // sum over leads AND particles of w_i exp(add_ik - max_k): the reference's
                    // np.sum has no axis (pf.py:74-76), so only the grand total is needed
                    double e = 0.0;
                    for (int k = 0; k < nact_prev; ++k)
                        e += (double)mth.exp((REAL)((double)pa[(size_t)k * N + ii] - pmaxv[k]));
                    ptot += p * e;
                }
                const double inc = wave_incl_scan(p);
                if (v) cdf[cdf_phys(i)] = inc;
                if (lane == WAVE - 1) red_scan[j * NW + wave] = inc;
            }
            if (pred_upd) {
                ptot = wave_sum(ptot);
                if (lane == 0) red_pt[wave] = ptot;
            }
            if (needS) {
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    const double tot = wave_sum(part[h]);
                    if (lane == 0) red_S[h * NW + wave] = tot;
                }
            }
        }
        __syncthreads();                                                        // barrier 2
        if (wave == 0) {
            // exclusive offsets of the nchunk*NW wave totals (<= 256): 4 per lane + one wave scan
            const int ntot = nchunk * NW;
            double v4[4], loc = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = lane * 4 + q;
                v4[q] = idx < ntot ? red_scan[idx] : 0.0;
                loc += v4[q];
            }
            const double inc = wave_incl_scan(loc);
            double run = inc - loc;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = lane * 4 + q;
                if (idx < ntot) red_off[idx] = run;
                run += v4[q];
            }
            if (lane == WAVE - 1) red_W[0] = inc;
        }
        __syncthreads();                                                        // barrier 2b
        W = red_W[0];
        const double invW = 1.0 / W;
        if (needS) {
#pragma unroll
            for (int h = 0; h < H; ++h) {
                double acc = 0.0;
#pragma unroll
                for (int w = 0; w < NW; ++w) acc += red_S[h * NW + w];
                S[h] = acc * invW;
            }
        }
        if (wave == 0) {
            if (t > 0 && (t - 1) >= t1 && (t - 1) < tL) ll += wt_prev * (m + log(W / (double)N));
            if (P.trace_ll && tid == 0) P.trace_ll[t] = ll;
        }
        if (is_filter && t > 0) {
#pragma unroll
            for (int h = 0; h < H; ++h) filt[h] += S[h];
        }
        if (pred_upd && tid < KP) {
            // stats_k += max_k + log(sum): leads without a statistic (outside the window, or
            // t+k >= T) have add = 0, i.e. max 0 and a unit contribution to the sum each
            double tot = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) tot += red_pt[w];
            tot = tot * invW + (double)(KP - nact_prev);
            predv[tid] = (predv[tid] + (tid < nact_prev ? pmaxv[tid] : 0.0)) + log(tot);
        }
        if (t == T) break;

        // ---- (D) normalise the CDF in place (own entries) -------------------------------------
        for (int j = 0; j < nchunk; ++j) {
            const int i = j * NT + tid;
            if (i < N) {
                const int pi = cdf_phys(i);
                cdf[pi] = (cdf[pi] + red_off[j * NW + wave]) * invW;
            }
        }
        __syncthreads();                                                        // barrier 3

        const double y_t = yv[t];
        const bool inside = (t >= t1) && (t < tL);
        const double wt = (inside && wv) ? wv[t - t1] : 1.0;
        const bool use_stat = inside && (stat != PFG_STAT_NONE) && !predictive;
        const int nact = (predictive && inside) ? (KP < T - t ? KP : T - t) : 0;
        // ---- (E..H) per particle: ancestor search, gather parent (HBM/L2), propose, publish ---
        auto sweep = [&](auto stat_tag) {
            constexpr int STAT = decltype(stat_tag)::value;
            for (int j = 0; j < nchunk; ++j) {
                const int i = j * NT + tid;
                const bool v = i < N;
                const int ii = v ? i : N - 1;
                double u;
                REAL z;
                if (RNG == PFG_RNG_REPLAY) { u = uv[(size_t)t * N + ii]; z = (REAL)zv[(size_t)t * N + ii]; }
                else { REAL zb; u = u01_32(rng.next()); mth.normal_pair(rng.next(), rng.next(), z, zb); }
                int pos = 0;
                for (int step = np2 >> 1; step >= 1; step >>= 1) {
                    const int probe = step - 1 + (step >= 32 ? (step >> 5) - 1 : 0);
                    pos += (cdf[pos + probe] <= u) ? step + (step >> 5) : 0;
                }
                int a = pos - ((pos * 993) >> 15) ;
                if (np2 > 8192) a = pos - pos / 33;          // exact mul-shift only below 8192
                a = a < N - 1 ? a : N - 1;
                if (RNG == PFG_RNG_REPLAY && v) {
                    const double hi = cdf[cdf_phys(a)] - u;
                    const double lo = a > 0 ? u - cdf[cdf_phys(a - 1)] : 1.0;
                    const double mg = hi < lo ? hi : lo;
                    tie = mg < tie ? mg : tie;
                }
                REAL xp[NS], sp[H], xn[NS], add[H], lwn;
                alignas(16) REAL rec[REC];
                rec_load<REC, REAL>(rec, cur + (size_t)a * REC);
#pragma unroll
                for (int d = 0; d < NS; ++d) xp[d] = rec[d];
#pragma unroll
                for (int h = 0; h < H; ++h) sp[h] = rec[NS + h];
                particle_step<MODEL, KERNEL, STAT, REAL>(c, mth, xp, (REAL)y_t, z, xn, lwn, add);
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    const REAL av = use_stat ? add[h] * (REAL)wt : (REAL)0;
                    const REAL sm = (lam * sp[h] + oml * (REAL)S[h]) + av;      // pf.py:175-179 / :78-80
                    sp[h] = is_filter ? av : sm;
                }
                if (predictive && inside) {
                    // [log Pr(y_{t+k} | x_{t+1})]_k of the new particle: svm/helper.py:352-395
                    // (Ntilde = 1), lgssm/helper.py:1281-1336, garch/helper.py:374-412
                    REAL xm = xn[0], s2 = (MODEL == PFG_MODEL_GARCH) ? xn[NS - 1] : (REAL)0;
                    REAL cov = (REAL)0;
                    const REAL Qv = (MODEL == PFG_MODEL_GARCH) ? (REAL)0 : (REAL)(1.0 / (double)c.Qinv);
                    for (int k = 0; k < nact; ++k) {
                        REAL zk = (REAL)0;
                        if (MODEL != PFG_MODEL_LGSSM) {
                            if (RNG == PFG_RNG_REPLAY) zk = (REAL)P.pred_z[((size_t)t * KP + k) * N + ii];
                            else { REAL zb; mth.normal_pair(rng.next(), rng.next(), zk, zb); }
                        }
                        const REAL yk = (REAL)yv[t + k];
                        REAL a;
                        if (MODEL == PFG_MODEL_SVM) {
                            const REAL ypc = c.R * mth.exp(xm + mth.sqrt(cov) * zk);
                            a = ((REAL)-0.5 * (yk * yk) / ypc + c.c0) - (REAL)0.5 * mth.log(ypc);
                            xm = c.A * xm;
                            cov = Qv + c.A * c.A * cov;
                        } else if (MODEL == PFG_MODEL_LGSSM) {
                            const REAL diff = yk - xm * c.C;
                            const REAL ypc = c.R + c.C * (cov * c.C);
                            a = ((REAL)-0.5 * (diff * diff) / ypc + c.c0) - (REAL)0.5 * mth.log(ypc);
                            xm = xm * c.A;
                            cov = Qv + c.A * (cov * c.A);
                        } else {
                            const REAL diff = yk - xm;
                            a = ((REAL)-0.5 * (diff * diff) / c.R + c.c0) - (REAL)0.5 * mth.log(c.R);
                            const REAL s2n = c.alpha + c.beta * (xm * xm) + c.gamma * s2;   // prior_kernel.rv
                            xm = mth.sqrt(s2n) * zk;
                            s2 = s2n;
                        }
                        if (v) pa[(size_t)k * N + i] = a * (REAL)wt;
                    }
                }
                if (v) {
                    lwg[i] = lwn;
#pragma unroll
                    for (int d = 0; d < NS; ++d) rec[d] = xn[d];
#pragma unroll
                    for (int h = 0; h < H; ++h) rec[NS + h] = sp[h];
                    rec_store<REC, REAL>(nxt + (size_t)i * REC, rec);
                    if (P.trace_x) {
                        const size_t row = (size_t)(t + 1) * N + i;
                        if (P.trace_anc) P.trace_anc[(size_t)t * N + i] = a;
#pragma unroll
                        for (int d = 0; d < NS; ++d) P.trace_x[row * NS + d] = (double)xn[d];
                        P.trace_logw[row] = (double)lwn;
                        if (P.trace_stats && !is_filter) {
#pragma unroll
                            for (int h = 0; h < H; ++h) P.trace_stats[row * H + h] = (double)sp[h];
                        }
                    }
                }
            }
        };
        // PaRIS for N > 1024 (pf.py:183-341): as pf_reg_kernel's paris_slots, with the particle state
        // in the L2-resident scratch.  Per backward draw: accept-reject rounds per child against the
        // filter CDF; children that never accept queue up and are served one per wave by an exact
        // categorical draw over all parents (chunk sums kept one per lane, index order preserved).
        auto paris_sweep = [&](auto stat_tag) {
            constexpr int STAT = decltype(stat_tag)::value;
            const int Nt = P.Ntilde, R = P.max_accept_reject;
            const double *__restrict__ const pidx = P.paris_idx_u;
            const double *__restrict__ const pacc = P.paris_acc_u;
            const double *__restrict__ const pman = P.paris_man_u;
            auto search = [&](double u) {
                int pos = 0;
                for (int step = np2 >> 1; step >= 1; step >>= 1) {
                    const int probe = step - 1 + (step >= 32 ? (step >> 5) - 1 : 0);
                    pos += (cdf[pos + probe] <= u) ? step + (step >> 5) : 0;
                }
                int a = pos - pos / 33;
                return a < N - 1 ? a : N - 1;
            };
            // ---- 1. propose every child from its filter ancestor, publish x' and log-weight ----
            for (int j = 0; j < nchunk; ++j) {
                const int i = j * NT + tid;
                const bool v = i < N;
                const int ii = v ? i : N - 1;
                double u;
                REAL z;
                if (RNG == PFG_RNG_REPLAY) { u = uv[(size_t)t * N + ii]; z = (REAL)zv[(size_t)t * N + ii]; }
                else { REAL zb; u = u01_32(rng.next()); mth.normal_pair(rng.next(), rng.next(), z, zb); }
                const int a = search(u);
                if (RNG == PFG_RNG_REPLAY && v) {
                    const double hi = cdf[cdf_phys(a)] - u;
                    const double lo = a > 0 ? u - cdf[cdf_phys(a - 1)] : 1.0;
                    const double mg = hi < lo ? hi : lo;
                    tie = mg < tie ? mg : tie;
                }
                REAL xp[NS], xn[NS], add[H], lwn;
                alignas(16) REAL rec[REC];
                rec_load<REC, REAL>(rec, cur + (size_t)a * REC);
#pragma unroll
                for (int d = 0; d < NS; ++d) xp[d] = rec[d];
                particle_step<MODEL, KERNEL, STAT, REAL>(c, mth, xp, (REAL)y_t, z, xn, lwn, add);
                if (v) {
                    lwn_g[i] = lwn;
#pragma unroll
                    for (int q = 0; q < REC; ++q) rec[q] = (REAL)0;
#pragma unroll
                    for (int d = 0; d < NS; ++d) rec[d] = xn[d];
                    rec_store<REC, REAL>(nxt + (size_t)i * REC, rec);
                    if (P.trace_x && P.trace_anc) P.trace_anc[(size_t)t * N + i] = a;
                }
            }
            __syncthreads();
            // contribution of parent J to child ci:  stats[J] + w_t h(x_J, x_ci), added to the child's record
            auto contribute = [&](int ci, int J) {
                alignas(16) REAL rc[REC], rp[REC];
                rec_load<REC, REAL>(rc, nxt + (size_t)ci * REC);
                rec_load<REC, REAL>(rp, cur + (size_t)J * REC);
                const REAL aux = (MODEL == PFG_MODEL_SVM) ? mth.exp(-rc[0]) : (REAL)0;
                REAL aj[H];
                additive_stat<MODEL, STAT, REAL>(c, rp, rc, (REAL)y_t, aux, aj);
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    const REAL a = use_stat ? aj[h] * (REAL)wt : (REAL)0;
                    rc[NS + h] += rp[NS + h] + a;
                }
                rec_store<REC, REAL>(nxt + (size_t)ci * REC, rc);
            };
            const unsigned long long ltmask = (1ull << lane) - 1ull;
            for (int jt = 0; jt < Nt; ++jt) {
                if (tid == 0) *qcount = 0;
                __syncthreads();
                // ---- 2. accept-reject against the filter weights, up to R rounds per child ------
                // as pf_reg_kernel's paris_slots: pending children compacted in wave-local queues
                // (here in the scratch), one candidate per child and pass while more than half a
                // wave is pending, K = 2^k consecutive rounds per child and pass below that
                int *qa = wq0 + wave * (nchunk * WAVE), *qb = wq1 + wave * (nchunk * WAVE);
                int cnt = 0;
                for (int j = 0; j < nchunk; ++j) {
                    const int i = j * NT + tid;
                    const bool v = i < N;
                    const unsigned long long mk = __ballot(v);
                    if (v) qa[cnt + __popcll(mk & ltmask)] = i;
                    cnt += __popcll(mk);
                }
                auto candidate = [&](int child, int round, bool act, int &Iout) {
                    double u1, u2;
                    if (RNG == PFG_RNG_REPLAY) {
                        const size_t at = (((size_t)t * Nt + jt) * R + (act ? round : 0)) * N + child;
                        u1 = pidx[at]; u2 = pacc[at];
                    } else { u1 = u01_32(rng.next()); u2 = u01_32(rng.next()); }
                    const int I = search(u1);
                    REAL xI[NS], xc[NS];
#pragma unroll
                    for (int d = 0; d < NS; ++d) { xI[d] = cur[(size_t)I * REC + d]; xc[d] = nxt[(size_t)child * REC + d]; }
                    const double thr = (double)mth.exp(backward_log_ratio<MODEL, REAL>(c, mth, xI, xc));
                    Iout = I;
                    return act && u2 <= thr;
                };
                int r0 = 0;
                while (cnt > 0 && r0 < R) {                       // wave-uniform
                    __threadfence_block();                        // queue stores visible to the other lanes
                    int ncnt = 0;
                    if (cnt > WAVE / 2) {
                        for (int e0 = 0; e0 < cnt; e0 += WAVE) {
                            const int e = e0 + lane;
                            const bool act = e < cnt;
                            const int child = qa[act ? e : 0];
                            int I;
                            const bool acc = candidate(child, r0, act, I);
                            if (acc) contribute(child, I);
                            const bool rej = act && !acc;
                            const unsigned long long mk = __ballot(rej);
                            if (rej) qb[ncnt + __popcll(mk & ltmask)] = child;
                            ncnt += __popcll(mk);
                        }
                        r0 += 1;
                    } else {
                        int logK = 1;
                        while ((cnt << (logK + 1)) <= WAVE) ++logK;           // cnt * 2^logK <= 64
                        const int K = 1 << logK;
                        const int e = lane >> logK, o = lane & (K - 1);
                        const bool have = e < cnt;
                        const bool act = have && (r0 + o) < R;
                        const int child = qa[have ? e : 0];
                        int I;
                        const bool acc = candidate(child, r0 + o, act, I);
                        const unsigned long long am = __ballot(acc);
                        const unsigned long long segmask = (K >= 64) ? ~0ull : ((1ull << K) - 1ull);
                        const unsigned long long seg = (am >> (e << logK)) & segmask;
                        const int first = __ffsll((long long)seg) - 1;       // lowest accepting round
                        if (acc && o == first) contribute(child, I);
                        const bool rej = have && o == 0 && seg == 0ull;
                        const unsigned long long mk = __ballot(rej);
                        if (rej) qb[__popcll(mk & ltmask)] = child;
                        ncnt = __popcll(mk);
                        r0 += K;
                    }
                    { int *tq = qa; qa = qb; qb = tq; }
                    cnt = ncnt;
                }
                __threadfence_block();
                for (int e0 = 0; e0 < cnt; e0 += WAVE) {          // never accepted: exact draw below
                    const int e = e0 + lane;
                    if (e < cnt) {
                        const int i = qa[e];
                        const int slot = atomicAdd(qcount, 1);
                        qchild[slot] = i;
                        qum[slot] = (REAL)((RNG == PFG_RNG_REPLAY) ? pman[((size_t)t * Nt + jt) * N + i]
                                                                    : u01_32(rng.next()));
                    }
                }
                __syncthreads();
                const int nq = *qcount;
                // ---- 3. exact categorical draw for the queued children, one child per wave -------
                const int nch64 = (N + WAVE - 1) / WAVE;           // <= 256: chunk sums kept [4] per lane
                for (int e = wave; e < nq; e += NW) {
                    const int ci = qchild[e];
                    REAL xc[NS];
#pragma unroll
                    for (int d = 0; d < NS; ++d) xc[d] = nxt[(size_t)ci * REC + d];
                    const double um = (double)qum[e];
                    auto logit = [&](int q) {
                        REAL xq[NS];
#pragma unroll
                        for (int d = 0; d < NS; ++d) xq[d] = cur[(size_t)q * REC + d];
                        return lwg[q] + backward_log_ratio<MODEL, REAL>(c, mth, xq, xc);
                    };
                    if constexpr (RNG == PFG_RNG_DEVICE) {
                        // device generator: lane-major enumeration (see paris_slots).  Pass 1: per-lane
                        // sums of the lane's own parents lane, lane+64, ...; one wave scan picks the lane;
                        // pass 2: the wave re-evaluates that lane's <= 256 entries together.
                        REAL mm = (REAL)m;                           // fp64: block max of the parents' lw
                        if (sizeof(REAL) == 4) {
                            float mxf2 = -INFINITY;
                            for (int q = lane; q < N; q += WAVE) mxf2 = fmaxf(mxf2, (float)logit(q));
                            mm = (REAL)wave_max(mxf2);
                        }
                        double tl = 0.0;
                        for (int q = lane; q < N; q += WAVE) tl += (double)mth.exp((REAL)(logit(q) - mm));
                        const double incl = wave_incl_scan(tl);
                        const double target = um * bcast_lane63(incl);
                        int Lsel = (int)wave_sum(incl <= target ? 1.0 : 0.0);
                        Lsel = __builtin_amdgcn_readfirstlane(Lsel < WAVE - 1 ? Lsel : WAVE - 1);
                        const double locl = target - (incl - tl);
                        const double loc = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(locl), Lsel),
                                                            __builtin_amdgcn_readlane(__double2loint(locl), Lsel));
                        const int nown = (N - Lsel + WAVE - 1) / WAVE;      // entries of lane Lsel (>= 1)
                        int nle = 0;
                        double base = 0.0;
                        for (int sI = 0; sI * WAVE < nown; ++sI) {
                            const int mI = sI * WAVE + lane;
                            const bool ok = mI < nown;
                            const int q = ok ? mI * WAVE + Lsel : Lsel;
                            const double ev = ok ? (double)mth.exp((REAL)(logit(q) - mm)) : 0.0;
                            const double inc = wave_incl_scan(ev) + base;
                            nle += (ok && inc <= loc) ? 1 : 0;
                            base = bcast_lane63(inc);
                        }
                        int msel = (int)wave_sum((double)nle);
                        msel = msel < nown - 1 ? msel : nown - 1;
                        if (lane == 0) qres[e] = msel * WAVE + Lsel;
                        continue;
                    }
                    float mxf = -INFINITY;
                    for (int q = lane; q < N; q += WAVE) mxf = fmaxf(mxf, (float)logit(q));
                    const REAL mm = (REAL)wave_max(mxf);
                    double keep[4] = {0.0, 0.0, 0.0, 0.0};        // chunk s*64 + lane lives in keep[s]
                    double tot = 0.0;
#pragma unroll
                    for (int sI = 0; sI < 4; ++sI) {
                        for (int cl = 0; cl < WAVE; ++cl) {
                            const int ch = sI * WAVE + cl;
                            if (ch >= nch64) break;
                            const int q = ch * WAVE + lane;
                            const double ev = q < N ? (double)mth.exp((REAL)(logit(q < N ? q : N - 1) - mm)) : 0.0;
                            const double cs = wave_sum(ev);
                            keep[sI] = (lane == cl) ? cs : keep[sI];
                            tot += cs;
                        }
                    }
                    const double target = um * tot;
                    // chunk holding the target: first chunk whose inclusive running sum exceeds it
                    int nle = 0;
                    double base = 0.0, before = 0.0;
                    double incs[4];
#pragma unroll
                    for (int sI = 0; sI < 4; ++sI) {
                        const double inc = wave_incl_scan(keep[sI]) + base;
                        incs[sI] = inc;
                        const bool validc = (sI * WAVE + lane) < nch64;
                        nle += (validc && inc <= target) ? 1 : 0;
                        base = bcast_lane63(inc);
                    }
                    int msel = (int)wave_sum((double)nle);
                    msel = msel < nch64 - 1 ? msel : nch64 - 1;
#pragma unroll
                    for (int sI = 0; sI < 4; ++sI) {
                        const bool here = (sI * WAVE + lane) == msel;
                        before += here ? incs[sI] - keep[sI] : 0.0;
                    }
                    before = wave_sum(before);
                    const int q = msel * WAVE + lane;
                    const double ev = q < N ? (double)mth.exp((REAL)(logit(q < N ? q : N - 1) - mm)) : 0.0;
                    const double inc = wave_incl_scan(ev) + before;
                    int cnt = (q < N && inc <= target) ? 1 : 0;
                    cnt = msel * WAVE + (int)wave_sum((double)cnt);
                    if (lane == 0) qres[e] = cnt < N - 1 ? cnt : N - 1;
                }
                __syncthreads();
                // ---- 4. queued children: rewired parent's contribution ---------------------------
                for (int e = tid; e < nq; e += NT) contribute(qchild[e], qres[e]);
                __syncthreads();
            }
            // ---- 5. average over the Ntilde draws, traces -----------------------------------------
            for (int j = 0; j < nchunk; ++j) {
                const int i = j * NT + tid;
                if (i < N) {
                    alignas(16) REAL rc[REC];
                    rec_load<REC, REAL>(rc, nxt + (size_t)i * REC);
#pragma unroll
                    for (int h = 0; h < H; ++h) rc[NS + h] = rc[NS + h] / (REAL)Nt;
                    rec_store<REC, REAL>(nxt + (size_t)i * REC, rc);
                    if (P.trace_x) {
                        const size_t row = (size_t)(t + 1) * N + i;
#pragma unroll
                        for (int d = 0; d < NS; ++d) P.trace_x[row * NS + d] = (double)rc[d];
                        P.trace_logw[row] = (double)lwn_g[i];
                        if (P.trace_stats) {
#pragma unroll
                            for (int h = 0; h < H; ++h) P.trace_stats[row * H + h] = (double)rc[NS + h];
                        }
                    }
                }
            }
            { REAL *tmp = lwg; lwg = lwn_g; lwn_g = tmp; }
        };
        if constexpr (PARIS) {
            if (stat == PFG_STAT_SCORE) paris_sweep(std::integral_constant<int, PFG_STAT_SCORE>{});
            else paris_sweep(std::integral_constant<int, PFG_STAT_SUFF>{});
        } else {
            if (stat == PFG_STAT_SCORE) sweep(std::integral_constant<int, PFG_STAT_SCORE>{});
            else sweep(std::integral_constant<int, PFG_STAT_SUFF>{});
        }
        { REAL *tmp = cur; cur = nxt; nxt = tmp; }
        wt_prev = wt;
        nact_prev = nact;
        // children (global stores) must be visible to next step's gathers: barrier 1 of the next
        // iteration orders them (__syncthreads = waitcnt + workgroup barrier, same CU / same L1)
    }

    // ---- outputs --------------------------------------------------------------------------
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
        P.out[4] = ll; P.out[5] = W; P.out[6] = m; P.out[7] = tie;
    }
    if (predictive && P.pred_out) {
        __syncthreads();
        if (tid < PFG_MAX_PRED) P.pred_out[tid] = tid < KP ? predv[tid] : 0.0;
    }
    if (P.final_x) {
        for (int i = tid; i < N; i += NT) {
#pragma unroll
            for (int d = 0; d < NS; ++d) P.final_x[(size_t)i * NS + d] = (double)cur[(size_t)i * REC + d];
            if (P.final_logw) P.final_logw[i] = (double)lwg[i];
            if (P.final_stats && !is_filter) {
#pragma unroll
                for (int h = 0; h < H; ++h) P.final_stats[(size_t)i * H + h] = (double)cur[(size_t)i * REC + NS + h];
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// Large-N kernel, device-RNG fast path (N <= NP2, NP2 = 4096 | 16384): same phases and scratch
// layout as pf_mem_kernel, restructured around what the device generator allows:
//  * the resampling CDF is built in THREAD-major order (position tid*CH2 + j <-> particle j*1024+tid;
//    resampling does not care how particles are labelled): one in-register running sum and ONE
//    wave scan per thread-step instead of one scan per 1024-particle chunk;
//  * the binary search is unrolled for the compile-time NP2 (probe offsets fold into the ds_read
//    immediates) and two chunks are in flight per iteration (two independent search / gather
//    chains per lane, and both normals of a Box-Muller pair are used);
//  * the NW wave totals are prefix-summed redundantly by every wave with one DPP row scan,
//    which drops a barrier.
// REPLAY (reference index order), PaRIS and the predictive statistic stay on pf_mem_kernel.
// ------------------------------------------------------------------------------------
template <typename REAL>
__host__ __device__ inline size_t big_kernel_lds_bytes(int NP2) {
    return ((size_t)NP2 + NP2 / 32) * 8 + (size_t)(2 * MEM_NW + PFG_MAX_STAT * MEM_NW + 8) * 8 +
           tab_bytes<REAL, PFG_RNG_DEVICE, true>();
}

template <int MODEL, int KERNEL, typename REAL, int NP2>
__global__ __launch_bounds__(MEM_NT) void pf_big_kernel(const pfg_dev_problem *__restrict__ probs) {
    constexpr int RNG = PFG_RNG_DEVICE;
    constexpr int NS = ModelDims<MODEL>::NS;
    constexpr int H = ModelDims<MODEL>::H;
    constexpr int NT = MEM_NT, NW = MEM_NW;
    constexpr int CH2 = NP2 / NT;                                   // CDF positions per thread (4 | 16)
    constexpr int LOG_CH2 = CH2 == 4 ? 2 : 4;
    static_assert(CH2 == 4 || CH2 == 16, "NP2 must be 4096 or 16384");
    static_assert(NW == 16, "the wave-total prefix is one 16-lane DPP row scan");
    constexpr int G = 2;                                            // chunks in flight
    // NP2 = 4096, f32 state: a thread's (<= 4) log-weights never leave its registers (it is the only
    // reader and writer of its particles' weights): 8 of the 40 B per particle-step stay out of
    // memory (measured 8.66 -> 7.53 ms per 256 windows of N = 4000).  In fp64 the 8 extra VGPRs
    // push the kernel over the 128-VGPR cap of a 1024-thread workgroup (53 spills, 15.5 -> 18.6 ms).
    constexpr bool LWREG = (CH2 == 4) && sizeof(REAL) == 4;
    extern __shared__ __align__(16) unsigned char smem[];

    const pfg_dev_problem &P = probs[blockIdx.x];
    const int N = P.N, T = P.T, t1 = P.t1, tL = P.tL;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
    const int nchunk = (N + NT - 1) / NT;                           // <= CH2
    const bool is_filter = (P.smoother == PFG_SMOOTHER_FILTER);
    const int stat = P.stat;
    const double lam_d = is_filter ? 0.0 : P.lambduh;
    const REAL lam = (REAL)lam_d, oml = (REAL)(1.0 - lam_d);
    const bool needS_every = is_filter || (lam_d != 1.0);
    const double *__restrict__ const yv = P.y;
    const double *__restrict__ const wv = P.weights;

    double *cdf = reinterpret_cast<double *>(smem);                 // [NP2 + NP2/32] physical
    double *red_scan = cdf + (NP2 + NP2 / 32);                      // [NW] wave totals
    double *red_max = red_scan + NW;                                // [NW]
    float *red_maxf = reinterpret_cast<float *>(red_max);
    double *red_S = red_max + NW;                                   // [H*NW]
    double *tabmem = red_S + PFG_MAX_STAT * NW + 8;

    constexpr int REC = mem_rec_len<MODEL, REAL>();
    REAL *lwg = reinterpret_cast<REAL *>(P.scratch);                // [N]
    REAL *cur = reinterpret_cast<REAL *>((reinterpret_cast<uintptr_t>(lwg + N) + 15) & ~(uintptr_t)15);   // [N][REC]
    REAL *nxt = cur + (size_t)REC * N;

    Math<REAL, true> mth;
    mth.t.e2 = tabmem;
    mth.t.lg = reinterpret_cast<const double2 *>(tabmem + TAB_E2);
    mth.t.sc = reinterpret_cast<const double2 *>(tabmem + TAB_E2 + 2 * TAB_LG);
    tab_fill(tabmem, true, tid, NT);

    const Consts<REAL> c = make_consts<MODEL, REAL>(P.theta);
    LaneRng rng = lane_rng_init(P.seed, P.stream, P.step_ctr ? *P.step_ctr : 0ull, (uint32_t)tid);
    REAL lwr[LWREG ? CH2 : 1];
#pragma unroll
    for (int j = 0; j < (LWREG ? CH2 : 1); ++j) lwr[j] = (REAL)(-INFINITY);

    // ---- x0 or warm start ---------------------------------------------------------------
    {
        double pv = P.prior_var;
        if (MODEL == PFG_MODEL_GARCH && (P.flags & PFG_FLAG_GARCH_STATIONARY_PRIOR))
            pv = (double)c.alpha / (1.0 - (double)c.beta - (double)c.gamma);
        const double sd = sqrt(pv);
#pragma unroll (LWREG ? CH2 : 1)
        for (int jj = 0; jj < (LWREG ? CH2 : MEM_MAX_CHUNKS); ++jj) {
            const int i = jj * NT + tid;
            if (i >= N) break;
            REAL x[NS], s[H], l0 = (REAL)0;
#pragma unroll
            for (int d = 0; d < NS; ++d) x[d] = (REAL)0;
#pragma unroll
            for (int h = 0; h < H; ++h) s[h] = (REAL)0;
            if (P.init_x) {
#pragma unroll
                for (int d = 0; d < NS; ++d) x[d] = (REAL)P.init_x[(size_t)i * NS + d];
                l0 = (REAL)P.init_logw[i];
                if (P.init_stats && !is_filter) {
#pragma unroll
                    for (int h = 0; h < H; ++h) s[h] = (REAL)P.init_stats[(size_t)i * H + h];
                }
            } else {
                REAL a, b;
                mth.normal_pair(rng.next(), rng.next(), a, b);
                x[0] = (REAL)(P.prior_mean + sd * (double)a);
            }
            if (LWREG) lwr[LWREG ? jj : 0] = l0;
            else lwg[i] = l0;
            alignas(16) REAL rec[REC] = {};
#pragma unroll
            for (int d = 0; d < NS; ++d) rec[d] = x[d];
#pragma unroll
            for (int h = 0; h < H; ++h) rec[NS + h] = s[h];
            rec_store<REC, REAL>(cur + (size_t)i * REC, rec);
            if (P.trace_x) {
#pragma unroll
                for (int d = 0; d < NS; ++d) P.trace_x[(size_t)i * NS + d] = (double)x[d];
                P.trace_logw[i] = (double)l0;
                if (P.trace_stats && !is_filter) {
#pragma unroll
                    for (int h = 0; h < H; ++h) P.trace_stats[(size_t)i * H + h] = (double)s[h];
                }
            }
        }
    }
    __syncthreads();

    double ll = 0.0, wt_prev = 1.0;
    double filt[H], S[H];
#pragma unroll
    for (int h = 0; h < H; ++h) { filt[h] = 0.0; S[h] = 0.0; }
    double m = 0.0, W = (double)N;

    for (int t = 0; t <= T; ++t) {
        // ---- (A) max of the log weights (f32-rounded shift, see wave_max) -------------------
        float ml = -INFINITY;
        if (LWREG) {
#pragma unroll
            for (int j = 0; j < (LWREG ? CH2 : 1); ++j) ml = fmaxf(ml, (float)lwr[j]);   // slots past N hold -inf
        } else {
            for (int i = tid; i < N; i += NT) ml = fmaxf(ml, (float)lwg[i]);
        }
        ml = wave_max(ml);
        if (lane == 0) red_maxf[wave] = ml;
        __syncthreads();                                                        // barrier 1
        {
            float mm = red_maxf[0];
#pragma unroll
            for (int w = 1; w < NW; ++w) mm = fmaxf(mm, red_maxf[w]);
            m = uniform_f64((double)mm);
        }
        // ---- (B,C) weights; thread-local running sums into the CDF, one wave scan ------------
        const bool needS = needS_every || (t == T);
        double thr_exc;
        {
            double part[H], run = 0.0;
#pragma unroll
            for (int h = 0; h < H; ++h) part[h] = 0.0;
#pragma unroll 2
            for (int j = 0; j < CH2; ++j) {
                if (j < nchunk) {
                    const int i = j * NT + tid;
                    const bool v = i < N;
                    const int ii = v ? i : N - 1;
                    const REAL lwv = LWREG ? lwr[LWREG ? j : 0] : lwg[ii];
                    double p = (double)mth.exp((REAL)(lwv - (REAL)m));
                    p = v ? p : 0.0;
                    if (needS) {
#pragma unroll
                        for (int h = 0; h < H; ++h) part[h] += (double)cur[(size_t)ii * REC + NS + h] * p;
                    }
                    run += p;
                }
                cdf[cdf_phys(tid * CH2 + j)] = run;               // positions past nchunk: flat
            }
            const double inc = wave_incl_scan(run);
            thr_exc = inc - run;
            if (lane == WAVE - 1) red_scan[wave] = inc;
            if (needS) {
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    const double tot = wave_sum(part[h]);
                    if (lane == 0) red_S[h * NW + wave] = tot;
                }
            }
        }
        __syncthreads();                                                        // barrier 2
        double off_w;
        {
            // every wave: exclusive prefix of the 16 wave totals by one DPP row scan
            const double tot = (lane < NW) ? red_scan[lane] : 0.0;
            double inc = tot;
            inc += dpp_shr0_f64<0x111>(inc);
            inc += dpp_shr0_f64<0x112>(inc);
            inc += dpp_shr0_f64<0x114>(inc);
            inc += dpp_shr0_f64<0x118>(inc);
            const double exc = inc - tot;
            off_w = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(exc), wave),
                                     __builtin_amdgcn_readlane(__double2loint(exc), wave));
            W = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(inc), NW - 1),
                                 __builtin_amdgcn_readlane(__double2loint(inc), NW - 1));
        }
        const double invW = uniform_f64(1.0 / W);
        if (needS) {
#pragma unroll
            for (int h = 0; h < H; ++h) {
                double acc = 0.0;
#pragma unroll
                for (int w = 0; w < NW; ++w) acc += red_S[h * NW + w];
                S[h] = uniform_f64(acc * invW);
            }
        }
        if (wave == 0) {
            if (t > 0 && (t - 1) >= t1 && (t - 1) < tL) ll = uniform_f64(ll + wt_prev * (m + log(W / (double)N)));
            if (P.trace_ll && tid == 0) P.trace_ll[t] = ll;
        }
        if (is_filter && t > 0) {
#pragma unroll
            for (int h = 0; h < H; ++h) filt[h] = uniform_f64(filt[h] + S[h]);
        }
        if (t == T) break;

        // ---- (D) globalise + normalise the own CDF entries ---------------------------------
        {
            const double off = thr_exc + off_w;
#pragma unroll 2
            for (int j = 0; j < CH2; ++j) {
                const int pi = cdf_phys(tid * CH2 + j);
                cdf[pi] = (cdf[pi] + off) * invW;
            }
        }
        __syncthreads();                                                        // barrier 3

        const double y_t = yv[t];
        const bool inside = (t >= t1) && (t < tL);
        const double wt = (inside && wv) ? wv[t - t1] : 1.0;
        const bool use_stat = inside && (stat != PFG_STAT_NONE);
        // ---- (E..H) two chunks per iteration: search, gather parent (L2), propose, publish ----
        auto sweep = [&](auto stat_tag) {
            constexpr int STAT = decltype(stat_tag)::value;
            for (int j0 = 0; j0 < nchunk; j0 += G) {
                int i[G], a[G];
                bool v[G];
                double u[G];
                REAL z[G];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    i[g] = (j0 + g) * NT + tid;
                    v[g] = i[g] < N;
                    u[g] = u01_32(rng.next());
                    a[g] = 0;
                }
                mth.normal_pair(rng.next(), rng.next(), z[0], z[1]);
#pragma unroll
                for (int step = NP2 >> 1; step >= 1; step >>= 1) {
                    const int probe = step - 1 + (step >= 32 ? (step >> 5) - 1 : 0);
                    const int adv = step + (step >> 5);
#pragma unroll
                    for (int g = 0; g < G; ++g) a[g] += (cdf[a[g] + probe] <= u[g]) ? adv : 0;
                }
                alignas(16) REAL rec[G][REC];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    a[g] -= (a[g] * 993) >> 15;                    // physical -> CDF position (exact < 32768)
                    a[g] = (a[g] & (CH2 - 1)) * NT + (a[g] >> LOG_CH2);   // -> particle index
                    a[g] = a[g] < N - 1 ? a[g] : N - 1;
                    rec_load<REC, REAL>(rec[g], cur + (size_t)a[g] * REC);
                }
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    REAL xp[NS], sp[H], xn[NS], add[H], lwn;
#pragma unroll
                    for (int d = 0; d < NS; ++d) xp[d] = rec[g][d];
#pragma unroll
                    for (int h = 0; h < H; ++h) sp[h] = rec[g][NS + h];
                    particle_step<MODEL, KERNEL, STAT, REAL>(c, mth, xp, (REAL)y_t, z[g], xn, lwn, add);
#pragma unroll
                    for (int h = 0; h < H; ++h) {
                        const REAL av = use_stat ? add[h] * (REAL)wt : (REAL)0;
                        const REAL sm = (lam * sp[h] + oml * (REAL)S[h]) + av;      // pf.py:175-179 / :78-80
                        sp[h] = is_filter ? av : sm;
                    }
                    if (LWREG) {
                        // register slot j0 + g, selected without dynamic indexing (rolled loop)
                        const REAL keep = v[g] ? lwn : (REAL)(-INFINITY);
#pragma unroll
                        for (int q = 0; q < (LWREG ? CH2 : 1); ++q) lwr[q] = (q == j0 + g) ? keep : lwr[q];
                    }
                    if (v[g]) {
                        if (!LWREG) lwg[i[g]] = lwn;
#pragma unroll
                        for (int d = 0; d < NS; ++d) rec[g][d] = xn[d];
#pragma unroll
                        for (int h = 0; h < H; ++h) rec[g][NS + h] = sp[h];
                        rec_store<REC, REAL>(nxt + (size_t)i[g] * REC, rec[g]);
                        if (P.trace_x) {
                            const size_t row = (size_t)(t + 1) * N + i[g];
                            if (P.trace_anc) P.trace_anc[(size_t)t * N + i[g]] = a[g];
#pragma unroll
                            for (int d = 0; d < NS; ++d) P.trace_x[row * NS + d] = (double)xn[d];
                            P.trace_logw[row] = (double)lwn;
                            if (P.trace_stats && !is_filter) {
#pragma unroll
                                for (int h = 0; h < H; ++h) P.trace_stats[row * H + h] = (double)sp[h];
                            }
                        }
                    }
                }
            }
        };
        if (stat == PFG_STAT_SCORE) sweep(std::integral_constant<int, PFG_STAT_SCORE>{});
        else sweep(std::integral_constant<int, PFG_STAT_SUFF>{});
        { REAL *tmp = cur; cur = nxt; nxt = tmp; }
        wt_prev = wt;
        // children (global stores) become visible to the next step's gathers at its barriers
    }

    // ---- outputs --------------------------------------------------------------------------
    if (tid == 0 && P.out) {
#pragma unroll
        for (int h = 0; h < PFG_MAX_STAT; ++h) P.out[h] = 0.0;
#pragma unroll
        for (int h = 0; h < H; ++h) P.out[h] = is_filter ? filt[h] : S[h];
        P.out[4] = ll; P.out[5] = W; P.out[6] = m; P.out[7] = 1.0;
    }
    if (P.final_x) {
#pragma unroll (LWREG ? CH2 : 1)
        for (int jj = 0; jj < (LWREG ? CH2 : MEM_MAX_CHUNKS); ++jj) {
            const int i = jj * NT + tid;
            if (i >= N) break;
#pragma unroll
            for (int d = 0; d < NS; ++d) P.final_x[(size_t)i * NS + d] = (double)cur[(size_t)i * REC + d];
            if (P.final_logw) P.final_logw[i] = (double)(LWREG ? lwr[LWREG ? jj : 0] : lwg[i]);
            if (P.final_stats && !is_filter) {
#pragma unroll
                for (int h = 0; h < H; ++h) P.final_stats[(size_t)i * H + h] = (double)cur[(size_t)i * REC + NS + h];
            }
        }
    }
}

}  // namespace pfg
