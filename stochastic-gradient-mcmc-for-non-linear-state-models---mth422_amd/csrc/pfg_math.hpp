// libpfgrad device code: wave primitives (DPP), Philox / lane generators, LDS-table fp64 math.
// Part of pfg_device.hpp (see there for the overview).
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

// Pointers that arrive inside a descriptor (pfg_dev_problem) are generic to the compiler, and a generic access is a FLAT
// instruction: it counts in lgkmcnt as well as vmcnt and may return out of order with LDS operations, so EVERY wait for
// an LDS result (and every barrier) also waits for the flat loads in flight -- a load issued a timestep ahead is waited
// for at the next ds_read.  Everything the descriptors point to is device memory (pfg_launch_device's contract): saying
// so (pointers typed address_space(1)) turns the accesses
// into global_load / global_store, which only count in vmcnt.
template <class T> using gptr = T __attribute__((address_space(1))) *;
template <class T> __device__ __forceinline__ gptr<T> global_ptr(T *p) { return (gptr<T>)p; }

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
// the same in f32: one v_add_f32 with a DPP operand per step (6 instructions; the f64 form moves two halves per step)
template <int CTRL>
__device__ __forceinline__ float dpp_shr0_f32(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_incl_scan_f32(float v) {
    v += dpp_shr0_f32<0x111>(v);
    v += dpp_shr0_f32<0x112>(v);
    v += dpp_shr0_f32<0x114>(v);
    v += dpp_shr0_f32<0x118>(v);
    v += dpp_f32<0x142, 0xa>(0.0f, v);
    v += dpp_f32<0x143, 0xc>(0.0f, v);
    return v;
}

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
    // one v_max_f32 with a DPP operand per step (the compiler lowers the builtin form to
    // mov + mov_dpp + cmp + cndmask: 36 instructions instead of 6).  A lane whose DPP source does not
    // exist (bound_ctrl off) or whose row is masked keeps its value; the s_nop 1 are the two wait
    // states between a VALU write and a DPP read of the same register.
    asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
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
// Device RNG (PFG_RNG_DEVICE): one small generator per lane, its 128-bit state keyed by
// Philox4x32-10(seed; lane, stream = global chain id, step counter), so streams are reproducible and independent
// of how chains are spread over GPUs.  Default: jsf32 (Jenkins' small fast generator, two-rotate form 27 / 17:
// adds, xors and rotates only -- 32-bit multiplies are quarter-rate on CDNA --, 7 instructions per word, eight
// warm-up rounds behind the key); -DPFG_RNG_XOSHIRO=1 builds xoshiro128++ (Blackman & Vigna 2019, 11 instructions per
// word) instead.  Round 2 measured no difference between the two (15.74 vs 15.78 ms per launch: the kernel then
// stalled elsewhere); with the LDS stalls of round 3 gone the shorter generator is worth 1.3-2.8 % on every config
// (ms per bench launch, xoshiro / jsf32: SVM 46.8 / 45.5, N = 4000 12.23 / 11.92, GARCH windows 2.71 / 2.67, one wave
// 1.73 / 1.70, N = 10000 6.79 / 6.70; profiles/r03_ab_jsf32.txt).  The draws are inputs of the filter; their
// distribution is tested on what the kernels record (tests/test_gpu_device_replay.py::test_recorded_draws_are_standard,
// tests/test_gpu_ensemble.py::test_device_generator_normals_and_uniform_streams, ::test_device_generator_large_sample;
// 2e9 draws: profiles/r03_generator_tails.txt).
// ------------------------------------------------------------------------------------
struct LaneRng {
    uint32_t s0, s1, s2, s3;
#ifdef PFG_RNG_XOSHIRO
    __device__ __forceinline__ uint32_t next() {        // xoshiro128++: 11 instructions per word
        const uint32_t sum = s0 + s3;
        const uint32_t result = ((sum << 7) | (sum >> 25)) + s0;
        const uint32_t t = s1 << 9;
        s2 ^= s0; s3 ^= s1; s1 ^= s2; s0 ^= s3;
        s2 ^= t;
        s3 = (s3 << 11) | (s3 >> 21);
        return result;
    }
#else
    __device__ __forceinline__ uint32_t next() {        // jsf32: 7 instructions per word
        const uint32_t e = s0 - ((s1 << 27) | (s1 >> 5));
        s0 = s1 ^ ((s2 << 17) | (s2 >> 15));
        s1 = s2 + s3;
        s2 = s3 + e;
        s3 = e + s0;
        return s3;
    }
#endif
};

__device__ __forceinline__ LaneRng lane_rng_init(uint64_t seed, uint64_t stream, uint64_t step, uint32_t lane) {
    u32x4 r = philox4x32_10({lane, (uint32_t)step, (uint32_t)stream,
                             (uint32_t)(stream >> 32) ^ (uint32_t)(step >> 32)},
                            (uint32_t)seed, (uint32_t)(seed >> 32));
    LaneRng g;
    g.s0 = r.x; g.s1 = r.y; g.s2 = r.z; g.s3 = r.w | 1u;   // never the all-zero state (a fixed point of both generators)
#ifndef PFG_RNG_XOSHIRO
#pragma unroll
    for (int q = 0; q < 8; ++q) (void)g.next();             // jsf32: mix the key into all four words
#endif
    return g;
}

// exponential spacing -log(u), u in (0, 1) with 24 random bits, on the f32 log unit (pf_big_kernel's sorted
// resampling uniforms; an input of the filter like the Box-Muller normals)
__device__ __forceinline__ float spacing_f32(uint32_t w) {
    return -0.6931471805599453f * __builtin_amdgcn_logf(((float)(w >> 8) + 0.5f) * (1.0f / 16777216.0f));   // raw v_log_f32: the argument is a normal number
}

// f64 -> u32, saturating at both ends (what v_cvt_u32_f64 does; a C cast is undefined out of range)
__device__ __forceinline__ uint32_t cvt_u32_sat(double v) {
    uint32_t r;
    asm("v_cvt_u32_f64 %0, %1" : "=v"(r) : "v"(v));
    return r;
}

// uniform in (0,1) with 32 random bits (resampling needs resolution << 1/N only)
__device__ __forceinline__ double u01_32(uint32_t a) { return ((double)a + 0.5) * (1.0 / 4294967296.0); }

// ------------------------------------------------------------------------------------
// fp64 elementary functions on small LDS tables.  ocml's exp / log / sincospi cost 42 / 98 / 70
// VALU instructions each (half of them re-materialising polynomial coefficients); the table
// forms below need 17 / 20 / 17 and one LDS read, at <= 2 ulp -- well inside the parity
// tolerance.  Tables are filled once per workgroup with ocml.  Explicit fma(): the file is
// compiled with -ffp-contract=off.
//   e2[j] = 2^(j/TAB_E2)                                j < TAB_E2 (128; 32 in the device-generator units)
//   lg[j] = {1/c_j, log c_j},  c_j = 1 + (j+0.5)/128    j < 128
//   sc[j] = {sin, cos}(2 pi (j+0.5)/256)                j < 256
// A/B (-DPFG_TAB_E2_FAST=32, device-generator units): a 32-entry 2^(j/32) table is exactly the 64 LDS banks, so a
// ds_read_b64 with any 64 indices is conflict-free, at the price of one more polynomial term.  Measured (ms per bench
// launch, 32 / 128 entries): SVM 47.45 / 46.97, GARCH windows 2.66 / 2.71, N = 4000 12.62 / 12.30, one wave 1.74 / 1.73,
// N = 10000 6.82 / 6.77 -- the table reads are not where the LDS cycles go; 128 stays.
// ------------------------------------------------------------------------------------
constexpr int TAB_E2_ACC = 128, TAB_E2_FAST = 32;
#ifdef PFG_FAST_ALGEBRA
#ifndef PFG_TAB_E2_FAST
#define PFG_TAB_E2_FAST 128
#endif
constexpr int TAB_E2 = PFG_TAB_E2_FAST;
#else
constexpr int TAB_E2 = TAB_E2_ACC;
#endif
constexpr int TAB_LG = 128, TAB_SC = 0;     // no sin/cos table: see Math<double,true>::normal_pair
constexpr int TAB_DOUBLES_EXP = TAB_E2 + 2 * TAB_LG, TAB_DOUBLES_RNG = 2 * TAB_SC;   // exp+log always; sincos with the device RNG
// table doubles of a kernel of the device-generator (fast) / REPLAY units, for host code compiled with other flags
__host__ __device__ constexpr int tab_doubles_exp(bool fast) { return (fast ? 128 : TAB_E2_ACC) + 2 * TAB_LG; }

struct TabF64 {
    const double *e2;
    const double2 *lg;
    const double2 *sc;
};

__device__ inline void tab_fill(double *mem, bool with_rng, int tid, int nthreads) {
    double *lg = mem + TAB_E2, *sc = lg + 2 * TAB_LG;
    for (int j = tid; j < TAB_E2; j += nthreads) mem[j] = exp2((double)j * (1.0 / TAB_E2));
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

// exp(x), any x (overflow -> inf, underflow -> 0, -inf -> 0).  FINITE = the caller guarantees a
// finite argument: no clamp (a huge negative x still ends in ldexp's underflow to 0).
template <bool FINITE = false>
__device__ __forceinline__ double exp_tab(double x, const double *__restrict__ e2) {
    if (!FINITE) x = fmax(x, -1000.0);
#if defined(PFG_FAST_ALGEBRA) && PFG_TAB_E2_FAST == 32
    // device-generator units: one-step reduction to |r| <= ln2/64 and a quartic for expm1 -- relative error
    // r^5/120 < 1.3e-12, far below the Monte-Carlo noise these kernels carry; the REPLAY units keep the <= 2 ulp form
    const double kd = rint(x * 46.16624130844683);                   // 32/ln2
    const int k = (int)kd;
    const double r = fma(kd, -0.021660849392498290, x);              // ln2/32
    const double t = e2[k & (TAB_E2 - 1)];
    double p = fma(r, 0.041666666666666664, 0.16666666666666666);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = p * r;
    return ldexp(fma(t, p, t), k >> 5);
#elif defined(PFG_FAST_ALGEBRA)
    const double kd = rint(x * 184.6649652337873);                  // 128/ln2
    const int k = (int)kd;
    // 128-entry table (-DPFG_TAB_E2_FAST=128, A/B): one-step reduction and a cubic for expm1 -- relative error < 3e-12
    // (|r| <= ln2/256: r^4/24 = 2e-12)
    const double r = fma(kd, -0.0054152123481245725, x);             // ln2/128
    const double t = e2[k & (TAB_E2 - 1)];
    double p = fma(r, 0.16666666666666666, 0.5);
    p = fma(p, r, 1.0);
    p = p * r;
    return ldexp(fma(t, p, t), k >> 7);
#else
    const double kd = rint(x * 184.6649652337873);                  // 128/ln2
    const int k = (int)kd;
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

// the <= 2 ulp form whatever the unit's flags (the whole-GPU window builds its resampling CDF with it also in the
// device-generator units: at N = 10^5 .. 10^6 the 3e-12 of the cubic form above would flip ancestors between the kernel
// and the oracle that replays its recorded draws -- 2 N^2 delta per step)
__device__ __forceinline__ double exp_tab_acc(double x, const double *__restrict__ e2) {
    static_assert(TAB_E2 == 128, "exp_tab_acc uses the 128-entry table");
    x = fmax(x, -1000.0);
    const double kd = rint(x * 184.6649652337873);                  // 128/ln2
    const int k = (int)kd;
    double r = fma(kd, -0.00541521234663378, x);                     // ln2/128, 32-bit head
    r = fma(kd, -1.4907929134926466e-12, r);                         //          tail
    const double t = e2[k & 127];
    double p = fma(r, 0.008333333333333333, 0.041666666666666664);   // expm1(r), |r| <= ln2/256
    p = fma(p, r, 0.16666666666666666);
    p = fma(p, r, 0.5);
    p = p * r;
    p = fma(p, r, r);
    return ldexp(fma(t, p, t), k >> 7);
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

// 1/x for finite x > 0 (no special cases): v_rcp_f64 + two Newton steps (< 1 ulp) instead of the IEEE division
// sequence (div_scale x2, rcp, five fma, div_fmas, div_fixup)
__device__ __forceinline__ double rcp_pos(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return fma(fma(-x, r, 1.0), r, r);
}

template <typename REAL, bool TAB> struct Math;
template <> struct Math<double, true> {
    TabF64 t;
    __device__ __forceinline__ double exp(double x) const { return exp_tab(x, t.e2); }
    __device__ __forceinline__ double exp_acc(double x) const { return exp_tab_acc(x, t.e2); }
#ifdef PFG_FAST_ALGEBRA
    __device__ __forceinline__ double exp_finite(double x) const { return exp_tab<true>(x, t.e2); }
#else
    __device__ __forceinline__ double exp_finite(double x) const { return exp_tab(x, t.e2); }
#endif
    __device__ __forceinline__ double log(double x) const { return log_tab(x, t.lg); }
    __device__ __forceinline__ double sqrt(double x) const { return ::sqrt(x); }
    // finite positive arguments only (device-generator units): no special-case handling
    __device__ __forceinline__ double sqrt_pos(double x) const { return pfg::sqrt_pos(x); }
    __device__ __forceinline__ double rcp_pos(double x) const { return pfg::rcp_pos(x); }
    // two independent standard normals from two words (Box-Muller, both branches).  The draws
    // are INPUTS of the filter, like the 32-bit uniforms: they are generated with the f32
    // transcendental units (v_log / v_sin / v_cos: ~12 issue slots per normal instead of ~25 for
    // a table-based fp64 evaluation) and widened; all arithmetic on the state stays fp64.
    // u1 keeps its full exponent range ((a + 0.5) 2^-32: |z| up to 6.7), the angle has 24 bits.
    __device__ __forceinline__ void normal_pair_f32(uint32_t a, uint32_t b, float &z0, float &z1) const {
        const float u1 = ((float)a + 0.5f) * 2.3283064365386963e-10f;       // (0, 1]
        const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);             // [0,1): angle / 2pi
        // raw v_log_f32 / v_sqrt_f32 (1 ulp each; u1 >= 2^-33 is a normal number, the radicand is in [0, 46]):
        // the library logf / sqrtf add a denormal rescue, an extended-precision ln 2 product and a
        // correctly-rounded-sqrt fix-up, ~20 instructions per pair that a 24-bit normal has no use for
        const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
        z0 = r * __builtin_amdgcn_cosf(u2);
        z1 = r * __builtin_amdgcn_sinf(u2);
    }
    __device__ __forceinline__ void normal_pair(uint32_t a, uint32_t b, double &z0, double &z1) const {
        float f0, f1;
        normal_pair_f32(a, b, f0, f1);
        z0 = (double)f0; z1 = (double)f1;
    }
};
template <> struct Math<double, false> {
    TabF64 t;
    __device__ __forceinline__ double exp(double x) const { return ::exp(x); }
    __device__ __forceinline__ double exp_acc(double x) const { return ::exp(x); }
    __device__ __forceinline__ double exp_finite(double x) const { return ::exp(x); }
    __device__ __forceinline__ double log(double x) const { return ::log(x); }
    __device__ __forceinline__ double sqrt(double x) const { return ::sqrt(x); }
    __device__ __forceinline__ double sqrt_pos(double x) const { return ::sqrt(x); }
    __device__ __forceinline__ double rcp_pos(double x) const { return 1.0 / x; }
    __device__ __forceinline__ void normal_pair(uint32_t a, uint32_t b, double &z0, double &z1) const {
        const double u1 = ((double)a + 0.5) * (1.0 / 4294967296.0);
        const double r = ::sqrt(-2.0 * ::log(u1));
        double sn, cs;
        sincospi((double)b * (1.0 / 2147483648.0), &sn, &cs);
        z0 = r * cs; z1 = r * sn;
    }
    __device__ __forceinline__ void normal_pair_f32(uint32_t a, uint32_t b, float &z0, float &z1) const {
        double d0, d1;
        normal_pair(a, b, d0, d1);
        z0 = (float)d0; z1 = (float)d1;
    }
};
template <bool TAB> struct Math<float, TAB> {
    TabF64 t;
    __device__ __forceinline__ float exp(float x) const { return __expf(x); }
    __device__ __forceinline__ float exp_acc(float x) const { return __expf(x); }
    __device__ __forceinline__ float exp_finite(float x) const { return __expf(x); }
    __device__ __forceinline__ float log(float x) const { return __logf(x); }
    __device__ __forceinline__ float sqrt(float x) const { return sqrtf(x); }
    __device__ __forceinline__ float sqrt_pos(float x) const { return sqrtf(x); }
    __device__ __forceinline__ float rcp_pos(float x) const { return 1.0f / x; }
    __device__ __forceinline__ void normal_pair(uint32_t a, uint32_t b, float &z0, float &z1) const {
        const float u1 = ((float)(a >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0,1), 24 bits
        const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);           // [0,1): angle / 2pi
        const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));   // raw units, see Math<double, true>
        // v_sin_f32 / v_cos_f32 take their argument in revolutions
        z0 = r * __builtin_amdgcn_cosf(u2); z1 = r * __builtin_amdgcn_sinf(u2);
    }
    __device__ __forceinline__ void normal_pair_f32(uint32_t a, uint32_t b, float &z0, float &z1) const {
        normal_pair(a, b, z0, z1);
    }
};

// bytes of LDS math tables a kernel instantiation carries
template <typename REAL, int RNG, bool TAB>
__host__ __device__ constexpr size_t tab_bytes() {
    return (TAB && sizeof(REAL) == 8) ? (size_t)8 * (TAB_DOUBLES_EXP + (RNG == PFG_RNG_DEVICE ? TAB_DOUBLES_RNG : 0)) : 0;
}

}  // namespace pfg
