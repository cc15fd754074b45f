// libpfgrad host code: NumPy's legacy RandomState streams (MT19937 + polar Gaussians), bit-identical,
// generated natively.  The REPLAY path of the drop-in Sampler API must hand the kernel exactly the
// numbers the reference's particle filter takes from the global `np.random` state -- per window
//     z0 = normal(size=N);  for t in range(T):  u[t] = random_sample(N);  z[t] = normal(size=N)
// (particle_filters/pf.py:26-38 via np.random.choice / Kernel.rv; SURVEY finding 1) -- and drawing
// them row by row through NumPy cost 14 of the 16.9 ms of a T = N = 1000 SGLD step.  Here the MT19937
// word stream and the polar method's accept/reject decisions run in one tight sequential loop, and the
// expensive part of every accepted pair, sqrt(-2 log(r2) / r2) with the host libm (the same functions
// NumPy calls), is spread over worker threads.  The generator state is taken from / returned to
// RandomState.get_state() / set_state(), so everything else keeps consuming `np.random` as before.
// Algorithms restated from their published definitions: MT19937 (Matsumoto & Nishimura 1998), the
// 53-bit double (a >> 5, b >> 6), Marsaglia's polar method with the second variate cached.
// Host-only translation unit, compiled with -ffp-contract=off (x1*x1 + x2*x2 must not be fused).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "pfgrad.h"

namespace {

// MT19937 block step: 624 new state words, then their tempered outputs.  Written as plain loops over
// independent elements (the recurrence reaches back 227 / 397 words, further than any vector is wide) so
// that the compiler vectorises them; the AVX2 clone is picked at run time when the CPU has it.
#define PFG_MT_BLOCK_BODY                                                                       \
    constexpr uint32_t UP = 0x80000000u, LO = 0x7fffffffu, A = 0x9908b0dfu;                    \
    for (int kk = 0; kk < 624 - 397; ++kk) {                                                    \
        const uint32_t y = (mt[kk] & UP) | (mt[kk + 1] & LO);                                   \
        mt[kk] = mt[kk + 397] ^ (y >> 1) ^ ((0u - (y & 1u)) & A);                               \
    }                                                                                           \
    for (int kk = 624 - 397; kk < 623; ++kk) {                                                  \
        const uint32_t y = (mt[kk] & UP) | (mt[kk + 1] & LO);                                   \
        mt[kk] = mt[kk - 227] ^ (y >> 1) ^ ((0u - (y & 1u)) & A);                               \
    }                                                                                           \
    {                                                                                           \
        const uint32_t y = (mt[623] & UP) | (mt[0] & LO);                                       \
        mt[623] = mt[396] ^ (y >> 1) ^ ((0u - (y & 1u)) & A);                                   \
    }
#define PFG_MT_TEMPER_BODY                                                                      \
    for (int kk = from; kk < 624; ++kk) {                                                       \
        uint32_t y = mt[kk];                                                                    \
        y ^= (y >> 11);                                                                         \
        y ^= (y << 7) & 0x9d2c5680u;                                                            \
        y ^= (y << 15) & 0xefc60000u;                                                           \
        y ^= (y >> 18);                                                                         \
        out[kk] = y;                                                                            \
    }
void mt_block_generic(uint32_t *__restrict__ mt) { PFG_MT_BLOCK_BODY }
void mt_temper_generic(const uint32_t *__restrict__ mt, uint32_t *__restrict__ out, int from) { PFG_MT_TEMPER_BODY }
__attribute__((target("avx2"))) void mt_block_avx2(uint32_t *__restrict__ mt) { PFG_MT_BLOCK_BODY }
__attribute__((target("avx2"))) void mt_temper_avx2(const uint32_t *__restrict__ mt, uint32_t *__restrict__ out, int from) { PFG_MT_TEMPER_BODY }

// bulk conversions straight from the tempered block (vectorised): doubles from word pairs, and the polar
// method's candidate attempts (4 words each).  Exactly the scalar formulas, element by element.
#define PFG_DOUBLES_BODY                                                                        \
    for (int i = 0; i < n; ++i) {                                                               \
        const uint32_t a = w[2 * i] >> 5, b = w[2 * i + 1] >> 6;                                \
        out[i] = ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;                     \
    }
#define PFG_CANDIDATES_BODY                                                                     \
    for (int i = 0; i < n; ++i) {                                                               \
        const uint32_t a0 = w[4 * i] >> 5, b0 = w[4 * i + 1] >> 6;                              \
        const uint32_t a1 = w[4 * i + 2] >> 5, b1 = w[4 * i + 3] >> 6;                          \
        const double d0 = ((double)a0 * 67108864.0 + (double)b0) / 9007199254740992.0;          \
        const double d1 = ((double)a1 * 67108864.0 + (double)b1) / 9007199254740992.0;          \
        const double v1 = 2.0 * d0 - 1.0, v2 = 2.0 * d1 - 1.0;                                  \
        x1[i] = v1; x2[i] = v2; r2[i] = v1 * v1 + v2 * v2;                                      \
    }
void doubles_generic(const uint32_t *__restrict__ w, int n, double *__restrict__ out) { PFG_DOUBLES_BODY }
__attribute__((target("avx2"))) void doubles_avx2(const uint32_t *__restrict__ w, int n, double *__restrict__ out) { PFG_DOUBLES_BODY }
void candidates_generic(const uint32_t *__restrict__ w, int n, double *__restrict__ x1, double *__restrict__ x2, double *__restrict__ r2) { PFG_CANDIDATES_BODY }
__attribute__((target("avx2"))) void candidates_avx2(const uint32_t *__restrict__ w, int n, double *__restrict__ x1, double *__restrict__ x2, double *__restrict__ r2) { PFG_CANDIDATES_BODY }

struct MT {
    uint32_t *key;          // the RandomState's state words (untempered), advanced block by block
    int pos;
    uint32_t out[624];      // tempered outputs of the current block
    bool avx2;
    void init() {
        avx2 = __builtin_cpu_supports("avx2");
        if (pos < 624) { if (avx2) mt_temper_avx2(key, out, pos); else mt_temper_generic(key, out, pos); }
    }
    void refill() {
        if (avx2) { mt_block_avx2(key); mt_temper_avx2(key, out, 0); }
        else { mt_block_generic(key); mt_temper_generic(key, out, 0); }
        pos = 0;
    }
    inline uint32_t next32() {
        if (pos >= 624) refill();
        return out[pos++];
    }
    inline double next_double() {
        const uint32_t a = next32() >> 5, b = next32() >> 6;
        return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
    }
};

// one accepted attempt of the polar method, transform still to do
struct Pair { double x1, x2, r2; };

inline void transform(const Pair &p, double &first, double &second) {
    const double f = std::sqrt(-2.0 * std::log(p.r2) / p.r2);
    first = f * p.x2;          // returned by this call
    second = f * p.x1;         // cached for the next call
}

}  // namespace

extern "C" int pfg_legacy_streams(uint32_t *key, int32_t *pos, int32_t *has_gauss, double *gauss, int N, int T,
                                  double *z0, double *u, double *z, int threads) {
    if (!key || !pos || !has_gauss || !gauss || N < 1 || T < 0 || !z0 || (T > 0 && (!u || !z))) return PFG_ERR_INVALID;
    if (*pos < 0 || *pos > 624) return PFG_ERR_INVALID;
    MT mt;
    mt.key = key; mt.pos = *pos;
    mt.init();
    // normals are consumed row after row: z0, z[0], z[1], ... with the cached second variate carried
    // across rows.  dst[k] is the k-th normal overall; pair q fills dst[2q + off], dst[2q + 1 + off].
    const size_t total = (size_t)N * ((size_t)T + 1);
    std::vector<Pair> pairs;
    try { pairs.resize(total / 2 + 2); } catch (...) { return PFG_ERR_NOMEM; }
    auto dst = [&](size_t k) -> double * { return k < (size_t)N ? z0 + k : z + (k - N); };
    size_t k = 0, npairs = 0;
    const bool lead_cached = *has_gauss != 0;       // the first normal comes from the cache
    if (lead_cached) { *dst(0) = *gauss; k = 1; }
    for (int row = 0; row <= T; ++row) {
        const size_t row_end = (size_t)N * (row + 1);
        if (row > 0) {                                // uniforms of timestep row - 1 come BEFORE its normals
            double *ur = u + (size_t)(row - 1) * N;
            int i = 0;
            while (i < N) {
                const int fit = (624 - mt.pos) / 2;       // whole doubles left in the current block
                if (fit < 1) { ur[i++] = mt.next_double(); continue; }     // straddles a block boundary (or refill)
                const int n = fit < N - i ? fit : N - i;
                if (mt.avx2) doubles_avx2(mt.out + mt.pos, n, ur + i); else doubles_generic(mt.out + mt.pos, n, ur + i);
                mt.pos += 2 * n;
                i += n;
            }
        }
        // attempts until this row's normals are covered (a pair started in this row may spill one
        // variate into the next row: the cache)
        const size_t have = (lead_cached ? 1 : 0) + 2 * npairs;
        size_t need_pairs = row_end > have ? (row_end - have + 1) / 2 : 0;
        while (need_pairs > 0) {
            const int avail = (624 - mt.pos) / 4;         // whole attempts left in the current block
            if (avail < 1) {                              // an attempt that straddles the block boundary (or a refill)
                const double x1 = 2.0 * mt.next_double() - 1.0;
                const double x2 = 2.0 * mt.next_double() - 1.0;
                const double r2 = x1 * x1 + x2 * x2;
                if (!(r2 >= 1.0 || r2 == 0.0)) { pairs[npairs++] = Pair{x1, x2, r2}; --need_pairs; }
                continue;
            }
            double cx1[156], cx2[156], cr2[156];
            if (mt.avx2) candidates_avx2(mt.out + mt.pos, avail, cx1, cx2, cr2);
            else candidates_generic(mt.out + mt.pos, avail, cx1, cx2, cr2);
            int used = 0;
            size_t got = 0;
            while (used < avail && got < need_pairs) {    // in order, stop at the last attempt this row consumes
                const double r2 = cr2[used];
                pairs[npairs + got] = Pair{cx1[used], cx2[used], r2};      // branch-free: a rejected attempt is overwritten
                got += (r2 >= 1.0 || r2 == 0.0) ? 0 : 1;
                ++used;
            }
            npairs += got;
            need_pairs -= got;
            mt.pos += 4 * used;
        }
    }
    // transforms: independent per pair -> worker threads
    const size_t off = lead_cached ? 1 : 0;
    double spill = 0.0;
    bool spilled = false;
    auto work = [&](size_t q0, size_t q1) {
        for (size_t q = q0; q < q1; ++q) {
            double a, b;
            transform(pairs[q], a, b);
            const size_t ka = 2 * q + off, kb = ka + 1;
            *dst(ka) = a;
            if (kb < total) *dst(kb) = b;
            else { spill = b; spilled = true; }       // only the last pair can spill (one writer)
        }
    };
    // measured on the MI355X box's host (T = N = 1000): 4.4 / 3.5 / 4.0 / 3.8 ms with 1 / 2 / 4 / 8 threads
    // (the sequential word + accept stage dominates; NumPy's own calls: 13.6 ms)
    int nt = threads > 0 ? threads : 2;
    if (nt > 16) nt = 16;
    if (npairs < 100000) nt = 1;
    if (nt == 1) {
        work(0, npairs);
    } else {
        std::vector<std::thread> pool;
        const size_t chunk = (npairs + nt - 1) / nt;
        for (int w = 0; w < nt; ++w) {
            const size_t a = (size_t)w * chunk, b = a + chunk < npairs ? a + chunk : npairs;
            if (a < b) pool.emplace_back(work, a, b);
        }
        for (auto &th : pool) th.join();
    }
    (void)k;
    *pos = mt.pos;
    *has_gauss = spilled ? 1 : 0;
    *gauss = spilled ? spill : 0.0;
    return PFG_OK;
}
