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
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <pthread.h>
#include <sched.h>
#include <cmath>
#include <immintrin.h>
#include <memory>
#include <cstdint>
#include <cstring>
#include <mutex>
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

// one accepted attempt of the polar method: (x1, x2, r2), transform still to do
inline void transform(double x1, double x2, double r2, double &first, double &second) {
    const double f = std::sqrt(-2.0 * std::log(r2) / r2);
    first = f * x2;          // returned by this call
    second = f * x1;         // cached for the next call
}

// Acceptance, in order: copy the attempts with 0 < r2 < 1 to the front of (o1, o2, o3), stop after `want`
// accepted ones.  Returns the number of attempts consumed; `got` = accepted.  The destination needs 4 doubles
// of slack behind the last accepted entry (vector stores).
inline int accept_generic(const double *c1, const double *c2, const double *c3, int n, size_t want,
                          double *o1, double *o2, double *o3, size_t &got) {
    int used = 0;
    got = 0;
    while (used < n && got < want) {                      // branch-free: a rejected attempt is overwritten
        const double r2 = c3[used];
        o1[got] = c1[used]; o2[got] = c2[used]; o3[got] = r2;
        got += (r2 >= 1.0 || r2 == 0.0) ? 0 : 1;
        ++used;
    }
    return used;
}

__attribute__((target("avx2"))) int accept_avx2(const double *c1, const double *c2, const double *c3, int n, size_t want,
                                                double *o1, double *o2, double *o3, size_t &got) {
    // left-pack permutation (pairs of 32-bit lanes) for every 4-bit acceptance mask
    alignas(32) static const int32_t LUT[16][8] = {
        {0, 1, 2, 3, 4, 5, 6, 7}, {0, 1, 2, 3, 4, 5, 6, 7}, {2, 3, 0, 1, 4, 5, 6, 7}, {0, 1, 2, 3, 4, 5, 6, 7},
        {4, 5, 0, 1, 2, 3, 6, 7}, {0, 1, 4, 5, 2, 3, 6, 7}, {2, 3, 4, 5, 0, 1, 6, 7}, {0, 1, 2, 3, 4, 5, 6, 7},
        {6, 7, 0, 1, 2, 3, 4, 5}, {0, 1, 6, 7, 2, 3, 4, 5}, {2, 3, 6, 7, 0, 1, 4, 5}, {0, 1, 2, 3, 6, 7, 4, 5},
        {4, 5, 6, 7, 0, 1, 2, 3}, {0, 1, 4, 5, 6, 7, 2, 3}, {2, 3, 4, 5, 6, 7, 0, 1}, {0, 1, 2, 3, 4, 5, 6, 7}};
    const __m256d one = _mm256_set1_pd(1.0), zero = _mm256_setzero_pd();
    int used = 0;
    got = 0;
    // whole groups of four while even four acceptances cannot overshoot `want`
    while (used + 4 <= n && got + 4 <= want) {
        const __m256d r = _mm256_loadu_pd(c3 + used);
        // accepted <=> !(r2 >= 1.0 || r2 == 0.0)  (a NaN cannot occur: r2 is a sum of squares of finite numbers)
        const __m256d rej = _mm256_or_pd(_mm256_cmp_pd(r, one, _CMP_GE_OQ), _mm256_cmp_pd(r, zero, _CMP_EQ_OQ));
        const int m = (~_mm256_movemask_pd(rej)) & 15;
        const __m256i idx = _mm256_load_si256(reinterpret_cast<const __m256i *>(LUT[m]));
        _mm256_storeu_pd(o1 + got, _mm256_castsi256_pd(_mm256_permutevar8x32_epi32(_mm256_castpd_si256(_mm256_loadu_pd(c1 + used)), idx)));
        _mm256_storeu_pd(o2 + got, _mm256_castsi256_pd(_mm256_permutevar8x32_epi32(_mm256_castpd_si256(_mm256_loadu_pd(c2 + used)), idx)));
        _mm256_storeu_pd(o3 + got, _mm256_castsi256_pd(_mm256_permutevar8x32_epi32(_mm256_castpd_si256(r), idx)));
        got += (size_t)__builtin_popcount((unsigned)m);
        used += 4;
    }
    size_t tail = 0;
    used += accept_generic(c1 + used, c2 + used, c3 + used, n - used, want - got, o1 + got, o2 + got, o3 + got, tail);
    got += tail;
    return used;
}

// Copy n doubles to a destination nobody reads again on this core (the stream arrays are consumed by the GPU's
// DMA engine): whole cache lines go out with non-temporal stores, so the 16 MB of a T = N = 1000 window are
// written once instead of read-for-ownership and written.
__attribute__((target("avx2"))) void stream_copy_avx2(double *__restrict__ dst, const double *__restrict__ src, size_t n) {
    size_t i = 0;
    while (i < n && (reinterpret_cast<uintptr_t>(dst + i) & 63u)) { dst[i] = src[i]; ++i; }
    for (; i + 8 <= n; i += 8) {
        _mm256_stream_pd(dst + i, _mm256_loadu_pd(src + i));
        _mm256_stream_pd(dst + i + 4, _mm256_loadu_pd(src + i + 4));
    }
    for (; i < n; ++i) dst[i] = src[i];
}
inline void stream_copy(bool avx2, double *dst, const double *src, size_t n) {
    if (avx2) stream_copy_avx2(dst, src, n); else std::memcpy(dst, src, n * sizeof(double));
}

// ---- where the transform workers run -------------------------------------------------------------------------
// The accepted attempts travel from the sequential stage to a worker through cache lines; when the scheduler puts the
// worker on the other socket (the GPU box's host: 2 x 64 cores, cgroup quota, no cpuset) every line of the ring costs a
// cross-socket transfer and the SEQUENTIAL stage slows down (measured, T = N = 1000, two threads: 2.9 ms unplaced,
// 2.0-2.2 ms with the worker on a core that shares the caller's L3; one thread: 3.8 ms).  So the short-lived worker
// threads -- never the caller's thread -- are placed on cores that share the L3 of the CPU the caller is running on
// right now: one hardware thread per core, the caller's own core left out, only CPUs the process may use.
// PFGRAD_RNG_PIN=0 leaves placement to the scheduler.
std::vector<int> parse_cpu_list(const char *path) {
    std::vector<int> out;
    FILE *f = std::fopen(path, "r");
    if (!f) return out;
    char buf[4096];
    if (std::fgets(buf, sizeof buf, f)) {
        const char *p = buf;
        while (*p) {
            char *e;
            const long a = std::strtol(p, &e, 10);
            if (e == p) break;
            long b = a;
            p = e;
            if (*p == '-') { b = std::strtol(p + 1, &e, 10); p = e; }
            for (long c = a; c <= b && out.size() < 4096; ++c) out.push_back((int)c);
            if (*p == ',') ++p; else break;
        }
    }
    std::fclose(f);
    return out;
}

std::vector<int> cores_near(int cpu) {
    static std::mutex mu;
    static std::vector<std::pair<int, std::vector<int>>> cache;
    std::lock_guard<std::mutex> lock(mu);
    for (const auto &c : cache) if (c.first == cpu) return c.second;
    std::vector<int> out;
    char path[128];
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (cpu >= 0 && sched_getaffinity(0, sizeof allowed, &allowed) == 0) {
        std::snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
        const std::vector<int> l3 = parse_cpu_list(path);
        std::snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", cpu);
        const std::vector<int> mine = parse_cpu_list(path);
        for (int c : l3) {
            if (c < 0 || c >= CPU_SETSIZE || !CPU_ISSET(c, &allowed)) continue;
            bool own = c == cpu;
            for (int m : mine) own = own || m == c;
            if (own) continue;
            std::snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list", c);
            const std::vector<int> sib = parse_cpu_list(path);
            bool first = true;                            // one hardware thread per core: the lowest-numbered sibling
            for (int sc : sib) first = first && sc >= c;
            if (first) out.push_back(c);
        }
    }
    if (cache.size() < 512) cache.emplace_back(cpu, out);
    return out;
}

}  // namespace

extern "C" int pfg_legacy_streams(uint32_t *key, int32_t *pos, int32_t *has_gauss, double *gauss, int N, int T,
                                  double *z0, double *u, double *z, int threads) {
    if (!key || !pos || !has_gauss || !gauss || N < 1 || T < 0 || !z0 || (T > 0 && (!u || !z))) return PFG_ERR_INVALID;
    if (*pos < 0 || *pos > 624) return PFG_ERR_INVALID;
    MT mt;
    mt.key = key; mt.pos = *pos;
    mt.init();
    const bool avx2 = mt.avx2;
    // normals are consumed row after row: z0, z[0], z[1], ... with the cached second variate carried
    // across rows.  dst[k] is the k-th normal overall; pair q fills dst[2q + off], dst[2q + 1 + off].
    const size_t total = (size_t)N * ((size_t)T + 1);
    const bool lead_cached = *has_gauss != 0;       // the first normal comes from the cache
    if (lead_cached) z0[0] = *gauss;
    const size_t off = lead_cached ? 1 : 0;
    double spill = 0.0;
    bool spilled = false;

    // The accepted attempts of a SLICE of rows wait in a small ring of buffers (cache resident) for their
    // transform -- sqrt, log, divide per pair with the host libm NumPy calls -- which runs on worker threads
    // behind the sequential word / acceptance loop (threads = 1: on this thread, slice by slice).
    // Measured on the MI355X box's host (EPYC 9575F), T = N = 1000 (profiles/r03_host_generator.txt): sequential stage
    // 1.9 ms (MT19937 blocks 0.70, uniforms 0.27, attempts 0.50, acceptance 0.25), transforms 1.9 ms: two workers keep up.
    int nt = threads > 0 ? threads : 3;
    if (nt > 16) nt = 16;
    if (total < 200000) nt = 1;
    const int W = nt - 1;                                       // transform workers
    // a slice = ~8k accepted pairs = 192 KB of attempts, whatever the row length (N = 10^6: a row is 61 slices; sliced by
    // rows it was one 12 MB slice per row in a 96 MB ring)
    constexpr size_t SLICE_PAIRS = 8192;
    const size_t cap = SLICE_PAIRS + 156 + 16;
    constexpr int RING = 8;
    // the ring lives as long as the calling thread (1.6 MB; allocating and zero-filling it per call was 40 us of a
    // 49 k-draw window's 150); the sequential stage owns it for the duration of the call, workers only read their slice
    static thread_local std::vector<double> ring_mem;
    if (ring_mem.size() < (size_t)RING * 3 * cap) {
        try { ring_mem.resize((size_t)RING * 3 * cap); } catch (...) { return PFG_ERR_NOMEM; }
    }
    // What the workers spin on lives on cache lines of its own: next to the main loop's locals (this stack frame) every
    // poll of a waiting worker would pull the line the sequential stage is writing to (measured on the GPU box's host,
    // two threads: the sequential stage 1.9 -> 5.2 ms with the flags on the stack).
    struct alignas(64) Slot { std::atomic<long> tag{0}; size_t q0 = 0, n = 0; };      // tag = slice + 1 while that slice waits in the buffer, 0 = free
    struct alignas(64) Shared { Slot slots[RING]; alignas(64) std::atomic<long> n_slices{-1}; char pad[64]; };
    std::unique_ptr<Shared> shared;
    try { shared.reset(new Shared); } catch (...) { return PFG_ERR_NOMEM; }
    Slot *const slots = shared->slots;
    std::atomic<long> &n_slices = shared->n_slices;             // set when the main loop is done
    double *const ring = ring_mem.data();
    auto seg = [ring, cap](long j, int which) -> double * { return ring + ((size_t)(j % RING) * 3 + which) * cap; };

    // (the workers' lambdas hold copies of the constants they need: nothing they read sits in this stack frame)
    auto transform_slice = [&spill, &spilled, slots, seg, off, total, N, z0, z, avx2](long j) {
        const Slot &sl = slots[j % RING];
        const double *a1 = seg(j, 0), *a2 = seg(j, 1), *a3 = seg(j, 2);
        double tmp[1024];
        size_t q = 0;
        while (q < sl.n) {
            const size_t m = sl.n - q < 512 ? sl.n - q : 512;
            for (size_t i = 0; i < m; ++i) transform(a1[q + i], a2[q + i], a3[q + i], tmp[2 * i], tmp[2 * i + 1]);
            // normals k0 .. k0 + 2m - 1 of the overall order: z0 first, then the rows of z
            size_t k0 = 2 * (sl.q0 + q) + off, cnt = 2 * m;
            const double *src = tmp;
            if (k0 + cnt > total) { spill = tmp[cnt - 1]; spilled = true; --cnt; }      // only the very last pair
            if (k0 < (size_t)N) {
                const size_t h = (size_t)N - k0 < cnt ? (size_t)N - k0 : cnt;
                std::memcpy(z0 + k0, src, h * sizeof(double));
                k0 += h; src += h; cnt -= h;
            }
            if (cnt) stream_copy(avx2, z + (k0 - N), src, cnt);
            q += m;
        }
    };
    std::vector<int> near;
    {
        const char *pin = std::getenv("PFGRAD_RNG_PIN");
        if (W > 0 && !(pin && pin[0] == '0')) near = cores_near(sched_getcpu());
    }
    auto worker = [slots, &n_slices, transform_slice, W, &near](int w) {
        if ((size_t)w < near.size()) {
            cpu_set_t cs;
            CPU_ZERO(&cs);
            CPU_SET(near[(size_t)w], &cs);
            pthread_setaffinity_np(pthread_self(), sizeof cs, &cs);        // best effort: a refusal leaves the thread where it is
        }
        for (long j = w;; j += W) {
            Slot &sl = slots[j % RING];
            unsigned spins = 0;
            while (sl.tag.load(std::memory_order_acquire) != j + 1) {      // another slice may still sit in this buffer
                const long ns = n_slices.load(std::memory_order_acquire);
                if (ns >= 0 && j >= ns) return;
                if (++spins < 64) _mm_pause(); else { std::this_thread::yield(); spins = 0; }
            }
            transform_slice(j);
            sl.tag.store(0, std::memory_order_release);
        }
    };
    std::vector<std::thread> pool;
    for (int w = 0; w < W; ++w) pool.emplace_back(worker, w);

    size_t npairs = 0;                       // accepted so far (all slices)
    long j = 0;                              // current slice
    size_t in_slice = 0;
    double *p1 = seg(0, 0), *p2 = seg(0, 1), *p3 = seg(0, 2);
    auto close_slice = [&]() {
        Slot &sl = slots[j % RING];
        sl.q0 = npairs - in_slice; sl.n = in_slice;
        if (W > 0) sl.tag.store(j + 1, std::memory_order_release);
        else transform_slice(j);
        ++j;
        in_slice = 0;
        if (W > 0) {                         // the next buffer of the ring must have been drained
            unsigned spins = 0;
            while (slots[j % RING].tag.load(std::memory_order_acquire) != 0) {
                if (++spins < 64) _mm_pause(); else { std::this_thread::yield(); spins = 0; }
            }
        }
        p1 = seg(j, 0); p2 = seg(j, 1); p3 = seg(j, 2);
    };
    double ubuf[312];
    for (int row = 0; row <= T; ++row) {
        const size_t row_end = (size_t)N * (row + 1);
        if (row > 0) {                                // uniforms of timestep row - 1 come BEFORE its normals
            double *ur = u + (size_t)(row - 1) * N;
            int i = 0;
            while (i < N) {
                const int fit = (624 - mt.pos) / 2;       // whole doubles left in the current block
                if (fit < 1) { ur[i++] = mt.next_double(); continue; }     // straddles a block boundary (or refill)
                const int n = fit < N - i ? fit : N - i;
                if (avx2) { doubles_avx2(mt.out + mt.pos, n, ubuf); stream_copy_avx2(ur + i, ubuf, (size_t)n); }
                else doubles_generic(mt.out + mt.pos, n, ur + i);
                mt.pos += 2 * n;
                i += n;
            }
        }
        // attempts until this row's normals are covered (a pair started in this row may spill one
        // variate into the next row: the cache)
        const size_t have = off + 2 * npairs;
        size_t need_pairs = row_end > have ? (row_end - have + 1) / 2 : 0;
        while (need_pairs > 0) {
            const int avail = (624 - mt.pos) / 4;         // whole attempts left in the current block
            if (avail < 1) {                              // an attempt that straddles the block boundary (or a refill)
                const double x1 = 2.0 * mt.next_double() - 1.0;
                const double x2 = 2.0 * mt.next_double() - 1.0;
                const double r2 = x1 * x1 + x2 * x2;
                if (!(r2 >= 1.0 || r2 == 0.0)) {
                    p1[in_slice] = x1; p2[in_slice] = x2; p3[in_slice] = r2;
                    ++in_slice; ++npairs; --need_pairs;
                    if (in_slice >= SLICE_PAIRS) close_slice();
                }
                continue;
            }
            double cx1[156], cx2[156], cr2[156];
            if (avx2) candidates_avx2(mt.out + mt.pos, avail, cx1, cx2, cr2);
            else candidates_generic(mt.out + mt.pos, avail, cx1, cx2, cr2);
            size_t got = 0;          // in order, stopping at the last attempt this row consumes
            const int used = avx2 ? accept_avx2(cx1, cx2, cr2, avail, need_pairs, p1 + in_slice, p2 + in_slice, p3 + in_slice, got)
                                  : accept_generic(cx1, cx2, cr2, avail, need_pairs, p1 + in_slice, p2 + in_slice, p3 + in_slice, got);
            in_slice += got;
            npairs += got;
            need_pairs -= got;
            mt.pos += 4 * used;
            if (in_slice >= SLICE_PAIRS) close_slice();
        }
        if (row == T) close_slice();
    }
    n_slices.store(j, std::memory_order_release);
    for (auto &th : pool) th.join();
    if (avx2) _mm_sfence();                                   // non-temporal stores visible before the buffers are handed on
    *pos = mt.pos;
    *has_gauss = spilled ? 1 : 0;
    *gauss = spilled ? spill : 0.0;
    return PFG_OK;
}
