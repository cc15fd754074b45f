// libpfgrad device code: pf_mem_kernel, the general large-N particle filter (state in an HBM scratch;
// REPLAY parity path, predictive statistic, PaRIS).
#pragma once
#include "pfg_reg_kernel.hpp"

namespace pfg {

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
// the same for records in device memory addressed as such (global_load / global_store instead of flat accesses: see
// global_ptr in pfg_math.hpp)
template <int REC, typename REAL>
__device__ __forceinline__ void rec_load(REAL *dst, gptr<const REAL> src) {
    typedef float V __attribute__((ext_vector_type(4)));      // (a builtin vector: float4's operators want a generic `this`)
#pragma unroll
    for (int v = 0; v < REC * (int)sizeof(REAL) / 16; ++v)
        reinterpret_cast<V *>(dst)[v] = ((gptr<const V>)src)[v];
}
template <int REC, typename REAL>
__device__ __forceinline__ void rec_store(gptr<REAL> dst, const REAL *src) {
    typedef float V __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int v = 0; v < REC * (int)sizeof(REAL) / 16; ++v)
        ((gptr<V>)dst)[v] = reinterpret_cast<const V *>(src)[v];
}
template <typename REAL, int RNG>
__host__ __device__ inline size_t mem_kernel_lds_bytes(int N) {
    const size_t np2 = (size_t)mem_np2(N);
    return (np2 + np2 / 32) * 8 + (size_t)(2 * MEM_MAX_CHUNKS * MEM_NW + MEM_NW + PFG_MAX_STAT * MEM_NW + 8) * 8 +
           (size_t)(PFG_MAX_PRED * MEM_NW + MEM_NW + 2 * PFG_MAX_PRED) * 8 + tab_bytes<REAL, RNG, true>();
}

#ifndef PFG_MEM_PREFETCH
#define PFG_MEM_PREFETCH 0      /* measured (round 4, c4 REPLAY, ms per 64 / 256 chains): 22.6 / 29.1 with, 14.1 / 20.0 without: 18 more spilled VGPRs */
#endif
// LW4 (round 4; N <= 4096, plain statistics, not PaRIS): a thread's (<= 4) log-weights never leave its registers -- it is
// their only reader and writer --, which takes the global-memory round trip out of the maximum and the weight phases
// of every timestep (BASELINE config 4 through the seed-compatible path).
// SCORE1 (PFG_SMOOTHER_POYIADJIS_N launches of the N <= 4096 variant): the Poyiadjis O(N) score only -- the filter, the
// lambda != 1 shrinkage and the other statistics compiled out.  BASELINE config 4 through the seed-compatible path:
// 37.9 -> 32.1 ms per 402 windows (-15 %; the N = 10^4 kernel gains nothing: profiles/r04_ab_score1_twin.txt).  REPLAY
// units are built without contraction: bitwise the general kernel's numbers.  A window that is not that estimator gets NaNs.
template <int MODEL, int KERNEL, typename REAL, int RNG, bool PARIS = false, bool LW4 = false, bool SCORE1 = false>
__global__ __launch_bounds__(MEM_NT) void pf_mem_kernel(const pfg_dev_problem *__restrict__ probs) {
    static_assert(!(PARIS && LW4), "LW4 is a variant of the plain kernel");
    static_assert(!SCORE1 || LW4, "the score-only twin exists for the N <= 4096 variant");
    constexpr int NS = ModelDims<MODEL>::NS;
    constexpr int H = ModelDims<MODEL>::H;
    constexpr int NT = MEM_NT, NW = MEM_NW;
    extern __shared__ __align__(16) unsigned char smem[];

    const pfg_dev_problem &P = probs[blockIdx.x];
    const int N = P.N, T = P.T, t1 = P.t1, tL = P.tL;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    // PFG_FLAG_PARIS_RAW_STREAM (the window's whole np.random stream, see pfgrad.h): the fp64 REPLAY instantiation takes it
    // (round 4: one launch per window for 1024 < N <= 16384 too); any other instantiation has no u / z to fall back on and
    // reports "stream too short" instead of reading NULL
    constexpr bool RAWCAP = PARIS && RNG == PFG_RNG_REPLAY && sizeof(REAL) == 8;
    if (PARIS && !RAWCAP && (P.flags & PFG_FLAG_PARIS_RAW_STREAM)) {
        if (tid == 0 && P.paris_consumed) P.paris_consumed[0] = -1ll;
        return;
    }
    const int wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
    const int nchunk = (N + NT - 1) / NT;
    const int np2 = mem_np2(N);
    if constexpr (SCORE1) {
        if (P.smoother != PFG_SMOOTHER_NEMETH || P.lambduh != 1.0 || P.stat != PFG_STAT_SCORE) {
            if (threadIdx.x < PFG_OUT_DOUBLES && P.out) P.out[threadIdx.x] = __builtin_nan("");
            return;
        }
    }
    const bool is_filter = !SCORE1 && (P.smoother == PFG_SMOOTHER_FILTER);
    const int stat = SCORE1 ? (int)PFG_STAT_SCORE : P.stat;
    const double lam_d = SCORE1 ? 1.0 : is_filter ? 0.0 : (PARIS ? 1.0 : P.lambduh);
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

    // ---- PaRIS, the reference's whole np.random stream in ONE launch (as pf_reg_kernel does for N <= 1024) ---------------
    // The N normals of a call are NumPy's legacy Gaussians -- Marsaglia's polar method on pairs of doubles, the second
    // variate of a pair cached for the next draw --, generated here from the raw doubles: attempts are evaluated NT * 4 at a
    // time, the accepted ones ranked in stream order by a workgroup-wide count, the q-th accepted pair fills normals 2q and
    // 2q + 1 of the call, the attempt that completes the call moves the cursor.  Accept / reject is exact fp64 arithmetic
    // (no contraction in the REPLAY units): the CONSUMPTION is the reference's to the double; the values go through the
    // device's log (<= 1 ulp from the host libm's: REPLAY tolerance).
    long long paris_cursor = 0;            // doubles of P.paris_stream consumed so far
    bool paris_overflow = false;
    const bool raw = RAWCAP && (P.flags & PFG_FLAG_PARIS_RAW_STREAM) != 0 && P.paris_stream != nullptr;
    bool carry_has = false;                // a cached second variate is pending (workgroup-uniform); its value sits in red_W[2]
    double *const zbuf = reinterpret_cast<double *>(qum);                   // [N] normals of the current call (the fallback queue's uniforms are not in use then)
    long long *const raw_slots = reinterpret_cast<long long *>(red_W + 3);  // [0] cut-off attempt of a call, [1] stream position of the cached pair
    [[maybe_unused]] auto legacy_normals = [&]() __attribute__((noinline)) {
        constexpr int RPT = 4;
        const gptr<const double> strm = global_ptr(P.paris_stream);
        const long long cap = P.paris_stream_len;
        const unsigned long long ltm = (1ull << lane) - 1ull;
        int *const wc = reinterpret_cast<int *>(red_scan);                 // [RPT][NW] accepted attempts per (slot, wave)
        const int produced0 = carry_has ? 1 : 0;
        __syncthreads();
        if (carry_has && tid == 0) zbuf[0] = red_W[2];
        const int need = (N - produced0 + 1) >> 1;                         // pairs to accept
        const long long base = paris_cursor;
        int acc = 0;
        for (int round = 0; acc < need && !paris_overflow; ++round) {
            double x1[RPT], x2[RPT], r2[RPT];
            bool fl[RPT];
            int rank[RPT];
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const long long p = base + 2 * ((long long)round * (NT * RPT) + k * NT + tid);
                const bool in = p + 1 < cap;
                const double d0 = in ? strm[p] : 0.5, d1 = in ? strm[p + 1] : 0.5;
                x1[k] = 2.0 * d0 - 1.0;
                x2[k] = 2.0 * d1 - 1.0;
                r2[k] = x1[k] * x1[k] + x2[k] * x2[k];
                fl[k] = in && !(r2[k] >= 1.0 || r2[k] == 0.0);
                const unsigned long long mk = __ballot(fl[k]);
                rank[k] = __popcll(mk & ltm);
                if (lane == 0) wc[k * NW + wave] = __popcll(mk);
            }
            __syncthreads();
            int Sacc = 0;
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                for (int w = 0; w < NW; ++w) {
                    if (w == wave) rank[k] += Sacc;
                    Sacc += wc[k * NW + w];
                }
            }
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const int q = acc + rank[k];
                if (fl[k] && q < need) {
                    const double f = ::sqrt(-2.0 * ::log(r2[k]) / r2[k]);
                    const int i0 = produced0 + 2 * q;
                    zbuf[i0] = f * x2[k];                                  // returned by this call of legacy_gauss
                    const long long j = (long long)round * (NT * RPT) + k * NT + tid;
                    if (i0 + 1 < N) zbuf[i0 + 1] = f * x1[k];              // the cached one, returned by the next call
                    else { red_W[2] = f * x1[k]; raw_slots[1] = base + 2 * j; }
                    if (q == need - 1) raw_slots[0] = j;
                }
            }
            acc += Sacc;
            if (acc < need && base + 2 * ((long long)(round + 1) * (NT * RPT)) + 1 >= cap) paris_overflow = true;
            __syncthreads();
        }
        if (need > 0 && !paris_overflow) paris_cursor = base + 2 * (raw_slots[0] + 1);
        carry_has = ((N - produced0) & 1) != 0;
    };
    if (RAWCAP && raw && (P.flags & PFG_FLAG_PARIS_RAW_CARRY)) {          // the generator came with a cached Gaussian: stream[0]
        carry_has = true;
        if (tid == 0) red_W[2] = P.paris_stream[0];
        paris_cursor = 1;
    }
    if constexpr (RAWCAP) { if (raw && !P.init_x) legacy_normals(); }
    REAL lwr[4];                                                    // LW4: this thread's log-weights (slots past N: -inf)
#pragma unroll
    for (int q = 0; q < 4; ++q) lwr[q] = (REAL)(-INFINITY);
    // ---- x0 or warm start ---------------------------------------------------------------
    {
        double pv = P.prior_var;
        if (MODEL == PFG_MODEL_GARCH && (P.flags & PFG_FLAG_GARCH_STATIONARY_PRIOR))
            pv = (double)c.alpha / (1.0 - (double)c.beta - (double)c.gamma);
        const double sd = sqrt(pv);
        for (int jj = 0; jj < MEM_MAX_CHUNKS; ++jj) {
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
                double z;
                if (RNG == PFG_RNG_REPLAY) z = (RAWCAP && raw) ? zbuf[i] : P.z0[i];
                else { REAL a, b; mth.normal_pair(rng.next(), rng.next(), a, b); z = (double)a; }
                x[0] = (REAL)(P.prior_mean + sd * z);
            }
            if (LW4) {
#pragma unroll
                for (int q = 0; q < 4; ++q) lwr[q] = q == jj ? l0 : lwr[q];
            } else {
                lwg[i] = l0;
            }
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
        // LW4 + REPLAY: this timestep's draws are asked for before anything else (they depend on nothing): their latency
        // runs under the maximum / weight / CDF phases instead of in front of the ancestor search
        double upre[4];
        REAL zpre[4];
        if (LW4 && RNG == PFG_RNG_REPLAY && PFG_MEM_PREFETCH && t < T) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int i = g * NT + tid;
                const int ii = i < N ? i : N - 1;
                upre[g] = uv[(size_t)t * N + ii];
                zpre[g] = (REAL)zv[(size_t)t * N + ii];
            }
        }
        // ---- (A) max of the log weights (f32-rounded shift, see wave_max) -------------------
        float ml = -INFINITY;
        if (LW4) {
#pragma unroll
            for (int q = 0; q < 4; ++q) ml = fmaxf(ml, (float)lwr[q]);
        } else {
            for (int i = tid; i < N; i += NT) ml = fmaxf(ml, (float)lwg[i]);
        }
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
#pragma unroll (LW4 ? 4 : 1)
            for (int j = 0; j < (LW4 ? 4 : nchunk); ++j) {
                if (LW4 && j >= nchunk) break;
                const int i = j * NT + tid;
                const bool v = i < N;
                const int ii = v ? i : N - 1;
                const REAL lwv = LW4 ? lwr[LW4 ? j : 0] : lwg[ii];
                double p = (double)mth.exp((REAL)(lwv - (REAL)m));
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
        // The same sweep for the plain statistics, four chunks at a time (round 4): the four children's draws are loaded
        // together, their ancestor searches run as four independent probe chains, the four parent records are in flight
        // together -- one exposed memory latency per group of four instead of two per child (config 4 through the
        // seed-compatible path: 16.8 -> see DESIGN 4.2 us per timestep).  Arithmetic and its order are those of `sweep`.
        auto sweep4 = [&](auto stat_tag) {
            constexpr int STAT = decltype(stat_tag)::value;
            constexpr int GQ = 4;
            for (int j0 = 0; j0 < nchunk; j0 += GQ) {
                int i[GQ], a[GQ], pos[GQ];
                bool v[GQ];
                double u[GQ];
                REAL z[GQ];
#pragma unroll
                for (int g = 0; g < GQ; ++g) {
                    i[g] = (j0 + g) * NT + tid;
                    v[g] = i[g] < N;
                    const int ii = v[g] ? i[g] : N - 1;
                    if (LW4 && RNG == PFG_RNG_REPLAY && PFG_MEM_PREFETCH) { u[g] = upre[g]; z[g] = zpre[g]; }
                    else if (RNG == PFG_RNG_REPLAY) { u[g] = uv[(size_t)t * N + ii]; z[g] = (REAL)zv[(size_t)t * N + ii]; }
                    else { REAL zb; u[g] = u01_32(rng.next()); mth.normal_pair(rng.next(), rng.next(), z[g], zb); }
                    pos[g] = 0;
                }
                for (int step = np2 >> 1; step >= 1; step >>= 1) {
                    const int probe = step - 1 + (step >= 32 ? (step >> 5) - 1 : 0);
                    const int adv = step + (step >> 5);
#pragma unroll
                    for (int g = 0; g < GQ; ++g) pos[g] += (cdf[pos[g] + probe] <= u[g]) ? adv : 0;
                }
                alignas(16) REAL rec[GQ][REC];
#pragma unroll
                for (int g = 0; g < GQ; ++g) {
                    a[g] = np2 > 8192 ? pos[g] - pos[g] / 33 : pos[g] - ((pos[g] * 993) >> 15);
                    a[g] = a[g] < N - 1 ? a[g] : N - 1;
                    rec_load<REC, REAL>(rec[g], cur + (size_t)a[g] * REC);
                    if (RNG == PFG_RNG_REPLAY && v[g]) {
                        const double hi = cdf[cdf_phys(a[g])] - u[g];
                        const double lo = a[g] > 0 ? u[g] - cdf[cdf_phys(a[g] - 1)] : 1.0;
                        const double mg = hi < lo ? hi : lo;
                        tie = mg < tie ? mg : tie;
                    }
                }
#pragma unroll
                for (int g = 0; g < GQ; ++g) {
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
                    if (LW4) lwr[g] = v[g] ? lwn : (REAL)(-INFINITY);       // (LW4: nchunk <= 4, one group: chunk index = g)
                    if (v[g]) {
                        if (!LW4) lwg[i[g]] = lwn;
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
            // whole-window stream: this timestep's N resampling uniforms sit at the cursor, the N normals of Kernel.rv behind
            // them (legacy Gaussians from the raw doubles), then the backward sampling's run
            long long cursor_u = 0;
            if constexpr (RAWCAP) {
                if (raw) {
                    cursor_u = paris_cursor;
                    if (paris_cursor + N > P.paris_stream_len) paris_overflow = true;
                    paris_cursor += N;
                    if (!paris_overflow) legacy_normals();
                    __syncthreads();
                }
            }
            // ---- 1. propose every child from its filter ancestor, publish x' and log-weight ----
            for (int j = 0; j < nchunk; ++j) {
                const int i = j * NT + tid;
                const bool v = i < N;
                const int ii = v ? i : N - 1;
                double u;
                REAL z;
                if (RAWCAP && raw) { u = paris_overflow ? 0.5 : P.paris_stream[cursor_u + ii]; z = (REAL)zbuf[ii]; }
                else if (RNG == PFG_RNG_REPLAY) { u = uv[(size_t)t * N + ii]; z = (REAL)zv[(size_t)t * N + ii]; }
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
            int jt_cur = 0;
            auto contribute = [&](int ci, int J) {
                if (P.trace_x && P.trace_paris_J) P.trace_paris_J[((size_t)t * Nt + jt_cur) * N + ci] = J;
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
                jt_cur = jt;
                if (tid == 0) *qcount = 0;
                __syncthreads();
                // ---- 2. accept-reject against the filter weights, up to R rounds per child ------
                // as pf_reg_kernel's paris_slots: pending children compacted in wave-local queues
                // (here in the scratch), one candidate per child and pass while more than half a
                // wave is pending, K = 2^k consecutive rounds per child and pass below that
                int *qa = wq0 + wave * (nchunk * WAVE), *qb = wq1 + wave * (nchunk * WAVE);
                int cnt = 0;
                const bool ordered = RNG == PFG_RNG_REPLAY && P.paris_stream != nullptr;
                if (ordered) {
                    // the REFERENCE's consumption order (pf.py:260-341), as in pf_reg_kernel's paris_slots: per round the
                    // k-th pending child IN INDEX ORDER takes the k-th double of the round's two blocks; the rank = a
                    // workgroup-wide exclusive count of the pending flags (chunk-major = particle-index order)
                    const double *__restrict__ const strm = P.paris_stream;
                    const long long cap = P.paris_stream_len;
                    const bool noar = (P.flags & PFG_FLAG_PARIS_NO_ACCEPT_REJECT) != 0;
                    const int mthr = P.paris_manual_threshold;
                    int *const wcnt = reinterpret_cast<int *>(red_scan);     // [nchunk][NW] (free outside phases B-D)
                    int *const woff = reinterpret_cast<int *>(red_off);      // [nchunk][NW] exclusive offsets, [256] = total
                    uint32_t pend = 0;
                    for (int j = 0; j < nchunk; ++j) pend |= (j * NT + tid < N) ? (1u << j) : 0u;
                    int Stot = 0;
                    for (int round = 0;; ++round) {
                        for (int j = 0; j < nchunk; ++j) {
                            const unsigned long long mk = __ballot((pend >> j) & 1u);
                            if (lane == 0) wcnt[j * NW + wave] = __popcll(mk);
                        }
                        __syncthreads();
                        if (wave == 0) {
                            const int ntot = nchunk * NW;
                            double v4[4], loc = 0.0;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int idx = lane * 4 + q;
                                v4[q] = idx < ntot ? (double)wcnt[idx] : 0.0;
                                loc += v4[q];
                            }
                            const double inc = wave_incl_scan(loc);
                            double run = inc - loc;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int idx = lane * 4 + q;
                                if (idx < ntot) woff[idx] = (int)run;
                                run += v4[q];
                            }
                            if (lane == WAVE - 1) woff[MEM_MAX_CHUNKS * NW] = (int)inc;
                        }
                        __syncthreads();
                        Stot = woff[MEM_MAX_CHUNKS * NW];
                        if (Stot == 0 || noar || Stot <= mthr || round >= R || paris_overflow) break;
                        if (paris_cursor + 2ll * Stot > cap) { paris_overflow = true; break; }
                        for (int j = 0; j < nchunk; ++j) {
                            const bool pj = (pend >> j) & 1u;
                            const unsigned long long mk = __ballot(pj);
                            if (!pj) continue;
                            const int i = j * NT + tid;
                            const int rank = woff[j * NW + wave] + __popcll(mk & ltmask);
                            const double u1 = strm[paris_cursor + rank], u2 = strm[paris_cursor + Stot + rank];
                            const int I = search(u1);
                            REAL xI[NS], xc[NS];
#pragma unroll
                            for (int d = 0; d < NS; ++d) { xI[d] = cur[(size_t)I * REC + d]; xc[d] = nxt[(size_t)i * REC + d]; }
                            const double thr = (double)mth.exp(backward_log_ratio<MODEL, REAL>(c, mth, xI, xc));
                            if (u2 <= thr) { contribute(i, I); pend &= ~(1u << j); }
                        }
                        paris_cursor += 2ll * Stot;
                        __syncthreads();                           // woff / wcnt are rewritten by the next round
                    }
                    if (Stot > 0 && !noar && paris_cursor + Stot > cap) paris_overflow = true;
                    // accept_reject = False: child i's draw j reads double i Ntilde + j of the timestep's N Ntilde -- behind the cursor
                    // in a whole-window stream, at (t N + i) Ntilde + j in a stream of backward draws only
                    const long long noar_base = (RAWCAP && raw) ? paris_cursor : (long long)t * N * Nt;
                    if (Stot > 0 && noar && noar_base + (long long)N * Nt > cap) paris_overflow = true;
                    if (!paris_overflow && Stot > 0) {
                        for (int j = 0; j < nchunk; ++j) {
                            const bool pj = (pend >> j) & 1u;
                            const unsigned long long mk = __ballot(pj);
                            if (!pj) continue;
                            const int i = j * NT + tid;
                            const int rank = woff[j * NW + wave] + __popcll(mk & ltmask);
                            qchild[rank] = i;
                            qum[rank] = (REAL)(noar ? strm[noar_base + (long long)i * Nt + jt] : strm[paris_cursor + rank]);
                        }
                        if (tid == 0) *qcount = Stot;
                        if (!noar) paris_cursor += Stot;
                    }
                    __syncthreads();
                }
                if (!ordered) {
                for (int j = 0; j < nchunk; ++j) {
                    const int i = j * NT + tid;
                    const bool v = i < N;
                    const unsigned long long mk = __ballot(v);
                    if (v) qa[cnt + __popcll(mk & ltmask)] = i;
                    cnt += __popcll(mk);
                }
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
            if (RAWCAP && raw && (P.flags & PFG_FLAG_PARIS_NO_ACCEPT_REJECT) && !paris_overflow)
                paris_cursor += (long long)N * Nt;            // accept_reject = False in a whole-window stream: the timestep's N Ntilde draws
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
        // Poyiadjis O(N^2) for 1024 < N <= 4096 (pf.py:84-136), in the PARIS instantiation (it has the second
        // log-weight array, so the parents stay intact while the children are built): every child is proposed from
        // its filter ancestor, then averages  stats_q + w_t h(x_q, child)  over ALL parents q with the backward
        // weights  log_normalize(logw_q + log q(child | x_q)) -- two passes over the parents (exact maximum, then
        // exp-sums), the parent records read with wave-uniform addresses from the L2-resident scratch.  One child
        // chunk at a time: the arithmetic order is pf_reg_kernel's n2_slots.
        auto n2_sweep = [&](auto stat_tag) {
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
                int a = pos - pos / 33;
                a = a < N - 1 ? a : N - 1;
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
                const REAL aux = (MODEL == PFG_MODEL_SVM) ? mth.exp(-xn[0]) : (REAL)0;
                REAL mx = (REAL)(-INFINITY);
#pragma unroll 2
                for (int q = 0; q < N; ++q) {
                    alignas(16) REAL rq[REC];
                    rec_load<REC, REAL>(rq, cur + (size_t)q * REC);
                    const REAL vq = lwg[q] + backward_log_ratio<MODEL, REAL>(c, mth, rq, xn);
                    mx = vq > mx ? vq : mx;
                }
                REAL den = (REAL)0, num[H];
#pragma unroll
                for (int h = 0; h < H; ++h) num[h] = (REAL)0;
#pragma unroll 2
                for (int q = 0; q < N; ++q) {
                    alignas(16) REAL rq[REC];
                    rec_load<REC, REAL>(rq, cur + (size_t)q * REC);
                    const REAL e = mth.exp((lwg[q] + backward_log_ratio<MODEL, REAL>(c, mth, rq, xn)) - mx);
                    REAL aj[H];
                    additive_stat<MODEL, STAT, REAL>(c, rq, xn, (REAL)y_t, aux, aj);
                    den += e;
#pragma unroll
                    for (int h = 0; h < H; ++h) {
                        const REAL av = use_stat ? aj[h] * (REAL)wt : (REAL)0;
                        num[h] += e * (rq[NS + h] + av);
                    }
                }
                if (v) {
                    lwn_g[i] = lwn;
#pragma unroll
                    for (int q = 0; q < REC; ++q) rec[q] = (REAL)0;
#pragma unroll
                    for (int d = 0; d < NS; ++d) rec[d] = xn[d];
#pragma unroll
                    for (int h = 0; h < H; ++h) rec[NS + h] = num[h] / den;
                    rec_store<REC, REAL>(nxt + (size_t)i * REC, rec);
                    if (P.trace_x) {
                        const size_t row = (size_t)(t + 1) * N + i;
                        if (P.trace_anc) P.trace_anc[(size_t)t * N + i] = a;
#pragma unroll
                        for (int d = 0; d < NS; ++d) P.trace_x[row * NS + d] = (double)xn[d];
                        P.trace_logw[row] = (double)lwn;
                        if (P.trace_stats) {
#pragma unroll
                            for (int h = 0; h < H; ++h) P.trace_stats[row * H + h] = (double)rec[NS + h];
                        }
                    }
                }
            }
            { REAL *tmp = lwg; lwg = lwn_g; lwn_g = tmp; }
        };
        if constexpr (PARIS) {
            if (P.smoother == PFG_SMOOTHER_POYIADJIS_N2) {
                if (stat == PFG_STAT_SCORE) n2_sweep(std::integral_constant<int, PFG_STAT_SCORE>{});
                else n2_sweep(std::integral_constant<int, PFG_STAT_SUFF>{});
            } else if (stat == PFG_STAT_SCORE) paris_sweep(std::integral_constant<int, PFG_STAT_SCORE>{});
            else paris_sweep(std::integral_constant<int, PFG_STAT_SUFF>{});
        } else {
            // measured (round 4, profiles/r04_ab_mem_kernel.txt): the grouped sweep pays for N <= 4096 (one group); at N = 10000
            // (ten chunks: 4 + 4 + 2) it costs 3-8 %, so the general kernel keeps the one-chunk sweep
            if (!LW4) {
                if (stat == PFG_STAT_SCORE) sweep(std::integral_constant<int, PFG_STAT_SCORE>{});
                else sweep(std::integral_constant<int, PFG_STAT_SUFF>{});
            } else if (stat == PFG_STAT_SCORE) sweep4(std::integral_constant<int, PFG_STAT_SCORE>{});
            else sweep4(std::integral_constant<int, PFG_STAT_SUFF>{});
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
    if (PARIS && tid == 0 && P.paris_consumed) {
        if ((P.flags & PFG_FLAG_PARIS_NO_ACCEPT_REJECT) && !(RAWCAP && raw)) paris_cursor = (long long)N * P.Ntilde * T;
        P.paris_consumed[0] = paris_overflow ? -1ll : paris_cursor;
        P.paris_consumed[1] = (RAWCAP && raw && carry_has && !paris_overflow) ? paris_cursor - raw_slots[1] : 0ll;
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
            if (P.final_logw) P.final_logw[i] = (double)(LW4 ? lwr[(i / NT) & 3] : lwg[i]);
            if (P.final_stats && !is_filter) {
#pragma unroll
                for (int h = 0; h < H; ++h) P.final_stats[(size_t)i * H + h] = (double)cur[(size_t)i * REC + NS + h];
            }
        }
    }
}

}  // namespace pfg
