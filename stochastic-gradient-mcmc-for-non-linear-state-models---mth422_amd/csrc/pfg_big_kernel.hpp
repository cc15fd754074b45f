// libpfgrad device code: pf_big_kernel, the device-generator fast path for large N.
#pragma once
#include "pfg_mem_kernel.hpp"

namespace pfg {

// ------------------------------------------------------------------------------------
// Large-N kernel, device-RNG fast path (N <= NP2, NP2 = 4096 | 16384): same phases and scratch layout as
// pf_mem_kernel, restructured around what the device generator allows:
//  * SORTED resampling uniforms.  The N uniforms of a timestep are drawn as the order statistics of N
//    i.i.d. uniforms -- exponential spacings e_r = -log(u_r), U_(r) = (e_1 + .. + e_r) / (e_1 + .. + e_{N+1}) --
//    by a prefix scan that rides on the weight scan's barriers, and child r takes U_(r).  Multinomial
//    resampling does not care which child gets which of its N uniforms; handed out in index order the
//    children of neighbouring lanes have neighbouring (often equal) ancestors, so the parent gather -- 32-byte
//    records out of a per-window scratch far larger than L2 -- reads runs of consecutive records instead of
//    one random 32-byte sector per child (profiles/hbm_traffic.json: 17.0 GB of fabric traffic per launch
//    against 9.8 GB of algorithmic bytes with independent uniforms).  CDF, ranks and storage all run in
//    particle order; the spacings are not stored (LDS is full with the CDF) but recomputed in the sweep
//    from a snapshot of the lane generator, their wave scan included;
//  * the binary search is unrolled for the compile-time NP2 (probe offsets fold into the ds_read
//    immediates, slots past N hold a sentinel) and two chunks are in flight per iteration (two independent
//    search / gather chains per lane, and both normals of a Box-Muller pair are used).
// REPLAY (the reference's own uniforms), PaRIS and the predictive statistic stay on pf_mem_kernel.
// ------------------------------------------------------------------------------------
template <typename REAL>
__host__ __device__ inline size_t big_kernel_lds_bytes(int NP2) {
    // CDF (padded) | 4 x [chunks * waves] scan totals / offsets (weights, spacings) | wave maxima | S partials | tables
    return ((size_t)NP2 + NP2 / 32) * 8 + (size_t)(4 * (NP2 / MEM_NT) * MEM_NW + MEM_NW + PFG_MAX_STAT * MEM_NW + 8) * 8 +
           tab_bytes<REAL, PFG_RNG_DEVICE, true>();
}

template <int MODEL, int KERNEL, typename REAL, int NP2>
__global__ __launch_bounds__(MEM_NT) void pf_big_kernel(const pfg_dev_problem *__restrict__ probs) {
    constexpr int RNG = PFG_RNG_DEVICE;
    constexpr int NS = ModelDims<MODEL>::NS;
    constexpr int H = ModelDims<MODEL>::H;
    constexpr int NT = MEM_NT, NW = MEM_NW;
    constexpr int CH2 = NP2 / NT;                                   // CDF positions per thread (4 | 16)
    static_assert(CH2 == 4 || CH2 == 16, "NP2 must be 4096 or 16384");
    static_assert(CH2 * NW <= 4 * WAVE, "the (chunk, wave) totals are prefix-summed 4 per lane by one wave");
#ifndef PFG_BIG_G
#define PFG_BIG_G 2
#endif
    constexpr int G = PFG_BIG_G;                                    // chunks in flight
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
    const gptr<const double> yv = global_ptr(P.y);
    const gptr<const double> wv = global_ptr(P.weights);

    double *cdf = reinterpret_cast<double *>(smem);                 // [NP2 + NP2/32] physical
    double *red_scan = cdf + (NP2 + NP2 / 32);                      // [CH2*NW] (chunk, wave) totals of the weights
    double *red_scanE = red_scan + CH2 * NW;                        // [CH2*NW] ... of the exponential spacings
    double *red_off = red_scanE + CH2 * NW;                         // [CH2*NW] exclusive offsets of red_scan
    double *red_offE = red_off + CH2 * NW;                          // [CH2*NW] ... of red_scanE
    double *red_max = red_offE + CH2 * NW;                          // [NW]
    float *red_maxf = reinterpret_cast<float *>(red_max);
    double *red_S = red_max + NW;                                   // [H*NW]
    double *red_W = red_S + PFG_MAX_STAT * NW;                      // [8]: W, total of the spacings
    double *tabmem = red_W + 8;

    constexpr int REC = mem_rec_len<MODEL, REAL>();
    gptr<REAL> lwg = global_ptr(reinterpret_cast<REAL *>(P.scratch));    // [N]
    gptr<REAL> cur = (gptr<REAL>)(((uintptr_t)(lwg + N) + 15) & ~(uintptr_t)15);   // [N][REC]
    gptr<REAL> nxt = cur + (size_t)REC * N;

    Math<REAL, true> mth;
    mth.t.e2 = tabmem;
    mth.t.lg = reinterpret_cast<const double2 *>(tabmem + TAB_E2);
    mth.t.sc = reinterpret_cast<const double2 *>(tabmem + TAB_E2 + 2 * TAB_LG);
    tab_fill(tabmem, true, tid, NT);
    for (int i = N + tid; i < NP2; i += NT) cdf[cdf_phys(i)] = 2.0;      // slots past N: never <= u (set once)

    const Consts<REAL> c = make_consts<MODEL, REAL>(P.theta);
    if (P.stamps && tid == 0) { P.stamps[0] = __builtin_amdgcn_s_memtime(); P.stamps[1] = __builtin_amdgcn_s_memrealtime(); }
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
                if (P.trace_x && P.rec_z0) P.rec_z0[i] = (double)a;
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
        // ---- (B,C) weights and exponential spacings: one wave scan per 1024-particle chunk each ----
        const bool needS = needS_every || (t == T);
        const LaneRng rngE = rng;                 // the sweep recomputes this step's spacings from here
        float e_extra = 0.0f;
        {
            double part[H];
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
                    const double inc = wave_incl_scan(p);
                    if (v) cdf[cdf_phys(i)] = inc;                  // wave-local; globalised in (D)
                    if (lane == WAVE - 1) red_scan[j * NW + wave] = inc;
                    if (t < T) {
                        // spacing of child i's sorted uniform (valid children only: N uniforms, N + 1 spacings)
                        const float ef = spacing_f32(rng.next());
                        const double incE = wave_incl_scan(v ? (double)ef : 0.0);
                        if (lane == WAVE - 1) red_scanE[j * NW + wave] = incE;
                    }
                }
            }
            if (t < T) e_extra = spacing_f32(rng.next());           // spacing N + 1: part of the total only
            if (needS) {
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    const double tot = wave_sum(part[h]);
                    if (lane == 0) red_S[h * NW + wave] = tot;
                }
            }
        }
        __syncthreads();                                                        // barrier 2
        if (wave < 2) {
            // exclusive offsets of the nchunk*NW (chunk, wave) totals (<= 256): 4 per lane + one wave scan.
            // Wave 0: weights; wave 1: spacings.
            const double *src = wave == 0 ? red_scan : red_scanE;
            double *dst = wave == 0 ? red_off : red_offE;
            const int ntot = nchunk * NW;
            double v4[4], loc = 0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = lane * 4 + q;
                v4[q] = (idx < ntot && (wave == 0 || t < T)) ? src[idx] : 0.0;
                loc += v4[q];
            }
            const double inc = wave_incl_scan(loc);
            double run = inc - loc;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = lane * 4 + q;
                if (idx < ntot) dst[idx] = run;
                run += v4[q];
            }
            if (lane == WAVE - 1) red_W[wave] = inc + (wave == 1 ? (double)e_extra : 0.0);
        }
        __syncthreads();                                                        // barrier 2b
        W = uniform_f64(red_W[0]);
        const double invW = uniform_f64(1.0 / W);
        const double invEtot = (t < T) ? uniform_f64(1.0 / red_W[1]) : 0.0;
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
#pragma unroll 2
        for (int j = 0; j < CH2; ++j) {
            const int i = j * NT + tid;
            if (j < nchunk && i < N) {
                const int pi = cdf_phys(i);
                cdf[pi] = (cdf[pi] + red_off[j * NW + wave]) * invW;
            }
        }
        __syncthreads();                                                        // barrier 3

        const double y_t = yv[t];
        const bool inside = (t >= t1) && (t < tL);
        const double wt = (inside && wv) ? wv[t - t1] : 1.0;
        const bool use_stat = inside && (stat != PFG_STAT_NONE);
        // ---- (E..H) two chunks per iteration: search, gather parent (L2), propose, publish ----
        LaneRng rngS = rngE;                      // regenerates the words the scan above turned into spacings
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
                    a[g] = 0;
                    u[g] = 2.0;
                    if (j0 + g < nchunk) {
                        // child i's uniform = the sorted uniform of rank i: the same word, spacing and wave scan
                        // as in (B,C), plus this (chunk, wave)'s offset
                        const float ef = spacing_f32(rngS.next());
                        const double incE = wave_incl_scan(v[g] ? (double)ef : 0.0);
                        u[g] = (incE + red_offE[(j0 + g) * NW + wave]) * invEtot;
                        if (P.trace_x && P.rec_ud && v[g]) P.rec_ud[(size_t)t * N + i[g]] = u[g];
                    }
                }
#pragma unroll
                for (int g = 0; g < G; g += 2) mth.normal_pair(rng.next(), rng.next(), z[g], z[g + 1]);
                if (P.trace_x && P.rec_z) {           // test instrumentation (see pfg_result.rec_z)
#pragma unroll
                    for (int g = 0; g < G; ++g)
                        if (v[g]) P.rec_z[(size_t)t * N + i[g]] = (double)z[g];
                }
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
                    a[g] -= (a[g] * 993) >> 15;                    // physical -> CDF position = particle index (exact < 32768)
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
        { gptr<REAL> tmp = cur; cur = nxt; nxt = tmp; }
        wt_prev = wt;
        // children (global stores) become visible to the next step's gathers at its barriers
    }

    // ---- outputs --------------------------------------------------------------------------
    if (P.stamps && tid == 0) { P.stamps[2] = __builtin_amdgcn_s_memtime(); P.stamps[3] = __builtin_amdgcn_s_memrealtime(); }
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
