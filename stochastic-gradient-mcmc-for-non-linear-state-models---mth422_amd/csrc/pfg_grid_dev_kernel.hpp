// libpfgrad device code: the timestep of the whole-GPU window with the DEVICE generator -- the throughput path.
// (Overview of the whole-GPU window: pfg_grid_kernel.hpp.)
//
// One launch = one timestep of every window of the batch; one workgroup = one tile of TILE = NT * PPT children.  A
// workgroup's timestep is a chain of dependent steps -- reduce the tile partials, place its sorted uniforms among the
// tiles' cumulative weights, per pair of parent tiles: load the tiles' scans, search them, gather the parents, propose,
// weight, write, then the partials of its children -- every link a memory or LDS round trip that nothing inside the
// workgroup overlaps.  What overlaps them is OTHER workgroups on the same CU (four at 256 x 4 and <= 128 VGPRs, three at
// 256 x 8), so the kernel is written to keep the chain short and the registers few:
//   * three barriers of prologue: every cross-wave exchange (maximum of the tile maxima; totals of the scaled tile
//     weights, of the spacings, of the statistic sums; wave totals of the children's spacings) shares them;
//   * the parents' CDF segment is not rebuilt from the log-weights (exp + scan + three barriers per parent tile): the
//     launch that created the parents stored the tile-local scan of exp(lw - m_b) INSTEAD of the log-weights (8 B per
//     particle either way), so a parent tile costs one coalesced load, one fused multiply-add per entry, one barrier;
//     two parent tiles are handled per pass (a child tile usually descends from one or two);
//   * the tile bounds of a workgroup's uniforms by G / 64 independent LDS reads + ballots, not log2 G dependent probes;
//   * the standard normals of a batch of four children are drawn after its parent gathers are issued;
//   * two barriers of epilogue (grid_dev_epilogue).
// Traffic per particle-step: scan read 8 B (per parent tile visited: ~2 tiles per child tile on average) + record read
// + record write + scan write 8 B = the algorithmic 2 (n + 1 + h) w bytes of SURVEY 8(d) plus the second visit's 8 B.
// Measured steps of this kernel: profiles/r04_ab_grid_tile_classes.txt.
#pragma once
#include "pfg_grid_kernel.hpp"

namespace pfg {

// -DPFG_GRID_STAMPS (diagnostic builds, tools/grid_phases.py): thread 0 of the LAST tile of window 0 writes s_memtime at
// the phase boundaries of the launch into P.stamps[4..]
#ifdef PFG_GRID_STAMPS
#define PFG_GSTAMP(slot) do { if (P.stamps && tid == 0 && b == G - 1) P.stamps[4 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PFG_GSTAMP(slot) do { } while (0)
#endif

template <int NT, int PPT>
__host__ __device__ constexpr size_t grid_dev_lds_doubles(int G) {
    // pwn [G, even] | scn [G, even] | two tile buffers | bitmap [G / 64 + 1] | red [..] | tables
    return 2 * (size_t)((G + 2) & ~1) + 2 * (size_t)NT * PPT + (size_t)(G / 64 + 2) +
           (size_t)((NT / WAVE) * (1 + 2 * PPT + PFG_MAX_STAT + 4 + PFG_MAX_STAT) + 8) + (size_t)(TAB_E2_ACC + 2 * TAB_LG);
}

// waves per SIMD: 3 at 256 x 8, 4 at 256 x 4 -- 5 for the SVM's score-only twin (92 VGPRs, 28 KB of LDS: five workgroups per
// CU; 8 windows of 4 10^5 particles 63.5 -> 62.2 us per step).  GARCH / LGSSM spill 20-34 registers at 5 and lose a third.
constexpr int grid_dev_occ(int MODEL, int PPT, bool SCORE1) { return PPT == 4 ? (SCORE1 && MODEL == PFG_MODEL_SVM ? 5 : 4) : 3; }
// SCORE1 (launches with PFG_SMOOTHER_POYIADJIS_N: every window is NEMETH, lambduh = 1, score): the filter, the lambda != 1
// shrinkage and the other statistics compiled out -- 167 -> 143 VGPRs (256 x 8), 122 -> 92 (256 x 4), g1 +2.2 %, four
// windows of 4 10^5 particles +5.7 % (profiles/r04_ab_grid_tile_classes.txt).  A window that is not that estimator is
// flagged (GH_ERR) and gets NaNs from the finish kernel instead of another estimator's numbers.
template <int MODEL, int KERNEL, typename REAL, int NT, int PPT, int KMAX, bool SCORE1 = false>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(grid_dev_occ(MODEL, PPT, SCORE1), grid_dev_occ(MODEL, PPT, SCORE1))))
void pfg_grid_step_dev_kernel(const pfg_dev_problem *__restrict__ probs, int t) {
    constexpr int NS = ModelDims<MODEL>::NS, H = ModelDims<MODEL>::H, TILE = NT * PPT, NW = NT / WAVE;
    constexpr int REC = mem_rec_len<MODEL, REAL>();
    constexpr int RNG = PFG_RNG_DEVICE;
    extern __shared__ __align__(16) unsigned char smem[];
    const pfg_dev_problem &P = probs[blockIdx.y];
    const int N = P.N, T = P.T, t1 = P.t1, tL = P.tL, b = blockIdx.x, tid = threadIdx.x;
    if (t >= T) return;
    const GridLayout L = grid_layout<MODEL, REAL>(N, false);
    if (b >= L.G || L.PPT != PPT || L.NT != NT) return;
    const int G = L.G, lane = tid & (WAVE - 1), wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
    const int cp = t & 1, np = cp ^ 1;
    char *base = static_cast<char *>(P.scratch);
    gptr<const double> csc = global_ptr(reinterpret_cast<const double *>(base + grid_sel(L.cs, cp)));
    gptr<const REAL> recc = global_ptr(reinterpret_cast<const REAL *>(base + grid_sel(L.rec, cp)));
    gptr<REAL> recx = global_ptr(reinterpret_cast<REAL *>(base + grid_sel(L.rec, np)));
    gptr<const double> partc = global_ptr(reinterpret_cast<const double *>(base + grid_sel(L.part, cp)));
    double *head = reinterpret_cast<double *>(base + L.head);

    double *pwn = reinterpret_cast<double *>(smem);                      // [G]: where tile j's CDF ends (inclusive prefix of the scaled tile weights / W)
    double *scn = pwn + ((G + 2) & ~1);                                  // [G]: exp(m_j - m) / W, what turns tile j's local scan into CDF increments
    double *cdfl = scn + ((G + 2) & ~1);                                 // [2][TILE]
    double *red = cdfl + 2 * TILE + (G / 64 + 2);
    double *redM = red, *redSE = redM + NW;                              // prologue: [NW] maxima | [PPT NW] spacing wave totals
    double *redV = redSE + PPT * NW;                                     // [NW][3 + H]: v, e, e_own, S_h wave totals
    double *tabmem = red + (NW * (1 + 2 * PPT + PFG_MAX_STAT + 4 + PFG_MAX_STAT) + 8);
    Math<REAL, true> mth;
    mth.t.e2 = tabmem;
    mth.t.lg = reinterpret_cast<const double2 *>(tabmem + TAB_E2);
    mth.t.sc = reinterpret_cast<const double2 *>(tabmem + TAB_E2 + 2 * TAB_LG);

    if constexpr (SCORE1) {
        if (P.smoother != PFG_SMOOTHER_NEMETH || P.lambduh != 1.0 || P.stat != PFG_STAT_SCORE) {
            if (b == 0 && tid == 0) head[GH_ERR] = 1.0;
            return;
        }
    }
    const bool is_filter = !SCORE1 && (P.smoother == PFG_SMOOTHER_FILTER);
    const int stat = SCORE1 ? (int)PFG_STAT_SCORE : P.stat;
    const double lam_d = SCORE1 ? 1.0 : is_filter ? 0.0 : P.lambduh;
    const REAL lam = (REAL)lam_d, oml = (REAL)(1.0 - lam_d);
    const bool needS = is_filter || (lam_d != 1.0);
    const gptr<const double> yv = global_ptr(P.y);
    const gptr<const double> wv = global_ptr(P.weights);

    PFG_GSTAMP(0);
    // ---- prologue, phase 1: every load that depends on nothing is issued first; the arithmetic that depends on nothing
    // (model constants: a few ocml log / exp / divisions; the spacings and their wave scans) runs under their latency ----------
    const int K = (G + NT - 1) / NT;                                    // tiles per thread in the reduction (<= KMAX)
    if (K > KMAX) return;                                               // (the host picks KMAX from the batch's largest N)
    double pmq[KMAX], pWq[KMAX], pEq[KMAX];
#pragma unroll
    for (int q = 0; q < KMAX; ++q) {
        const int bb = tid * K + q;
        const bool ok = q < K && bb < G;
        pmq[q] = ok ? partc[bb] : -INFINITY;
        pWq[q] = ok ? partc[G + bb] : 0.0;
        pEq[q] = ok ? partc[2 * (size_t)G + bb] : 0.0;
    }
    const double e_extra = partc[7 * (size_t)G], e_own_tile = partc[2 * (size_t)G + b];
    LaneRng rng;
    {
        const uint4 s = reinterpret_cast<const uint4 *>(base + L.rng)[b * NT + tid];
        rng.s0 = s.x; rng.s1 = s.y; rng.s2 = s.z; rng.s3 = s.w;
    }
    {
        gptr<const double> tabg = global_ptr(reinterpret_cast<const double *>(base + L.tab));
        for (int q = tid; q < TAB_E2 + 2 * TAB_LG; q += NT) tabmem[q] = tabg[q];
        if (PPT == 8) { for (int q = tid; q < G / 64 + 1; q += NT) reinterpret_cast<unsigned long long *>(cdfl + 2 * TILE)[q] = 0ull; }
    }
    const double y_t = yv[t];
    const bool inside = (t >= t1) && (t < tL);
    const double wt = (inside && wv) ? wv[t - t1] : 1.0;
    const bool use_stat = inside && (stat != PFG_STAT_NONE);
    const bool last = t + 1 == T;
    // model constants: derived once per window by the init kernel (ocml log / exp / divisions: ~600 instructions that
    // every workgroup of every timestep would repeat), here a handful of scalar loads
    const Consts<REAL> c = *reinterpret_cast<const Consts<REAL> *>(base + L.consts);
    float incE[PPT];                                                    // in-wave running sums of the spacings: f32 values
    bool v[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        v[k] = b * TILE + k * NT + tid < N;
        const float ef = spacing_f32(rng.next());
        incE[k] = wave_incl_scan_f32(v[k] ? ef : 0.0f);                 // the instructions of grid_dev_epilogue: same bits
        if (lane == WAVE - 1) redSE[k * NW + wave] = (double)incE[k];
    }
    (void)rng.next();                                                   // the (N+1)-th spacing's word (in the total already)
    double mloc = -INFINITY;
#pragma unroll
    for (int q = 0; q < KMAX; ++q) mloc = pmq[q] > mloc ? pmq[q] : mloc;
    mloc = wave_max(mloc);
    if (lane == 0) redM[wave] = mloc;
    __syncthreads();                                                            // barrier P1
    PFG_GSTAMP(1);
    double m = redM[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) m = redM[w] > m ? redM[w] : m;
    m = uniform_f64(m);
    // ---- phase 2: scaled tile weights, their scan; totals of spacings and statistic sums -------------------------------
    double vq[KMAX], scq[KMAX], loc = 0.0, el = 0.0, eown = 0.0, sl[H];
#pragma unroll
    for (int h = 0; h < H; ++h) sl[h] = 0.0;
#pragma unroll
    for (int q = 0; q < KMAX; ++q) {
        const int bb = tid * K + q;
        vq[q] = 0.0; scq[q] = 0.0;
        if (q < K && bb < G) {
            scq[q] = mth.exp_acc((REAL)(pmq[q] - m));
            vq[q] = pWq[q] * scq[q];
            loc += vq[q];
            el += pEq[q];
            eown += bb < b ? pEq[q] : 0.0;
            if (needS) {
#pragma unroll
                for (int h = 0; h < H; ++h) sl[h] += partc[(size_t)(3 + h) * G + bb] * scq[q];
            }
        }
    }
    const double incV = wave_incl_scan(loc);
    el = wave_sum(el);
    eown = wave_sum(eown);
    if (lane == WAVE - 1) { redV[wave * (3 + H)] = incV; redV[wave * (3 + H) + 1] = el; redV[wave * (3 + H) + 2] = eown; }
    if (needS) {
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const double sh = wave_sum(sl[h]);
            if (lane == WAVE - 1) redV[wave * (3 + H) + 3 + h] = sh;
        }
    }
    __syncthreads();                                                            // barrier P2
    PFG_GSTAMP(2);
    double W = 0.0, woff = 0.0, Etot = 0.0, PE_own = 0.0, S[H];
#pragma unroll
    for (int h = 0; h < H; ++h) S[h] = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const double x = redV[w * (3 + H)];
        woff += w < wave ? x : 0.0;
        W += x;
        Etot += redV[w * (3 + H) + 1];
        PE_own += redV[w * (3 + H) + 2];
        if (needS) {
#pragma unroll
            for (int h = 0; h < H; ++h) S[h] += redV[w * (3 + H) + 3 + h];
        }
    }
    W = uniform_f64(W);
    const double invW = uniform_f64(1.0 / W);
    const double invEtot = uniform_f64(1.0 / (Etot + e_extra));
    PE_own = uniform_f64(PE_own);
    if (needS) {
#pragma unroll
        for (int h = 0; h < H; ++h) S[h] = uniform_f64(S[h] * invW);
    }
    {
        double run = woff + (incV - loc);
#pragma unroll
        for (int q = 0; q < KMAX; ++q) {
            const int bb = tid * K + q;
            run += vq[q];
            if (q < K && bb < G) { pwn[bb] = run * invW; scn[bb] = scq[q] * invW; }
        }
    }
    if (b == 0 && tid == 0) {
        // log-likelihood of the step these parents were weighted by (buffered_smoother.py:124-126), filter accumulators
        double ll = head[GH_LL];
        if (t > 0 && (t - 1) >= t1 && (t - 1) < tL) {
            const double wprev = wv ? wv[t - 1 - t1] : 1.0;
            ll += wprev * (m + log(W / (double)N));
            head[GH_LL] = ll;
        }
        if (P.trace_ll) P.trace_ll[t] = ll;
        if (is_filter && t > 0) {
#pragma unroll
            for (int h = 0; h < H; ++h) head[GH_FILT + h] += S[h];
        }
    }
    __syncthreads();                                                            // barrier P3: pwn / scn complete
    PFG_GSTAMP(3);
    // ---- the children's sorted uniforms and their parent tiles ------------------------------------------------------------
    // Every uniform of this tile lies in [PE_own, PE_own + E_b] / Etot, so every parent tile lies between the tiles of those
    // two bounds: two searches over all G tiles that the whole workgroup does identically (scalar loads of the bounds, one
    // dependent chain of log2 G probes each, interleaved), then a per-child search over that handful of tiles.
    double u[PPT];
    int pt[PPT];
    int tlo, thi;
    {
        const double ulo = PE_own * invEtot, uhi = (PE_own + e_own_tile) * invEtot;
        // count of tiles j <= G - 2 with pwn[j] <= bound (at most G - 1): pwn is non-decreasing, so the count is the sum of the
        // lanes' hits -- G / 64 independent LDS reads and ballots per wave instead of log2 G dependent round trips
        int lo0 = 0, lo1 = 0;
        for (int q = lane; q < ((G - 1 + WAVE - 1) & ~(WAVE - 1)); q += WAVE) {
            const double e = q < G - 1 ? pwn[q] : 2.0;
            lo0 += __popcll(__ballot(e <= ulo));
            lo1 += __popcll(__ballot(e <= uhi));
        }
        tlo = __builtin_amdgcn_readfirstlane(lo0);
        thi = __builtin_amdgcn_readfirstlane(lo1);
    }
    {
        // offsets of this thread's children among the tile's spacings (wave totals of phase 1), then the uniforms
        double run = 0.0;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            double offE = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                offE = w == wave ? run : offE;
                run += redSE[k * NW + w];
            }
            u[k] = (PE_own + offE + (double)incE[k]) * invEtot;
            pt[k] = tlo;
        }
    }
    for (int j = tlo; j < thi; ++j) {                          // usually one or two iterations
        const double e = pwn[j];
#pragma unroll
        for (int k = 0; k < PPT; ++k) pt[k] += e <= u[k] ? 1 : 0;
    }
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int i = b * TILE + k * NT + tid;
        pt[k] = v[k] ? pt[k] : -1;
        if (v[k] && P.trace_x && P.rec_ud) P.rec_ud[(size_t)t * N + i] = u[k];
    }
    PFG_GSTAMP(4);

    // ---- the parent tiles this tile's children descend from, in order: search only ------------------------------------------
    // Every parent tile lies in tlo .. thi and -- the uniforms being N + 1 spacings' running sums -- a tile of that range
    // without a child is a tile of (next to) no weight.  Two ways to walk them, chosen per tile class by measurement on one
    // box (profiles/r04_ab_grid_tile_classes.txt):
    //   256 x 4: walk the whole range (an interior tile of exactly zero weight is stepped over) -- no per-child bookkeeping,
    //            no barrier, the first pass starts as soon as tlo is known (N = 4 10^5, 4 windows: 37.2 -> 36.0 us per step);
    //   256 x 8: a bitmap of the tiles somebody descends from (one LDS atomic per wave and candidate tile) and a barrier,
    //            then only the marked tiles (g1, N = 10^6: 1052 windows/s against 1008 with the walk).
#ifndef PFG_GRID_BITMAP
#define PFG_GRID_BITMAP (PPT == 8)
#endif
    constexpr bool BITMAP = PFG_GRID_BITMAP;
    unsigned long long *bitmap = reinterpret_cast<unsigned long long *>(cdfl + 2 * TILE);     // [G / 64 + 1]
    if constexpr (BITMAP) {
        for (int j = tlo; j <= thi; ++j) {
            bool hit = false;
#pragma unroll
            for (int k = 0; k < PPT; ++k) hit = hit || pt[k] == j;
            if (__ballot(hit) != 0ull && lane == 0) atomicOr(&bitmap[j >> 6], 1ull << (j & 63));
        }
        __syncthreads();                                                        // barrier P4: bitmap complete
    }
    auto next_tile = [&](int from) {
        if constexpr (BITMAP) {
            int w = from >> 6;
            const int nwords = G / 64 + 1;
            unsigned long long bits = w < nwords ? bitmap[w] & (~0ull << (from & 63)) : 0ull;
            while (bits == 0ull && ++w < nwords) bits = bitmap[w];
            const int r = bits ? w * 64 + __builtin_ctzll(bits) : G;
            return __builtin_amdgcn_readfirstlane(r);
        } else {
            int j = from;
            while (j < thi && !(pwn[j] > (j > 0 ? pwn[j - 1] : 0.0))) ++j;
            return __builtin_amdgcn_readfirstlane(j <= thi ? j : G);
        }
    };
    // Two parent tiles per pass (a child tile usually descends from one or two): both scans are loaded together, turned
    // into CDF segments side by side in LDS, and every child searches its own parent tile's half -- once.
    int anc[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) anc[k] = 0;
    int pA = next_tile(tlo);
    PFG_GSTAMP(5);
    while (pA < G) {
        const int pB = next_tile(pA + 1);
        const int nvA = (N - pA * TILE) < TILE ? (N - pA * TILE) : TILE;
        const int nvB = pB < G ? ((N - pB * TILE) < TILE ? (N - pB * TILE) : TILE) : 0;
        double cA[PPT], cB[PPT];
        // (requesting the scans of tiles tlo, tlo + 1 ahead of barrier P4 -- they usually are pA, pB -- was measured: the 16
        // doubles held across the barrier cost more than the latency they hide, g1 1053 -> 1020 windows/s)
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            cA[k] = (k * NT + tid < nvA) ? csc[(size_t)pA * TILE + k * NT + tid] : 0.0;
            cB[k] = (k * NT + tid < nvB) ? csc[(size_t)pB * TILE + k * NT + tid] : 0.0;
        }
        // a tile's CDF segment: where the previous tile ends + the tile-local scan in units of the whole sum
        const double scA = scn[pA], pwA = pA > 0 ? pwn[pA - 1] : 0.0;
        const double scB = pB < G ? scn[pB] : 0.0, pwB = pB < G ? pwn[pB - 1] : 0.0;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            cdfl[k * NT + tid] = (k * NT + tid < nvA) ? fma(cA[k], scA, pwA) : 2.0;
            cdfl[TILE + k * NT + tid] = (k * NT + tid < nvB) ? fma(cB[k], scB, pwB) : 2.0;
        }
        __syncthreads();
        // the search variable is the LDS byte address (the probe offsets fold into the ds_read immediates); a child of the
        // second tile starts TILE entries in
        using lds_f64 = const __attribute__((address_space(3))) double;
        const uint32_t cdf_base = (uint32_t)(uintptr_t)(lds_f64 *)cdfl;
        uint32_t pos[PPT];
#pragma unroll
        for (int k = 0; k < PPT; ++k) pos[k] = cdf_base + (pt[k] == pB ? 8u * TILE : 0u);
#pragma unroll
        for (int step = TILE >> 1; step >= 1; step >>= 1) {
#pragma unroll
            for (int k = 0; k < PPT; ++k) pos[k] += (*(lds_f64 *)(uintptr_t)(pos[k] + 8u * (step - 1)) <= u[k]) ? 8u * step : 0u;
        }
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const bool inB = pt[k] == pB;
            const int nv = inB ? nvB : nvA;
            const int pl = (int)((pos[k] - cdf_base) >> 3) - (inB ? TILE : 0);
            const int pp = pl < nv - 1 ? pl : nv - 1;
            anc[k] = (pt[k] == pA || (inB && pB < G)) ? (inB ? pB : pA) * TILE + pp : anc[k];
        }
        pA = pB < G ? next_tile(pB + 1) : G;
        if (pA < G) __syncthreads();                        // a third parent tile (rare): the buffer is reused
    }
    PFG_GSTAMP(8);

    // ---- gather the parents (monotone addresses; four records in flight per thread), propose, weight, write ----------------
    REAL lwn[PPT];
    auto propagate = [&](auto stat_tag) {
        constexpr int STAT = decltype(stat_tag)::value;
#pragma unroll
        for (int k0 = 0; k0 < PPT; k0 += 4) {               // four records in flight per thread
            alignas(16) REAL r[4][REC];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) rec_load<REC, REAL>(r[kk], recc + (size_t)anc[k0 + kk] * REC);
            // this batch's standard normals, under the latency of its gathers (generator order: the spacings, then the
            // PPT normals in child order)
            REAL zb[4];
            mth.normal_pair(rng.next(), rng.next(), zb[0], zb[1]);
            mth.normal_pair(rng.next(), rng.next(), zb[2], zb[3]);
            if (P.trace_x && P.rec_z) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int i = b * TILE + (k0 + kk) * NT + tid;
                    if (v[k0 + kk]) P.rec_z[(size_t)t * N + i] = (double)zb[kk];
                }
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int k = k0 + kk;
                const int i = b * TILE + k * NT + tid;
                REAL xp[NS], sp[H], xn[NS], add[H], lwv;
#pragma unroll
                for (int d = 0; d < NS; ++d) xp[d] = r[kk][d];
#pragma unroll
                for (int h = 0; h < H; ++h) sp[h] = r[kk][NS + h];
                particle_step<MODEL, KERNEL, STAT, REAL>(c, mth, xp, (REAL)y_t, zb[kk], xn, lwv, add);
#pragma unroll
                for (int h = 0; h < H; ++h) {
                    const REAL av = use_stat ? add[h] * (REAL)wt : (REAL)0;
                    const REAL sm = (lam * sp[h] + oml * (REAL)S[h]) + av;      // pf.py:175-179 / :78-80
                    sp[h] = is_filter ? av : sm;
                }
                lwn[k] = lwv;
                if (v[k]) {
#pragma unroll
                    for (int d = 0; d < NS; ++d) r[kk][d] = xn[d];
#pragma unroll
                    for (int h = 0; h < H; ++h) r[kk][NS + h] = sp[h];
                    rec_store<REC, REAL>(recx + (size_t)i * REC, r[kk]);
                    if (P.trace_x) {
                        const size_t row = (size_t)(t + 1) * N + i;
                        if (P.trace_anc) P.trace_anc[(size_t)t * N + i] = anc[k];
#pragma unroll
                        for (int d = 0; d < NS; ++d) P.trace_x[row * NS + d] = (double)xn[d];
                        P.trace_logw[row] = (double)lwv;
                        if (P.trace_stats && !is_filter) {
#pragma unroll
                            for (int h = 0; h < H; ++h) P.trace_stats[row * H + h] = (double)sp[h];
                        }
                    }
                }
            }
        }
    };
    if (stat == PFG_STAT_SCORE) propagate(std::integral_constant<int, PFG_STAT_SCORE>{});
    else propagate(std::integral_constant<int, PFG_STAT_SUFF>{});

    PFG_GSTAMP(6);
    // ---- epilogue ----------------------------------------------------------------------------------------------------------
    const bool needS_next = needS || last;
    if (needS_next) __syncthreads();
    grid_dev_epilogue<NT, PPT, H, REC, NS, REAL>(base, L, np, b, N, lwn, rng, !last, needS_next, last, mth, red, tid);
    reinterpret_cast<uint4 *>(base + L.rng)[b * NT + tid] = make_uint4(rng.s0, rng.s1, rng.s2, rng.s3);
    PFG_GSTAMP(7);
}

}  // namespace pfg
