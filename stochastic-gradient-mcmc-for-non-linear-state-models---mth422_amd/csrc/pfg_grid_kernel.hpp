// libpfgrad device code: the WHOLE-GPU window for giant particle counts (MEM_MAX_N < N <= GRID_MAX_N).
//
// Every other kernel of the library runs one window in one workgroup (the T-loop never leaves a CU).  The reference's
// bias experiments call the hot-path entry with N = 10^6 particles on 48-step windows
// (nonlinear_ssm_pf_experiment_scripts/gradient_error_fig_scripts/svm_grad_compare.py:68-82, garch_grad_compare.py:86;
// particle_filters/pf.py:7-38 has no N limit): there ONE window is 80 MB of state per timestep, the byte roofline of
// SURVEY 8(d) is the real bound, and the particle axis of the window is what has to be spread over the 256 CUs.
//
// Structure: the particle axis is cut into tiles of TILE = 256 * PPT particles (PPT = 4 | 8), one workgroup per tile; one kernel
// launch per timestep (a dependent launch boundary is ~1.5 us, a software grid barrier 4-7 us, and a boundary also
// makes the other XCDs' L2 see this step's children).  State lives in a per-window HBM scratch:
//     lw[2][N]            log-weights, ping-pong by timestep parity (REPLAY; the device generator writes them for the last step only)
//     rec[2][N][REC]      particle records {x[NS], stats[H], pad} (array of records: a parent gather is 2-3 16-byte
//                         vectors from one place), ping-pong
//     partials[2][..]     per tile: maximum log-weight m_b, W_b = sum exp(lw - m_b), S_b[h] = sum stats_h exp(lw - m_b),
//                         E_b = total of the tile's exponential spacings (device generator); written by the launch that
//                         CREATES the particles (its epilogue), reduced redundantly by every workgroup of the next launch
//                         (G <= 2048 values): the step needs no separate reduction pass and no grid-wide synchronisation
//     rng[G*NT]           jsf32 lane-generator states between launches (device generator)
//     head[32]            log-likelihood, filter accumulators, tie margin
//     REPLAY only: cdf[N] (the reference's CDF, bit for bit: pfg_grid_cdf.hpp), coarse[C], walk list
//
// DEVICE generator (the throughput path, pfg_grid_step_dev_kernel in pfg_grid_dev_kernel.hpp): the resampling uniforms of a
// timestep are the order statistics of N i.i.d. uniforms (exponential spacings, as pf_big_kernel; multinomial resampling
// does not care which child gets which uniform), child r takes U_(r).  Children AND parents are then both sorted along the
// particle axis: the children of tile b descend from a contiguous run of parents, which the workgroup finds with a search
// over the tiles' cumulative weights and walks two tiles at a time -- per parent tile it loads the tile-local scan of
// exp(lw - m_b) that the launch which created the parents stored INSTEAD of their log-weights (cs[2][N], 8 B per particle
// either way; no CDF array in memory), turns it into the tile's CDF segment in LDS with one fused multiply-add per entry,
// searches it, gathers the parent records (monotone addresses: coalesced) and writes its children.  Traffic per
// particle-step: scan read (8 B per parent tile visited, ~2 visits) + record read + record write + scan write: the
// algorithmic 2 (n + 1 + h) w bytes of SURVEY 8(d) plus the second visit's 8 B (measured 1.20 x, profiles/hbm_traffic.json).
//
// REPLAY (the reference's own np.random stream: child i takes u[t][i], z[t][i]): the CDF has to be the reference's --
// cumsum(p) / cumsum(p)[-1] with the roundings of a SEQUENTIAL fp64 sum, see pfg_grid_cdf.hpp for why and how -- it is
// materialised in HBM by the four pfg_grid_cdf_*_kernel launches (pfg_grid_cdf.hpp), and the step kernel searches it with i.i.d.
// uniforms: a coarse table (every S-th entry, <= 16384 doubles) in LDS, then log2(S) probes in memory.
#pragma once
#include "pfg_big_kernel.hpp"

namespace pfg {

constexpr int GRID_MAX_N = 1 << 22;
constexpr int GRID_MAX_TILES = 2048;           // GRID_MAX_N / 2048
constexpr int GRID_COARSE_MAX = 16384;
constexpr int GRID_HEAD_DOUBLES = 32;
// REPLAY: 8-byte slots the multi-workgroup CDF kernels exchange through: chunk sums [512] | per 4096-block: sum of p, integer
// total, walk count, integer prefix, reference running sum, its integer position [6][1024] | s_last, walk total [8]
constexpr int GRID_CDF_CHUNKS = GRID_MAX_N / 8192, GRID_CDF_BLOCKS = GRID_MAX_N / 4096;
constexpr int GRID_CDFX_SLOTS = GRID_CDF_CHUNKS + 6 * GRID_CDF_BLOCKS + 8;
// head slots
constexpr int GH_LL = 0, GH_FILT = 1 /* ..4 */, GH_TIE = 5, GH_WALK = 6, GH_M = 7, GH_W = 8, GH_S = 9 /* ..12 */, GH_ERR = 13;

// Tile classes (256 threads per tile; the class is a function of N alone: every kernel of a window must agree on it).
//   N <= 2^19:          4 children per thread, 1024-particle tiles -- a timestep is one short round of workgroups, bound by
//                       the length of a workgroup's dependency chain; small tiles = more workgroups to overlap it;
//   2^19 < N <= 2^22:   8 children per thread, 2048-particle tiles -- the per-tile lists every workgroup reduces (all tiles'
//                       partials) and searches are half as long, and the per-thread fixed work of a timestep is spread over
//                       twice the particles (measured, profiles/r04_ab_grid_tile_classes.txt: the reference's ten-window
//                       computation at N = 10^6 0.41 -> 0.45 of the HBM peak, a lone window unchanged).
constexpr int GRID_NT = 256;
constexpr int GRID_SMALL_N = 1 << 19;
__host__ __device__ inline int grid_nt(int) { return GRID_NT; }
__host__ __device__ inline int grid_ppt(int N) { return N <= GRID_SMALL_N ? 4 : 8; }
// tiles per thread in the device-generator step kernel's reduction over the tile partials, at most: its loops are unrolled
// to this bound, so the N <= 2^20 launches (<= 512 tiles) run an instantiation with 2, larger ones with 8
__host__ __device__ inline int grid_kmax(int N) { return N <= (1 << 20) ? 2 : 8; }

struct GridLayout {
    int N, NT, PPT, TILE, G, C, S, PSTRIDE;          // C coarse entries of stride S (REPLAY); PSTRIDE doubles per partial parity
    size_t lw[2], rec[2], part[2], rng, head, cdf, coarse, walk_i, walk_p, walk_q, walk_s, cdfx, cs[2], tab, consts, bytes;
};

// a ping-pong offset by runtime parity WITHOUT indexing the struct's arrays (dynamic indexing would put the whole
// layout struct into scratch memory)
__host__ __device__ inline size_t grid_sel(const size_t (&a)[2], int q) { return q ? a[1] : a[0]; }

// partials of one parity: pm[G] | pW[G] | pE[G] | pS[4][G] | extra[8]  (extra[0] = the (N+1)-th spacing)
__host__ __device__ inline size_t grid_align(size_t x) { return (x + 255) & ~(size_t)255; }

template <int MODEL, typename REAL>
__host__ __device__ inline GridLayout grid_layout(int N, bool replay) {
    GridLayout L;
    L.N = N;
    L.NT = grid_nt(N);
    L.PPT = grid_ppt(N);
    L.TILE = L.NT * L.PPT;
    L.G = (N + L.TILE - 1) / L.TILE;
    int S = 64;
    while ((N + S - 1) / S > GRID_COARSE_MAX) S <<= 1;
    L.S = S;
    L.C = (N + S - 1) / S;
    L.PSTRIDE = 7 * L.G + 8;
    constexpr int REC = mem_rec_len<MODEL, REAL>();
    size_t o = 0;
    for (int q = 0; q < 2; ++q) { L.lw[q] = o; o = grid_align(o + (size_t)N * sizeof(REAL)); }
    for (int q = 0; q < 2; ++q) { L.rec[q] = o; o = grid_align(o + (size_t)N * REC * sizeof(REAL)); }
    for (int q = 0; q < 2; ++q) { L.part[q] = o; o = grid_align(o + (size_t)L.PSTRIDE * 8); }
    L.rng = o; o = grid_align(o + (size_t)L.G * L.NT * 16);
    L.head = o; o = grid_align(o + GRID_HEAD_DOUBLES * 8);
    L.cdf = L.coarse = L.walk_i = L.walk_p = L.walk_q = L.walk_s = L.cdfx = L.cs[0] = L.cs[1] = L.tab = L.consts = o;
    if (!replay) {
        // device generator: the tile-local inclusive scans of exp(lw - m_b) (what the next launch searches; written
        // instead of the log-weights on the hot path), and the math tables every launch loads into LDS
        for (int q = 0; q < 2; ++q) { L.cs[q] = o; o = grid_align(o + (size_t)N * 8); }
        L.tab = o; o = grid_align(o + (size_t)(TAB_E2_ACC + 2 * TAB_LG) * 8);
        L.consts = o; o = grid_align(o + sizeof(Consts<double>));      // model constants, derived once per window
    }
    if (replay) {
        L.cdf = o; o = grid_align(o + (size_t)N * 8);
        L.coarse = o; o = grid_align(o + (size_t)L.C * 8);
        L.walk_i = o; o = grid_align(o + (size_t)N * 4);
        L.walk_p = o; o = grid_align(o + (size_t)N * 8);
        L.walk_q = o; o = grid_align(o + (size_t)N * 8);
        L.walk_s = o; o = grid_align(o + (size_t)N * 8);
        L.cdfx = o; o = grid_align(o + (size_t)GRID_CDFX_SLOTS * 8);      // per-chunk / per-block exchange of the CDF kernels (pfg_grid_cdf.hpp)
    }
    L.bytes = o;
    return L;
}

// ---- small workgroup-level helpers (NT threads, NW = NT / 64 waves) -------------------------------------------------
template <int NW>
__device__ __forceinline__ double block_max_f64(double v, double *red, int wave, int lane) {
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double m = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) m = red[w] > m ? red[w] : m;
    return uniform_f64(m);
}
template <int NW>
__device__ __forceinline__ double block_sum_f64(double v, double *red, int wave, int lane) {
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[w];          // fixed order: every workgroup gets the same bits
    return uniform_f64(s);
}

// Inclusive scan of PPT values per thread over the tile in RANK order r = k * NT + tid (k-major), fixed summation order:
// the owner's epilogue (tile total W_b) and every rebuild of the tile by another workgroup produce the same bits.
// red: [PPT * NW + PPT * NW] doubles.  Returns the tile total.
template <int NT, int PPT>
__device__ __forceinline__ double tile_scan(const double (&p)[PPT], double (&c)[PPT], double *red, int wave, int lane) {
    constexpr int NW = NT / WAVE;
    double inc[PPT];
    __syncthreads();                                        // red may still be read from an earlier use
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        inc[k] = wave_incl_scan(p[k]);
        if (lane == WAVE - 1) red[k * NW + wave] = inc[k];
    }
    __syncthreads();
    if (wave == 0) {
        // exclusive offsets of the PPT * NW (<= 64) wave totals in (k, wave) order: one per lane
        const int idx = lane;
        const double v = idx < PPT * NW ? red[idx] : 0.0;
        const double in = wave_incl_scan(v);
        const double up = __shfl_up(in, 1);
        if (idx < PPT * NW) red[PPT * NW + idx] = lane == 0 ? 0.0 : up;
        if (lane == WAVE - 1) red[2 * PPT * NW] = in;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PPT; ++k) c[k] = inc[k] + red[PPT * NW + k * NW + wave];
    return uniform_f64(red[2 * PPT * NW]);
}

// What every workgroup of a launch derives from the G tile partials of the particles it is about to resample.
struct GridReduced {
    double m, W, invW;
    double S[PFG_MAX_STAT];
    double PE_own, invEtot;          // device generator: spacings before this tile, 1 / total of the N + 1 spacings
};

// pw_lds: [G + 1] exclusive prefix of W_b * exp(m_b - m) over the tiles (pw_lds[G] = W).  Fixed order.
template <int NT, typename MATH>
__device__ __forceinline__ GridReduced grid_reduce_partials(const double *__restrict__ part, int G, int b_own, bool needS, bool needE,
                                                            int H, const MATH &mth, double *pw_lds, double *red, int tid) {
    constexpr int NW = NT / WAVE;
    const int lane = tid & (WAVE - 1), wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
    const double *pm = part, *pW = part + G, *pE = part + 2 * (size_t)G, *pS = part + 3 * (size_t)G;
    GridReduced R;
    double ml = -INFINITY;
    for (int b = tid; b < G; b += NT) { const double v = pm[b]; ml = v > ml ? v : ml; }
    R.m = block_max_f64<NW>(ml, red, wave, lane);
    // thread `tid` owns the K consecutive tiles [tid K, tid K + K): local running sums, one wave scan, wave offsets
    const int K = (G + NT - 1) / NT;                        // <= 8
    double v[8], loc = 0.0, sl[PFG_MAX_STAT] = {0.0, 0.0, 0.0, 0.0}, el = 0.0, eown = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        v[q] = 0.0;
        const int b = tid * K + q;
        if (q < K && b < G) {
            const double sc = ::exp(pm[b] - R.m);
            v[q] = pW[b] * sc;
            loc += v[q];
            if (needS) {
                for (int h = 0; h < H; ++h) sl[h] += pS[(size_t)h * G + b] * sc;
            }
            if (needE) { const double e = pE[b]; el += e; eown += b < b_own ? e : 0.0; }
        }
    }
    (void)mth;
    const double inc = wave_incl_scan(loc);
    __syncthreads();
    if (lane == WAVE - 1) red[wave] = inc;
    __syncthreads();
    double woff = 0.0, tot = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { const double x = red[w]; woff += w < wave ? x : 0.0; tot += x; }
    double run = woff + (inc - loc);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int b = tid * K + q;
        if (q < K && b < G) pw_lds[b] = run;
        run += v[q];
    }
    if (tid == 0) pw_lds[G] = tot;
    R.W = uniform_f64(tot);
    R.invW = uniform_f64(1.0 / R.W);
#pragma unroll
    for (int h = 0; h < PFG_MAX_STAT; ++h) R.S[h] = 0.0;
    if (needS) {
        for (int h = 0; h < H; ++h) R.S[h] = block_sum_f64<NW>(sl[h], red, wave, lane) * R.invW;
    }
    R.PE_own = 0.0; R.invEtot = 0.0;
    if (needE) {
        const double et = block_sum_f64<NW>(el, red, wave, lane);
        R.PE_own = block_sum_f64<NW>(eown, red, wave, lane);
        R.invEtot = uniform_f64(1.0 / (et + part[7 * (size_t)G]));
    }
    __syncthreads();                                        // pw_lds complete
    return R;
}

// atomic minimum of a non-negative double (its bit pattern orders like an unsigned integer)
__device__ __forceinline__ void atomic_min_pos_f64(double *addr, double v) {
    atomicMin(reinterpret_cast<unsigned long long *>(addr), (unsigned long long)__double_as_longlong(v));
}

template <int MODEL, typename REAL, int RNG>
using GridMath = Math<REAL, (RNG == PFG_RNG_DEVICE)>;

template <int NT, int PPT, typename REAL, int RNG>
__host__ __device__ constexpr size_t grid_step_lds_bytes(int C) {
    // pw [G+1 <= 2049] | tile CDF [TILE] or coarse [C] | red [4 PPT NW + 16] | tables
    return (size_t)(GRID_MAX_TILES + 1) * 8 +
           (size_t)C * 8 +
           (size_t)(4 * PPT * (NT / WAVE) + 16 + PFG_MAX_STAT * (NT / WAVE)) * 8 + tab_bytes<REAL, RNG, (RNG == PFG_RNG_DEVICE)>();
}

// ------------------------------------------------------------------------------------------------------------------
// Epilogue shared by the init and step kernels: the partials of the particles this workgroup has just created (their
// log-weights in lwn[], statistics in sn[][]), for the launch that will resample them.
// ------------------------------------------------------------------------------------------------------------------
template <int NT, int PPT, int H, typename REAL, typename MATH>
__device__ __forceinline__ void grid_tile_partials(double *__restrict__ part, int G, int b, int N, const REAL (&lwn)[PPT],
                                                   const REAL (&sn)[PPT][H], bool needS, const MATH &mth, double *red, int tid) {
    constexpr int NW = NT / WAVE;
    const int lane = tid & (WAVE - 1), wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
    double ml = -INFINITY;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const bool v = b * (NT * PPT) + k * NT + tid < N;
        const double l = v ? (double)lwn[k] : -INFINITY;
        ml = l > ml ? l : ml;
    }
    const double mb = block_max_f64<NW>(ml, red, wave, lane);
    double p[PPT], c[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const bool v = b * (NT * PPT) + k * NT + tid < N;
        p[k] = v ? (double)mth.exp_acc((REAL)((double)lwn[k] - mb)) : 0.0;
    }
    const double Wb = tile_scan<NT, PPT>(p, c, red, wave, lane);
    if (tid == 0) { part[b] = mb; part[G + b] = Wb; }
    if (needS) {
#pragma unroll
        for (int h = 0; h < H; ++h) {
            double a = 0.0;
#pragma unroll
            for (int k = 0; k < PPT; ++k) a += (double)sn[k][h] * p[k];
            const double s = block_sum_f64<NW>(a, red, wave, lane);
            if (tid == 0) part[(size_t)(3 + h) * G + b] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Device-generator epilogue (init and step kernels): everything the NEXT launch needs from the particles this workgroup
// has just created -- tile maximum m_b, the tile-local inclusive scan of exp(lw - m_b) (stored per particle: the next
// launch searches it instead of re-scanning the log-weights), its total W_b, the total E_b of the tile's exponential
// spacings for the next resampling step (from a COPY of the lane generator: the step kernel regenerates the same words
// from the saved state), and -- only when the next launch needs the weighted statistic sums (Nemeth lambda < 1, filter,
// the last step) -- S_b, from the records re-read out of L2.  Two workgroup barriers on the hot path.
// red: [NW | 2 PPT NW | H NW] doubles
// ------------------------------------------------------------------------------------------------------------------
template <int NT, int PPT, int H, int REC, int NS, typename REAL, typename MATH>
__device__ __forceinline__ void grid_dev_epilogue(char *base, const GridLayout &L, int np, int b, int N, const REAL (&lwn)[PPT],
                                                  LaneRng rng, bool want_spacings, bool needS, bool store_lw, const MATH &mth,
                                                  double *red, int tid) {
    constexpr int NW = NT / WAVE, TILE = NT * PPT;
    const int lane = tid & (WAVE - 1), wave = __builtin_amdgcn_readfirstlane(tid / WAVE), G = L.G;
    double *part = reinterpret_cast<double *>(base + grid_sel(L.part, np));
    gptr<double> csx = global_ptr(reinterpret_cast<double *>(base + grid_sel(L.cs, np)));
    double *redP = red + NW, *redE = redP + PPT * NW, *redS = redE + PPT * NW;
    double ml = -INFINITY;
    bool v[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        v[k] = b * TILE + k * NT + tid < N;
        const double l = v[k] ? (double)lwn[k] : -INFINITY;
        ml = l > ml ? l : ml;
    }
    ml = wave_max(ml);
    if (lane == 0) red[wave] = ml;
    __syncthreads();                                                            // barrier E1
    double mb = red[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) mb = red[w] > mb ? red[w] : mb;
    mb = uniform_f64(mb);
    double p[PPT], incP[PPT];
    float extra = 0.0f;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        p[k] = v[k] ? (double)mth.exp_acc((REAL)((double)lwn[k] - mb)) : 0.0;
        incP[k] = wave_incl_scan(p[k]);
        if (lane == WAVE - 1) redP[k * NW + wave] = incP[k];
        if (want_spacings) {
            // spacings are O(1) each: their in-wave running sums (< 2^10) in f32 are good to 1e-5 of a spacing; the
            // cross-wave / cross-tile sums are f64.  The step kernel regenerates them with the same instructions.
            const float ef = spacing_f32(rng.next());
            const double incE = (double)wave_incl_scan_f32(v[k] ? ef : 0.0f);
            if (lane == WAVE - 1) redE[k * NW + wave] = incE;
        }
    }
    if (want_spacings) extra = spacing_f32(rng.next());         // spacing N + 1: part of the total only
    __syncthreads();                                                            // barrier E2
    // offsets of the (k, wave) totals in rank order, by every thread for itself (PPT NW <= 32 broadcast reads)
    double run = 0.0, off[PPT], etot = 0.0;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        off[k] = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            off[k] = w == wave ? run : off[k];
            run += redP[k * NW + w];
            if (want_spacings) etot += redE[k * NW + w];
        }
    }
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int i = b * TILE + k * NT + tid;
        if (v[k]) {
            csx[i] = off[k] + incP[k];
            if (store_lw) reinterpret_cast<REAL *>(base + grid_sel(L.lw, np))[i] = lwn[k];
        }
    }
    if (tid == 0) {
        part[b] = mb;
        part[G + b] = run;
        if (want_spacings) {
            part[2 * (size_t)G + b] = etot;
            if (b == 0) part[7 * (size_t)G] = (double)extra;
        }
    }
    if (needS) {
        // weighted statistic sums of the tile: the records this thread has just written, re-read (rare path)
        gptr<const REAL> recx = global_ptr(reinterpret_cast<const REAL *>(base + grid_sel(L.rec, np)));
        double a[H];
#pragma unroll
        for (int h = 0; h < H; ++h) a[h] = 0.0;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = b * TILE + k * NT + tid;
            if (v[k]) {
#pragma unroll
                for (int h = 0; h < H; ++h) a[h] += (double)recx[(size_t)i * REC + NS + h] * p[k];
            }
        }
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const double s = wave_sum(a[h]);
            if (lane == 0) redS[h * NW + wave] = s;
        }
        __syncthreads();
        if (tid < H) {
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) s += redS[tid * NW + w];
            part[(size_t)(3 + tid) * G + b] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// init: x0 (or warm start), zero statistics and log-weights, partials of step 0, generator states
// ------------------------------------------------------------------------------------------------------------------
template <int MODEL, int KERNEL, typename REAL, int RNG, int NT, int PPT>
__global__ __launch_bounds__(NT) void pfg_grid_init_kernel(const pfg_dev_problem *__restrict__ probs) {
    constexpr int NS = ModelDims<MODEL>::NS, H = ModelDims<MODEL>::H, TILE = NT * PPT;
    constexpr int REC = mem_rec_len<MODEL, REAL>();
    extern __shared__ __align__(16) unsigned char smem[];
    const pfg_dev_problem &P = probs[blockIdx.y];
    const int N = P.N, b = blockIdx.x, tid = threadIdx.x;
    const GridLayout L = grid_layout<MODEL, REAL>(N, RNG == PFG_RNG_REPLAY);
    if (b >= L.G || L.PPT != PPT || L.NT != NT) return;
    char *base = static_cast<char *>(P.scratch);
    gptr<REAL> lw = global_ptr(reinterpret_cast<REAL *>(base + L.lw[0]));
    gptr<REAL> rec = global_ptr(reinterpret_cast<REAL *>(base + L.rec[0]));
    double *part = reinterpret_cast<double *>(base + L.part[0]);
    double *head = reinterpret_cast<double *>(base + L.head);
    double *red = reinterpret_cast<double *>(smem);
    double *tabmem = red + (4 * PPT * (NT / WAVE) + 16 + PFG_MAX_STAT * (NT / WAVE));
    GridMath<MODEL, REAL, RNG> mth;
    mth.t.e2 = tabmem;
    mth.t.lg = reinterpret_cast<const double2 *>(tabmem + TAB_E2);
    mth.t.sc = reinterpret_cast<const double2 *>(tabmem + TAB_E2 + 2 * TAB_LG);
    if (tab_bytes<REAL, RNG, (RNG == PFG_RNG_DEVICE)>() > 0) tab_fill(tabmem, true, tid, NT);
    const Consts<REAL> c = make_consts<MODEL, REAL>(P.theta);
    const bool is_filter = (P.smoother == PFG_SMOOTHER_FILTER);
    if (b == 0 && tid < GRID_HEAD_DOUBLES) head[tid] = tid == GH_TIE ? 1.0 : 0.0;
    LaneRng rng = {};
    if (RNG == PFG_RNG_DEVICE) rng = lane_rng_init(P.seed, P.stream, P.step_ctr ? *P.step_ctr : 0ull, (uint32_t)(b * NT + tid));
    double pv = P.prior_var;
    if (MODEL == PFG_MODEL_GARCH && (P.flags & PFG_FLAG_GARCH_STATIONARY_PRIOR))
        pv = (double)c.alpha / (1.0 - (double)c.beta - (double)c.gamma);
    const double sd = sqrt(pv);
    REAL lwn[PPT], sn[PPT][H];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int i = b * TILE + k * NT + tid;
        REAL x[NS], l0 = (REAL)0;
#pragma unroll
        for (int d = 0; d < NS; ++d) x[d] = (REAL)0;
#pragma unroll
        for (int h = 0; h < H; ++h) sn[k][h] = (REAL)0;
        REAL za = (REAL)0, zb = (REAL)0;
        if (RNG == PFG_RNG_DEVICE && !P.init_x) mth.normal_pair(rng.next(), rng.next(), za, zb);
        lwn[k] = l0;
        if (i < N) {
            if (P.init_x) {
#pragma unroll
                for (int d = 0; d < NS; ++d) x[d] = (REAL)P.init_x[(size_t)i * NS + d];
                l0 = (REAL)P.init_logw[i];
                if (P.init_stats && !is_filter) {
#pragma unroll
                    for (int h = 0; h < H; ++h) sn[k][h] = (REAL)P.init_stats[(size_t)i * H + h];
                }
            } else {
                const double z = RNG == PFG_RNG_REPLAY ? P.z0[i] : (double)za;
                x[0] = (REAL)(P.prior_mean + sd * z);
                if (RNG == PFG_RNG_DEVICE && P.rec_z0) P.rec_z0[i] = z;
            }
            lwn[k] = l0;
            lw[i] = l0;
            alignas(16) REAL r[REC] = {};
#pragma unroll
            for (int d = 0; d < NS; ++d) r[d] = x[d];
#pragma unroll
            for (int h = 0; h < H; ++h) r[NS + h] = sn[k][h];
            rec_store<REC, REAL>(rec + (size_t)i * REC, r);
            if (P.trace_x) {
#pragma unroll
                for (int d = 0; d < NS; ++d) P.trace_x[(size_t)i * NS + d] = (double)x[d];
                P.trace_logw[i] = (double)l0;
                if (P.trace_stats && !is_filter) {
#pragma unroll
                    for (int h = 0; h < H; ++h) P.trace_stats[(size_t)i * H + h] = (double)sn[k][h];
                }
            }
        }
    }
    const bool needS = is_filter || P.lambduh != 1.0 || P.T == 0;
    if constexpr (RNG == PFG_RNG_DEVICE) {
        if (b == 0) {
            double *tabg = reinterpret_cast<double *>(base + L.tab);
            for (int q = tid; q < TAB_E2 + 2 * TAB_LG; q += NT) tabg[q] = tabmem[q];      // the step launches load these
            if (tid == 0) *reinterpret_cast<Consts<REAL> *>(base + L.consts) = c;         // ... and these (scalar loads)
        }
        __syncthreads();        // this thread's records: written above, re-read by the epilogue when needS
        grid_dev_epilogue<NT, PPT, H, REC, NS, REAL>(base, L, 0, b, N, lwn, rng, P.T > 0, needS, true, mth, red, tid);
        reinterpret_cast<uint4 *>(base + L.rng)[b * NT + tid] = make_uint4(rng.s0, rng.s1, rng.s2, rng.s3);
    } else {
        grid_tile_partials<NT, PPT, H, REAL>(part, L.G, b, N, lwn, sn, needS, mth, red, tid);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// one timestep t (0 <= t < T): resample the particles of parity t & 1, propose, weight, accumulate, write parity (t+1) & 1
// ------------------------------------------------------------------------------------------------------------------
template <int MODEL, int KERNEL, typename REAL, int RNG, int NT, int PPT>
__global__ __launch_bounds__(NT) void pfg_grid_step_kernel(const pfg_dev_problem *__restrict__ probs, int t) {
    constexpr int NS = ModelDims<MODEL>::NS, H = ModelDims<MODEL>::H, TILE = NT * PPT, NW = NT / WAVE;
    constexpr int REC = mem_rec_len<MODEL, REAL>();
    constexpr bool DEV = RNG == PFG_RNG_DEVICE;
    extern __shared__ __align__(16) unsigned char smem[];
    const pfg_dev_problem &P = probs[blockIdx.y];
    const int N = P.N, T = P.T, t1 = P.t1, tL = P.tL, b = blockIdx.x, tid = threadIdx.x;
    if (t >= T) return;
    const GridLayout L = grid_layout<MODEL, REAL>(N, !DEV);
    if (b >= L.G || L.PPT != PPT || L.NT != NT) return;
    const int G = L.G, lane = tid & (WAVE - 1), wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
    const int cp = t & 1, np = cp ^ 1;
    char *base = static_cast<char *>(P.scratch);
    gptr<const REAL> lwc = global_ptr(reinterpret_cast<const REAL *>(base + grid_sel(L.lw, cp)));
    gptr<REAL> lwx = global_ptr(reinterpret_cast<REAL *>(base + grid_sel(L.lw, np)));
    gptr<const REAL> recc = global_ptr(reinterpret_cast<const REAL *>(base + grid_sel(L.rec, cp)));
    gptr<REAL> recx = global_ptr(reinterpret_cast<REAL *>(base + grid_sel(L.rec, np)));
    const double *partc = reinterpret_cast<const double *>(base + grid_sel(L.part, cp));
    double *partx = reinterpret_cast<double *>(base + grid_sel(L.part, np));
    double *head = reinterpret_cast<double *>(base + L.head);

    double *pw = reinterpret_cast<double *>(smem);                       // [G + 1]
    double *tab2 = pw + (GRID_MAX_TILES + 1);                            // DEVICE: tile CDF [TILE] + bitmap; REPLAY: coarse [C]
    double *red = tab2 + (DEV ? TILE + GRID_MAX_TILES / 64 : L.C);
    double *tabmem = red + (4 * PPT * NW + 16 + PFG_MAX_STAT * NW);
    GridMath<MODEL, REAL, RNG> mth;
    mth.t.e2 = tabmem;
    mth.t.lg = reinterpret_cast<const double2 *>(tabmem + TAB_E2);
    mth.t.sc = reinterpret_cast<const double2 *>(tabmem + TAB_E2 + 2 * TAB_LG);
    if (tab_bytes<REAL, RNG, DEV>() > 0) tab_fill(tabmem, true, tid, NT);

    const bool is_filter = (P.smoother == PFG_SMOOTHER_FILTER);
    const int stat = P.stat;
    const double lam_d = is_filter ? 0.0 : P.lambduh;
    const REAL lam = (REAL)lam_d, oml = (REAL)(1.0 - lam_d);
    const bool needS_every = is_filter || (lam_d != 1.0);
    const Consts<REAL> c = make_consts<MODEL, REAL>(P.theta);
    const gptr<const double> yv = global_ptr(P.y);
    const gptr<const double> wv = global_ptr(P.weights);

    // ---- prologue: what the whole launch knows about the parents ---------------------------------------------------
    const GridReduced R = grid_reduce_partials<NT>(partc, G, b, needS_every, DEV, H, mth, pw, red, tid);
    if (b == 0 && tid == 0) {
        // log-likelihood of the step these parents were weighted by (buffered_smoother.py:124-126), filter accumulators
        double ll = head[GH_LL];
        if (t > 0 && (t - 1) >= t1 && (t - 1) < tL) {
            const double wprev = wv ? wv[t - 1 - t1] : 1.0;
            ll += wprev * (R.m + log(R.W / (double)N));
            head[GH_LL] = ll;
        }
        if (P.trace_ll) P.trace_ll[t] = ll;
        if (is_filter && t > 0) {
            for (int h = 0; h < H; ++h) head[GH_FILT + h] += R.S[h];
        }
    }
    const double y_t = yv[t];
    const bool inside = (t >= t1) && (t < tL);
    const double wt = (inside && wv) ? wv[t - t1] : 1.0;
    const bool use_stat = inside && (stat != PFG_STAT_NONE);

    REAL lwn[PPT], sn[PPT][H];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        lwn[k] = (REAL)0;
#pragma unroll
        for (int h = 0; h < H; ++h) sn[k][h] = (REAL)0;
    }
    // one child: parent record -> proposal, weight, statistic -> child record (registers + memory)
    auto propagate = [&](auto stat_tag, int k, int i, int a, REAL z) {
        constexpr int STAT = decltype(stat_tag)::value;
        alignas(16) REAL r[REC];
        rec_load<REC, REAL>(r, recc + (size_t)a * REC);
        REAL xp[NS], sp[H], xn[NS], add[H], lwv;
#pragma unroll
        for (int d = 0; d < NS; ++d) xp[d] = r[d];
#pragma unroll
        for (int h = 0; h < H; ++h) sp[h] = r[NS + h];
        particle_step<MODEL, KERNEL, STAT, REAL>(c, mth, xp, (REAL)y_t, z, xn, lwv, add);
#pragma unroll
        for (int h = 0; h < H; ++h) {
            const REAL av = use_stat ? add[h] * (REAL)wt : (REAL)0;
            const REAL sm = (lam * sp[h] + oml * (REAL)R.S[h]) + av;      // pf.py:175-179 / :78-80
            sp[h] = is_filter ? av : sm;
        }
#pragma unroll
        for (int q = 0; q < PPT; ++q) {
            if (q == k) {
                lwn[q] = lwv;
#pragma unroll
                for (int h = 0; h < H; ++h) sn[q][h] = sp[h];
            }
        }
        lwx[i] = lwv;
#pragma unroll
        for (int d = 0; d < NS; ++d) r[d] = xn[d];
#pragma unroll
        for (int h = 0; h < H; ++h) r[NS + h] = sp[h];
        rec_store<REC, REAL>(recx + (size_t)i * REC, r);
        if (P.trace_x) {
            const size_t row = (size_t)(t + 1) * N + i;
            if (P.trace_anc) P.trace_anc[(size_t)t * N + i] = a;
#pragma unroll
            for (int d = 0; d < NS; ++d) P.trace_x[row * NS + d] = (double)xn[d];
            P.trace_logw[row] = (double)lwv;
            if (P.trace_stats && !is_filter) {
#pragma unroll
                for (int h = 0; h < H; ++h) P.trace_stats[row * H + h] = (double)sp[h];
            }
        }
    };
    auto step_child = [&](int k, int i, int a, REAL z) {
        if (stat == PFG_STAT_SCORE) propagate(std::integral_constant<int, PFG_STAT_SCORE>{}, k, i, a, z);
        else propagate(std::integral_constant<int, PFG_STAT_SUFF>{}, k, i, a, z);
    };

    static_assert(!DEV, "the device-generator timestep is pfg_grid_step_dev_kernel (pfg_grid_dev_kernel.hpp)");
    {
        // ---- REPLAY: the reference's uniforms in index order against the reference's CDF (pfg_grid_cdf_kernel) --------
        const gptr<const double> cdf = global_ptr(reinterpret_cast<const double *>(base + L.cdf));
        const gptr<const double> coarse_g = global_ptr(reinterpret_cast<const double *>(base + L.coarse));
        double *coarse = tab2;
        const int C = L.C, S = L.S;
        for (int q = tid; q < C; q += NT) coarse[q] = coarse_g[q];
        __syncthreads();
        const gptr<const double> uv = global_ptr(P.u), zv = global_ptr(P.z);
        double tie = 1.0;
        int cpow = 1;
        while (cpow < C) cpow <<= 1;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = b * TILE + k * NT + tid;
            if (i < N) {
                const double u = uv[(size_t)t * N + i];
                const REAL z = (REAL)zv[(size_t)t * N + i];
                // block: number of coarse entries (= last CDF entry of each block of S) at or below u, at most C - 1
                int cb = 0;
                for (int step = cpow >> 1; step >= 1; step >>= 1) {
                    const int q = cb + step - 1;
                    cb += (q < C - 1 && coarse[q] <= u) ? step : 0;
                }
                cb = cb < C - 1 ? cb : C - 1;
                const int j0 = cb * S;
                const int nb = (N - j0) < S ? (N - j0) : S;
                int pos = 0;
                for (int step = S >> 1; step >= 1; step >>= 1) {
                    const int q = pos + step - 1;
                    pos += (q < nb && cdf[j0 + q] <= u) ? step : 0;
                }
                int a = j0 + pos;
                a = a < N - 1 ? a : N - 1;
                {
                    const double hi = cdf[a] - u;
                    const double lo = a > 0 ? u - cdf[a - 1] : 1.0;
                    const double mg = hi < lo ? hi : lo;
                    tie = mg < tie ? mg : tie;
                }
                step_child(k, i, a, z);
            }
        }
        tie = -wave_max(-tie);
        if (lane == 0) atomic_min_pos_f64(&head[GH_TIE], tie < 0.0 ? 0.0 : tie);
    }

    // ---- epilogue: partials of the children for the next launch -------------------------------------------------------
    const bool needS_next = needS_every || (t + 1 == T);
    grid_tile_partials<NT, PPT, H, REAL>(partx, G, b, N, lwn, sn, needS_next, mth, red, tid);
}

// ------------------------------------------------------------------------------------------------------------------
// finish (after the last step): W_T, the mean statistic (average_statistic, buffered_smoother.py:151-154), the last
// log-likelihood term; final particles on request.  One workgroup per window reduces, the grid copies.
// ------------------------------------------------------------------------------------------------------------------
template <int MODEL, typename REAL, int RNG, int NT, int PPT>
__global__ __launch_bounds__(NT) void pfg_grid_finish_kernel(const pfg_dev_problem *__restrict__ probs) {
    constexpr int NS = ModelDims<MODEL>::NS, H = ModelDims<MODEL>::H, TILE = NT * PPT, NW = NT / WAVE;
    constexpr int REC = mem_rec_len<MODEL, REAL>();
    extern __shared__ __align__(16) unsigned char smem[];
    const pfg_dev_problem &P = probs[blockIdx.y];
    const int N = P.N, T = P.T, b = blockIdx.x, tid = threadIdx.x;
    const GridLayout L = grid_layout<MODEL, REAL>(N, RNG == PFG_RNG_REPLAY);
    if (b >= L.G || L.PPT != PPT || L.NT != NT) return;
    const int cp = T & 1;
    char *base = static_cast<char *>(P.scratch);
    const bool is_filter = (P.smoother == PFG_SMOOTHER_FILTER);
    if (b == 0) {
        double *pw = reinterpret_cast<double *>(smem);
        double *red = pw + (GRID_MAX_TILES + 1);
        const double *partc = reinterpret_cast<const double *>(base + grid_sel(L.part, cp));
        double *head = reinterpret_cast<double *>(base + L.head);
        const Math<double, false> mth = {};
        const GridReduced R = grid_reduce_partials<NT>(partc, L.G, 0, true, false, H, mth, pw, red, tid);
        if (tid == 0) {
            double ll = head[GH_LL];
            if (T > 0 && (T - 1) >= P.t1 && (T - 1) < P.tL) {
                const double wprev = P.weights ? P.weights[T - 1 - P.t1] : 1.0;
                ll += wprev * (R.m + log(R.W / (double)N));
            }
            if (P.trace_ll) P.trace_ll[T] = ll;
            if (P.out) {
                for (int h = 0; h < PFG_MAX_STAT; ++h) P.out[h] = 0.0;
                for (int h = 0; h < H; ++h) P.out[h] = is_filter ? head[GH_FILT + h] + (T > 0 ? R.S[h] : 0.0) : R.S[h];
                P.out[4] = ll; P.out[5] = R.W; P.out[6] = R.m; P.out[7] = head[GH_TIE];
                if (head[GH_ERR] != 0.0) {          // a specialised step kernel refused this window (see SCORE1 in pfg_grid_dev_kernel.hpp)
                    for (int h = 0; h < PFG_OUT_DOUBLES; ++h) P.out[h] = __builtin_nan("");
                }
            }
        }
    }
    if (P.final_x) {
        gptr<const REAL> lwc = global_ptr(reinterpret_cast<const REAL *>(base + grid_sel(L.lw, cp)));
        gptr<const REAL> recc = global_ptr(reinterpret_cast<const REAL *>(base + grid_sel(L.rec, cp)));
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int i = b * TILE + k * NT + tid;
            if (i < N) {
#pragma unroll
                for (int d = 0; d < NS; ++d) P.final_x[(size_t)i * NS + d] = (double)recc[(size_t)i * REC + d];
                if (P.final_logw) P.final_logw[i] = (double)lwc[i];
                if (P.final_stats && !is_filter) {
#pragma unroll
                    for (int h = 0; h < H; ++h) P.final_stats[(size_t)i * H + h] = (double)recc[(size_t)i * REC + NS + h];
                }
            }
        }
    }
    (void)NW;
}

}  // namespace pfg
