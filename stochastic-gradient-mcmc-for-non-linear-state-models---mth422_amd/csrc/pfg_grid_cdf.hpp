// libpfgrad device code: the reference's resampling CDF for giant N, BIT FOR BIT.
//
// np.random.choice (particle_filters/pf.py:26-30 of the reference; legacy RandomState.choice) resamples against
//     p = exp(lw - max lw); p /= p.sum();  cdf = p.cumsum(); cdf /= cdf[-1];  idx = cdf.searchsorted(u, side='right')
// -- a SEQUENTIAL fp64 sum.  Its rounding errors accumulate like a random walk (~ sqrt(N) ulp); a tree-shaped parallel
// scan rounds differently, by delta ~ 3e-14 at N = 10^6, and a uniform lands within delta of one of the N CDF edges with
// probability 2 N delta: 2 N^2 delta ~ 0.06 flipped ancestors per timestep.  One flipped ancestor changes a weight, the
// edges behind it move by about one inter-edge gap, and HALF of the next step's children descend from a neighbour of
// their reference parent: the run is a different (equally valid) Monte-Carlo draw from there on.  Up to N = 16384 the
// kernels' tolerance-level CDF is good for ~1e-9 flips per run; at N = 10^6 seed-for-seed parity needs the SAME roundings.
//
// They can be had in parallel.  For doubles s, p >= 0 with s in the binade [2^e, 2^(e+1)) and s + p still inside it,
//     fl(s + p) = s + RN(p / u) u,   u = 2^(e-52) (the binade's ulp),
// unless p / u lies exactly half way between two integers (round-to-even then depends on the parity of s / u).  Inside a
// binade the sequential sum IS an integer prefix sum in units of u: associative, so any scan order gives the same
// result.  The running sum never decreases, so the binades are visited in order, and an approximate scan (relative error
// <= eta, any order) certifies for almost every step that s_{k-1} and s_k lie inside one binade ("safe" steps).  What
// is left -- the binade crossings, steps within eta of a power of two, exact ties, element 0: a few dozen per million --
// are the "walk" elements: ONE wave chains them with genuine fp64 additions, and between two of them everything is the
// integer prefix sum at the quantum of the running sum the earlier one left behind.  A step whose p is below half the
// quantum of the LOWER candidate binade changes nothing in either (fl(s + p) = s) and needs no certificate.
// tests/helpers/exact_cumsum_model.py is the NumPy model of this algorithm (checked against np.cumsum on adversarial
// inputs on the CPU); tests/test_gpu_grid.py checks the kernel's CDF against NumPy's bitwise.
//
// pfg_grid_cdf_kernel below is the algorithm in ONE workgroup per window (three streaming passes: weights; classify +
// integer scan; apply + normalise; the sequential part is the walk chain only) -- kept as the A/B and cross-check form
// (PFGRAD_CDF_SINGLE=1); what runs by default are the four kernels at the end of this file, which spread the particle
// axis over the GPU.  The log-likelihood / statistics do not go through here.
#pragma once
#include "pfg_grid_kernel.hpp"

namespace pfg {

constexpr int CDF_NT = 1024, CDF_PPT = 4, CDF_BLK = CDF_NT * CDF_PPT, CDF_NW = CDF_NT / WAVE;
constexpr int CDF_MAX_BLOCKS = GRID_MAX_N / CDF_BLK;      // 1024

template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_shr0_u64(unsigned long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)v, CTRL, 0xf, 0xf, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(v >> 32), CTRL, 0xf, 0xf, true);
    return ((unsigned long long)hi << 32) | lo;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, ROW_MASK, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, ROW_MASK, 0xf, false);
    return ((unsigned long long)hi << 32) | lo;
}
// inclusive prefix sum over the wave, modulo 2^64
__device__ __forceinline__ unsigned long long wave_incl_scan_u64(unsigned long long v) {
    v += dpp_shr0_u64<0x111>(v);
    v += dpp_shr0_u64<0x112>(v);
    v += dpp_shr0_u64<0x114>(v);
    v += dpp_shr0_u64<0x118>(v);
    v += dpp_u64<0x142, 0xa>(v);
    v += dpp_u64<0x143, 0xc>(v);
    return v;
}
__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int l) {
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// "binade" of a non-negative double: floor(log2 x) for normal x, -1023 for zero and the subnormals
__device__ __forceinline__ int cdf_binade(double x) { return (int)(((unsigned)__double2hiint(x) >> 20) & 0x7ffu) - 1023; }
// exponent of the quantum (ulp) of a binade
__device__ __forceinline__ int cdf_qexp(int e) { return (e < -1022 ? -1022 : e) - 52; }

template <int MODEL, typename REAL>
__global__ __launch_bounds__(CDF_NT) void pfg_grid_cdf_kernel(const pfg_dev_problem *__restrict__ probs, int t) {
    constexpr int NT = CDF_NT, PPT = CDF_PPT, NW = CDF_NW, BLK = CDF_BLK;
    __shared__ double red[4 * PPT * NW + 16];
    __shared__ unsigned long long redq[2 * PPT * NW + 2];
    __shared__ int redc[2 * PPT * NW + 2];
    __shared__ int wstart[CDF_MAX_BLOCKS + 2];
    __shared__ int widx[BLK];
    const pfg_dev_problem &P = probs[blockIdx.x];
    if (t >= P.T) return;
    const int N = P.N, tid = threadIdx.x, lane = tid & (WAVE - 1), wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
    const GridLayout L = grid_layout<MODEL, REAL>(N, true);
    char *base = static_cast<char *>(P.scratch);
    gptr<const REAL> lwc = global_ptr(reinterpret_cast<const REAL *>(base + grid_sel(L.lw, t & 1)));
    const double *partc = reinterpret_cast<const double *>(base + grid_sel(L.part, t & 1));
    double *head = reinterpret_cast<double *>(base + L.head);
    gptr<double> A = global_ptr(reinterpret_cast<double *>(base + L.cdf));            // p -> Q (as bits) -> cdf, in place
    gptr<unsigned long long> AQ = (gptr<unsigned long long>)A;
    gptr<double> coarse = global_ptr(reinterpret_cast<double *>(base + L.coarse));
    gptr<int> walk_i = global_ptr(reinterpret_cast<int *>(base + L.walk_i));
    gptr<double> walk_p = global_ptr(reinterpret_cast<double *>(base + L.walk_p));
    gptr<unsigned long long> walk_q = global_ptr(reinterpret_cast<unsigned long long *>(base + L.walk_q));
    gptr<double> walk_s = global_ptr(reinterpret_cast<double *>(base + L.walk_s));
    const int nblk = (N + BLK - 1) / BLK;

    // ---- pass 0: m = np.max(lw), p = exp(lw - m), W = np.sum(p) in NumPy's summation order ----------------------------
    // np.sum of a contiguous float64 array: the reduction runs over chunks of 8192 elements (the ufunc buffer size), each
    // chunk summed PAIRWISE (numpy/_core/src/umath/loops_utils.h.src: halves down to blocks of <= 128 elements, a block as
    // eight strided accumulators combined ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))), the chunk sums added up in order.  A full
    // chunk is a perfect binary tree over 64 blocks of 128: eight lanes hold the accumulators of a block, a wave eight
    // blocks, eight waves a chunk, and xor-butterflies add in exactly that tree.  The ragged last chunk is summed by the
    // same recursion on one thread.
    double ml = -INFINITY;
    for (int b = tid; b < L.G; b += NT) { const double v = partc[b]; ml = v > ml ? v : ml; }
    const double m = block_max_f64<NW>(ml, red, wave, lane);
    double W = 0.0;
    {
        constexpr int CH = 8192, PER_IT = 2 * CH;
        const int leaf = lane >> 3, jj = lane & 7;
        const int nchunks = (N + CH - 1) / CH;
        double wacc = 0.0;                                     // thread 0: the running np.sum
        for (int it = 0; it * PER_IT < N; ++it) {
            const int base = it * PER_IT + wave * 1024 + leaf * 128 + jj;
            double r = 0.0;
#pragma unroll 4
            for (int i = 0; i < 16; ++i) {
                const int g = base + 8 * i;
                if (g < N) {
                    const double p = ::exp((double)lwc[g] - m);
                    A[g] = p;
                    r = i == 0 ? p : r + p;
                }
            }
            r += __shfl_xor(r, 1); r += __shfl_xor(r, 2); r += __shfl_xor(r, 4);        // the block: ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))
            r += __shfl_xor(r, 8); r += __shfl_xor(r, 16); r += __shfl_xor(r, 32);      // eight blocks: 1024 elements
            __syncthreads();
            if (lane == 0) red[wave] = r;
            __syncthreads();
            if (tid == 0) {
                for (int c2 = 0; c2 < 2; ++c2) {
                    const int c = 2 * it + c2;
                    if (c >= nchunks) break;
                    const int n = N - c * CH < CH ? N - c * CH : CH;
                    double sc;
                    if (n == CH) {
                        const double *w8 = red + 8 * c2;
                        sc = ((w8[0] + w8[1]) + (w8[2] + w8[3])) + ((w8[4] + w8[5]) + (w8[6] + w8[7]));
                    } else {
                        // the ragged chunk: NumPy's recursion, iteratively (depth <= 7)
                        gptr<const double> a = (gptr<const double>)A + (size_t)c * CH;
                        int lo_[10], n_[10], st_[10], sp = 0;
                        double left_[10], ret = 0.0;
                        lo_[0] = 0; n_[0] = n; st_[0] = 0; sp = 1;
                        while (sp > 0) {
                            const int q = sp - 1;
                            if (st_[q] == 0) {
                                const int nn = n_[q], lo = lo_[q];
                                if (nn < 8) {
                                    ret = 0.0;
                                    for (int i = 0; i < nn; ++i) ret += a[lo + i];
                                    --sp;
                                } else if (nn <= 128) {
                                    double r8[8];
                                    for (int j = 0; j < 8; ++j) r8[j] = a[lo + j];
                                    int i = 8;
                                    for (; i < nn - (nn % 8); i += 8)
                                        for (int j = 0; j < 8; ++j) r8[j] += a[lo + i + j];
                                    ret = ((r8[0] + r8[1]) + (r8[2] + r8[3])) + ((r8[4] + r8[5]) + (r8[6] + r8[7]));
                                    for (; i < nn; ++i) ret += a[lo + i];
                                    --sp;
                                } else {
                                    int n2 = nn / 2; n2 -= n2 % 8;
                                    st_[q] = 1;
                                    lo_[sp] = lo; n_[sp] = n2; st_[sp] = 0; ++sp;
                                }
                            } else if (st_[q] == 1) {
                                int n2 = n_[q] / 2; n2 -= n2 % 8;
                                left_[q] = ret;
                                st_[q] = 2;
                                lo_[sp] = lo_[q] + n2; n_[sp] = n_[q] - n2; st_[sp] = 0; ++sp;
                            } else {
                                ret = left_[q] + ret;
                                --sp;
                            }
                        }
                        sc = ret;
                    }
                    wacc = c == 0 ? sc : wacc + sc;
                }
            }
        }
        __syncthreads();
        if (tid == 0) red[0] = wacc;
        __syncthreads();
        W = uniform_f64(red[0]);
        __syncthreads();
    }

    // ---- pass 1: p / W, approximate scan, classification, integer quanta, their scan, walk list ----------------------
    const double eta = (2.0 * (double)N + 4096.0) * 2.220446049250313e-16;
    double carry = 0.0;                      // approximate running sum before the block
    unsigned long long qcarry = 0ull;
    int wcarry = 0;
    for (int blk = 0; blk < nblk; ++blk) {
        double pn[PPT], inc[PPT], exc[PPT];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int g = blk * BLK + k * NT + tid;
            pn[k] = g < N ? A[g] / W : 0.0;
            inc[k] = wave_incl_scan(pn[k]);
            const double up = __shfl_up(inc[k], 1);
            exc[k] = lane == 0 ? 0.0 : up;
            if (lane == WAVE - 1) red[k * NW + wave] = inc[k];
        }
        __syncthreads();
        if (wave == 0) {
            const double v = lane < PPT * NW ? red[lane] : 0.0;
            const double in = wave_incl_scan(v);
            // exclusive (k, wave) offsets as sums of the earlier totals (NOT in - v: its cancellation error would be
            // relative to the larger total, and the certificates below need every running sum to a relative eta)
            const double up = __shfl_up(in, 1);
            if (lane < PPT * NW) red[PPT * NW + lane] = lane == 0 ? 0.0 : up;
            if (lane == WAVE - 1) red[2 * PPT * NW] = in;
        }
        __syncthreads();
        unsigned long long q[PPT], qin[PPT];
        bool wk[PPT];
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int g = blk * BLK + k * NT + tid;
            const double off = carry + red[PPT * NW + k * NW + wave];
            const double cprev = off + exc[k], ccur = off + inc[k];
            const int elo = cdf_binade(cprev * (1.0 - eta)), ehi = cdf_binade(ccur * (1.0 + eta));
            const int qe = cdf_qexp(ehi);
            const bool null = pn[k] == 0.0 || pn[k] < ldexp(1.0, cdf_qexp(elo) - 1);
            const double scaled = ldexp(pn[k], -qe);
            const bool tie = (scaled - floor(scaled)) == 0.5;
            const bool safe = g >= N || (g > 0 && (elo == ehi || null) && !tie);
            wk[k] = !safe;
            q[k] = safe && g < N ? (unsigned long long)rint(scaled) : 0ull;
            qin[k] = wave_incl_scan_u64(q[k]);
            if (lane == WAVE - 1) redq[k * NW + wave] = qin[k];
            const unsigned long long bal = __ballot(wk[k]);
            if (lane == 0) redc[k * NW + wave] = __popcll(bal);
        }
        __syncthreads();
        if (wave == 0) {
            const unsigned long long v = lane < PPT * NW ? redq[lane] : 0ull;
            const unsigned long long in = wave_incl_scan_u64(v);
            if (lane < PPT * NW) redq[PPT * NW + lane] = in - v;
            if (lane == WAVE - 1) redq[2 * PPT * NW] = in;
        } else if (wave == 1) {
            const int v = lane < PPT * NW ? redc[lane] : 0;
            int in = v;
#pragma unroll
            for (int d = 1; d < WAVE; d <<= 1) { const int o = __shfl_up(in, d); in += lane >= d ? o : 0; }
            if (lane < PPT * NW) redc[PPT * NW + lane] = in - v;
            if (lane == WAVE - 1) redc[2 * PPT * NW] = in;
        }
        __syncthreads();
        if (tid == 0) wstart[blk] = wcarry;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int g = blk * BLK + k * NT + tid;
            const unsigned long long Qk = qcarry + redq[PPT * NW + k * NW + wave] + qin[k];
            if (g < N) AQ[g] = Qk;
            const unsigned long long bal = __ballot(wk[k]);
            if (wk[k]) {
                const int pos = wcarry + redc[PPT * NW + k * NW + wave] + __popcll(bal & ((1ull << lane) - 1ull));
                walk_i[pos] = g;
                walk_p[pos] = pn[k];
                walk_q[pos] = Qk;
            }
        }
        carry = uniform_f64(carry + red[2 * PPT * NW]);
        qcarry += redq[2 * PPT * NW];
        wcarry += redc[2 * PPT * NW];
    }
    __syncthreads();
    if (tid == 0) { wstart[nblk] = wcarry; wstart[nblk + 1] = wcarry; }
    __syncthreads();
    const int nwalk = wcarry;                // >= 1: element 0

    // ---- the walk chain: genuine fp64 additions, in order, by wave 0 -------------------------------------------------
    if (wave == 0) {
        double s = 0.0;
        unsigned long long qprev = 0ull;
        for (int w0 = 0; w0 < nwalk; w0 += WAVE) {
            const int j = w0 + lane;
            const double pj = j < nwalk ? walk_p[j] : 0.0;
            const unsigned long long qj = j < nwalk ? walk_q[j] : 0ull;
            double sj = 0.0;
            const int cnt = nwalk - w0 < WAVE ? nwalk - w0 : WAVE;
            for (int l = 0; l < cnt; ++l) {
                const double pl = readlane_f64(pj, l);
                const unsigned long long ql = readlane_u64(qj, l);
                if (w0 + l == 0) {
                    s = pl;                                   // cumsum[0] = p[0]
                } else {
                    const int qe = cdf_qexp(cdf_binade(s));
                    const double before = s + ldexp((double)(ql - qprev), qe);   // exact: multiples of one quantum inside one binade
                    s = before + pl;                                              // the reference's rounding
                }
                qprev = ql;
                sj = l == lane ? s : sj;
            }
            if (j < nwalk) walk_s[j] = sj;
        }
        // the last element's running sum: cumsum[-1]
        const int qe = cdf_qexp(cdf_binade(s));
        const double slast = s + ldexp((double)(qcarry - qprev), qe);
        if (lane == 0) { red[0] = slast; if ((double)nwalk > head[GH_WALK]) head[GH_WALK] = (double)nwalk; }
    }
    __syncthreads();
    const double slast = red[0];

    // ---- pass 2: every running sum from the walk element at or before it; cdf = s / s_last ---------------------------
    const int S = L.S;
    for (int blk = 0; blk < nblk; ++blk) {
        const int w0 = wstart[blk], w1 = wstart[blk + 1];        // walk entries inside this block: [w0, w1)
        const int nin = w1 - w0;
        __syncthreads();
        for (int q = tid; q < nin; q += NT) widx[q] = walk_i[w0 + q];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int g = blk * BLK + k * NT + tid;
            if (g < N) {
                // last walk entry at or before g: w0 - 1 + (number of this block's entries with index <= g)
                int lo = 0, hi = nin;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (widx[mid] <= g) lo = mid + 1; else hi = mid; }
                const int j = w0 - 1 + lo;
                const double sb = walk_s[j];
                const unsigned long long qb = walk_q[j];
                const double s = sb + ldexp((double)(AQ[g] - qb), cdf_qexp(cdf_binade(sb)));
                const double cv = s / slast;
                A[g] = cv;
                if (((g + 1) % S) == 0 || g == N - 1) coarse[g / S] = cv;
            }
        }
    }
}


// ======================================================================================================================
// The same CDF, the particle axis spread over the GPU: four launches per timestep instead of one workgroup's three
// streaming passes (N = 10^6: 2.26 ms for the lone workgroup -- 735 block iterations of dependent scans and barriers).
//   A  pfg_grid_cdf_sum_kernel    one workgroup per 16384 particles: p = exp(lw - m) (stored), the np.sum chunk sums of its
//                                 two 8192-element chunks (NumPy's pairwise tree, as above), the sums of p over its four
//                                 4096-blocks (any order: they only feed the approximate scan)
//   B  pfg_grid_cdf_class_kernel  one workgroup per 4096-block: W = the chunk sums added up in order (every workgroup for
//                                 itself), the approximate running sum before the block from the block sums, then the
//                                 classification, the integer quanta and their BLOCK-LOCAL scan, the block's walk list in
//                                 the block's own segment of the walk arrays, its integer total and walk count
//   C  pfg_grid_cdf_chain_kernel  ONE wave per window: integer prefixes of the blocks, the walk chain (genuine fp64
//                                 additions, in order), per block the running sum and integer position of the last walk
//                                 element before it, s_last
//   D  pfg_grid_cdf_apply_kernel  one workgroup per 4096-block: every running sum from the walk element at or before it,
//                                 cdf = s / s_last, coarse table
// Which elements are "walk" elements may differ from the lone workgroup's choice (the approximate sums differ in the last
// places); the CDF does not: a certified step and a walked step give the same double.  tests/test_gpu_grid.py compares both
// with NumPy's bitwise (PFGRAD_CDF_SINGLE=1 launches the lone-workgroup kernel).
// cdfx slots (GridLayout::cdfx): chunk_sum[512] | blk_p | qtot | wcount | qprefix | ref_s | ref_q [1024 each] | s_last, nwalk
// ======================================================================================================================
struct CdfX {
    double *chunk_sum, *blk_p, *ref_s, *tail;
    unsigned long long *qtot, *qprefix, *ref_q;
    long long *wcount;
};
__device__ __forceinline__ CdfX cdfx_of(char *base, const GridLayout &L) {
    CdfX X;
    double *d = reinterpret_cast<double *>(base + L.cdfx);
    X.chunk_sum = d;
    X.blk_p = d + GRID_CDF_CHUNKS;
    X.qtot = reinterpret_cast<unsigned long long *>(d + GRID_CDF_CHUNKS + GRID_CDF_BLOCKS);
    X.wcount = reinterpret_cast<long long *>(d + GRID_CDF_CHUNKS + 2 * GRID_CDF_BLOCKS);
    X.qprefix = reinterpret_cast<unsigned long long *>(d + GRID_CDF_CHUNKS + 3 * GRID_CDF_BLOCKS);
    X.ref_s = d + GRID_CDF_CHUNKS + 4 * GRID_CDF_BLOCKS;
    X.ref_q = reinterpret_cast<unsigned long long *>(d + GRID_CDF_CHUNKS + 5 * GRID_CDF_BLOCKS);
    X.tail = d + GRID_CDF_CHUNKS + 6 * GRID_CDF_BLOCKS;
    return X;
}

// NumPy's pairwise sum of a[0 .. n) for n < 8192 (the ragged last chunk of np.sum), iteratively (depth <= 7); one thread
__device__ inline double cdf_pairwise_ragged(gptr<const double> a, int n) {
    int lo_[10], n_[10], st_[10], sp = 0;
    double left_[10], ret = 0.0;
    lo_[0] = 0; n_[0] = n; st_[0] = 0; sp = 1;
    while (sp > 0) {
        const int q = sp - 1;
        if (st_[q] == 0) {
            const int nn = n_[q], lo = lo_[q];
            if (nn < 8) {
                ret = 0.0;
                for (int i = 0; i < nn; ++i) ret += a[lo + i];
                --sp;
            } else if (nn <= 128) {
                double r8[8];
                for (int j = 0; j < 8; ++j) r8[j] = a[lo + j];
                int i = 8;
                for (; i < nn - (nn % 8); i += 8)
                    for (int j = 0; j < 8; ++j) r8[j] += a[lo + i + j];
                ret = ((r8[0] + r8[1]) + (r8[2] + r8[3])) + ((r8[4] + r8[5]) + (r8[6] + r8[7]));
                for (; i < nn; ++i) ret += a[lo + i];
                --sp;
            } else {
                int n2 = nn / 2; n2 -= n2 % 8;
                st_[q] = 1;
                lo_[sp] = lo; n_[sp] = n2; st_[sp] = 0; ++sp;
            }
        } else if (st_[q] == 1) {
            int n2 = n_[q] / 2; n2 -= n2 % 8;
            left_[q] = ret;
            st_[q] = 2;
            lo_[sp] = lo_[q] + n2; n_[sp] = n_[q] - n2; st_[sp] = 0; ++sp;
        } else {
            ret = left_[q] + ret;
            --sp;
        }
    }
    return ret;
}

template <int MODEL, typename REAL>
__global__ __launch_bounds__(CDF_NT) void pfg_grid_cdf_sum_kernel(const pfg_dev_problem *__restrict__ probs, int t) {
    constexpr int NT = CDF_NT, NW = CDF_NW, CH = 8192, PER_IT = 2 * CH;
    __shared__ double red[NW + 16];
    const pfg_dev_problem &P = probs[blockIdx.y];
    if (t >= P.T) return;
    const int N = P.N, it = blockIdx.x, tid = threadIdx.x, lane = tid & (WAVE - 1), wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
    if ((long long)it * PER_IT >= N) return;
    const GridLayout L = grid_layout<MODEL, REAL>(N, true);
    char *base = static_cast<char *>(P.scratch);
    gptr<const REAL> lwc = global_ptr(reinterpret_cast<const REAL *>(base + grid_sel(L.lw, t & 1)));
    const double *partc = reinterpret_cast<const double *>(base + grid_sel(L.part, t & 1));
    gptr<double> A = global_ptr(reinterpret_cast<double *>(base + L.cdf));
    const CdfX X = cdfx_of(base, L);
    double ml = -INFINITY;
    for (int b = tid; b < L.G; b += NT) { const double v = partc[b]; ml = v > ml ? v : ml; }
    const double m = block_max_f64<NW>(ml, red, wave, lane);
    const int leaf = lane >> 3, jj = lane & 7;
    const int nchunks = (N + CH - 1) / CH;
    const int e0 = it * PER_IT + wave * 1024 + leaf * 128 + jj;
    double r = 0.0;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const int g = e0 + 8 * i;
        if (g < N) {
            const double p = ::exp((double)lwc[g] - m);
            A[g] = p;
            r = i == 0 ? p : r + p;
        }
    }
    r += __shfl_xor(r, 1); r += __shfl_xor(r, 2); r += __shfl_xor(r, 4);        // the block: ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))
    r += __shfl_xor(r, 8); r += __shfl_xor(r, 16); r += __shfl_xor(r, 32);      // eight blocks: 1024 elements
    __syncthreads();
    if (lane == 0) red[wave] = r;
    __syncthreads();
    if (tid < 2) {
        const int c = 2 * it + tid;
        if (c < nchunks) {
            const int n = N - c * CH < CH ? N - c * CH : CH;
            double sc;
            if (n == CH) {
                const double *w8 = red + 8 * tid;
                sc = ((w8[0] + w8[1]) + (w8[2] + w8[3])) + ((w8[4] + w8[5]) + (w8[6] + w8[7]));
            } else {
                sc = cdf_pairwise_ragged((gptr<const double>)A + (size_t)c * CH, n);
            }
            X.chunk_sum[c] = sc;
        }
    } else if (tid >= WAVE && tid < WAVE + 4) {
        // sums of p over the 4096-blocks of this workgroup (4 waves of 1024 each): approximate scan only
        const int q = tid - WAVE, blk = 4 * it + q;
        if ((long long)blk * CDF_BLK < N) X.blk_p[blk] = (red[4 * q] + red[4 * q + 1]) + (red[4 * q + 2] + red[4 * q + 3]);
    }
}

template <int MODEL, typename REAL>
__global__ __launch_bounds__(CDF_NT) void pfg_grid_cdf_class_kernel(const pfg_dev_problem *__restrict__ probs, int t) {
    constexpr int NT = CDF_NT, PPT = CDF_PPT, NW = CDF_NW, BLK = CDF_BLK, CH = 8192;
    __shared__ double red[4 * PPT * NW + 16];
    __shared__ unsigned long long redq[2 * PPT * NW + 2];
    __shared__ int redc[2 * PPT * NW + 2];
    const pfg_dev_problem &P = probs[blockIdx.y];
    if (t >= P.T) return;
    const int N = P.N, blk = blockIdx.x, tid = threadIdx.x, lane = tid & (WAVE - 1), wave = __builtin_amdgcn_readfirstlane(tid / WAVE);
    if ((long long)blk * BLK >= N) return;
    const GridLayout L = grid_layout<MODEL, REAL>(N, true);
    char *base = static_cast<char *>(P.scratch);
    gptr<double> A = global_ptr(reinterpret_cast<double *>(base + L.cdf));
    gptr<unsigned long long> AQ = (gptr<unsigned long long>)A;
    gptr<int> walk_i = global_ptr(reinterpret_cast<int *>(base + L.walk_i));
    gptr<double> walk_p = global_ptr(reinterpret_cast<double *>(base + L.walk_p));
    gptr<unsigned long long> walk_q = global_ptr(reinterpret_cast<unsigned long long *>(base + L.walk_q));
    const CdfX X = cdfx_of(base, L);
    // W = np.sum(p): the chunk sums in order (<= 512 dependent additions; every workgroup for itself), and the approximate
    // running sum before this block
    const int nchunks = (N + CH - 1) / CH;
    double part = 0.0;
    for (int b = tid; b < blk; b += NT) part += X.blk_p[b];
    const double before = block_sum_f64<NW>(part, red, wave, lane);
    __syncthreads();
    if (tid == 0) {
        double wacc = X.chunk_sum[0];
        for (int c = 1; c < nchunks; ++c) wacc += X.chunk_sum[c];
        red[0] = wacc;
    }
    __syncthreads();
    const double W = uniform_f64(red[0]);
    __syncthreads();
    const double carry = before / W;
    const double eta = (2.0 * (double)N + 4096.0) * 2.220446049250313e-16;
    double pn[PPT], inc[PPT], exc[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int g = blk * BLK + k * NT + tid;
        pn[k] = g < N ? A[g] / W : 0.0;
        inc[k] = wave_incl_scan(pn[k]);
        const double up = __shfl_up(inc[k], 1);
        exc[k] = lane == 0 ? 0.0 : up;
        if (lane == WAVE - 1) red[k * NW + wave] = inc[k];
    }
    __syncthreads();
    if (wave == 0) {
        const double v = lane < PPT * NW ? red[lane] : 0.0;
        const double in = wave_incl_scan(v);
        const double up = __shfl_up(in, 1);
        if (lane < PPT * NW) red[PPT * NW + lane] = lane == 0 ? 0.0 : up;
    }
    __syncthreads();
    unsigned long long q[PPT], qin[PPT];
    bool wk[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int g = blk * BLK + k * NT + tid;
        const double off = carry + red[PPT * NW + k * NW + wave];
        const double cprev = off + exc[k], ccur = off + inc[k];
        const int elo = cdf_binade(cprev * (1.0 - eta)), ehi = cdf_binade(ccur * (1.0 + eta));
        const int qe = cdf_qexp(ehi);
        const bool null = pn[k] == 0.0 || pn[k] < ldexp(1.0, cdf_qexp(elo) - 1);
        const double scaled = ldexp(pn[k], -qe);
        const bool tie = (scaled - floor(scaled)) == 0.5;
        const bool safe = g >= N || (g > 0 && (elo == ehi || null) && !tie);
        wk[k] = !safe;
        q[k] = safe && g < N ? (unsigned long long)rint(scaled) : 0ull;
        qin[k] = wave_incl_scan_u64(q[k]);
        if (lane == WAVE - 1) redq[k * NW + wave] = qin[k];
        const unsigned long long bal = __ballot(wk[k]);
        if (lane == 0) redc[k * NW + wave] = __popcll(bal);
    }
    __syncthreads();
    if (wave == 0) {
        const unsigned long long v = lane < PPT * NW ? redq[lane] : 0ull;
        const unsigned long long in = wave_incl_scan_u64(v);
        if (lane < PPT * NW) redq[PPT * NW + lane] = in - v;
        if (lane == WAVE - 1) redq[2 * PPT * NW] = in;
    } else if (wave == 1) {
        const int v = lane < PPT * NW ? redc[lane] : 0;
        int in = v;
#pragma unroll
        for (int d = 1; d < WAVE; d <<= 1) { const int o = __shfl_up(in, d); in += lane >= d ? o : 0; }
        if (lane < PPT * NW) redc[PPT * NW + lane] = in - v;
        if (lane == WAVE - 1) redc[2 * PPT * NW] = in;
    }
    __syncthreads();
    const size_t seg = (size_t)blk * BLK;                       // this block's segment of the walk arrays
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int g = blk * BLK + k * NT + tid;
        const unsigned long long Qk = redq[PPT * NW + k * NW + wave] + qin[k];        // block-local
        if (g < N) AQ[g] = Qk;
        const unsigned long long bal = __ballot(wk[k]);
        if (wk[k]) {
            const int pos = redc[PPT * NW + k * NW + wave] + __popcll(bal & ((1ull << lane) - 1ull));
            walk_i[seg + pos] = g;
            walk_p[seg + pos] = pn[k];
            walk_q[seg + pos] = Qk;
        }
    }
    if (tid == 0) { X.qtot[blk] = redq[2 * PPT * NW]; X.wcount[blk] = (long long)redc[2 * PPT * NW]; }
}

template <int MODEL, typename REAL>
__global__ __launch_bounds__(WAVE) void pfg_grid_cdf_chain_kernel(const pfg_dev_problem *__restrict__ probs, int t) {
    constexpr int BLK = CDF_BLK;
    const pfg_dev_problem &P = probs[blockIdx.x];
    if (t >= P.T) return;
    const int N = P.N, lane = threadIdx.x;
    const GridLayout L = grid_layout<MODEL, REAL>(N, true);
    char *base = static_cast<char *>(P.scratch);
    double *head = reinterpret_cast<double *>(base + L.head);
    gptr<const double> walk_p = global_ptr(reinterpret_cast<const double *>(base + L.walk_p));
    gptr<const unsigned long long> walk_q = global_ptr(reinterpret_cast<const unsigned long long *>(base + L.walk_q));
    gptr<double> walk_s = global_ptr(reinterpret_cast<double *>(base + L.walk_s));
    const CdfX X = cdfx_of(base, L);
    const int nblk = (N + BLK - 1) / BLK;
    double s = 0.0;                              // the running sum at the last walk element (uniform)
    unsigned long long qprev = 0ull, Qbase = 0ull;   // its global integer position; integer total of the blocks before this group
    long long nwalk = 0;
    for (int b0 = 0; b0 < nblk; b0 += WAVE) {
        const int b = b0 + lane;
        const unsigned long long qt = b < nblk ? X.qtot[b] : 0ull;
        const int cnt = b < nblk ? (int)X.wcount[b] : 0;
        const unsigned long long qin = wave_incl_scan_u64(qt);
        const unsigned long long qpre = Qbase + (qin - qt);
        if (b < nblk) X.qprefix[b] = qpre;
        double rs = 0.0;
        unsigned long long rq = 0ull;
        const int nb = nblk - b0 < WAVE ? nblk - b0 : WAVE;
        for (int l = 0; l < nb; ++l) {
            // block b0 + l: what the elements before its first walk element refer to, then its walk elements in order
            rs = l == lane ? s : rs;
            rq = l == lane ? qprev : rq;
            const int c = __builtin_amdgcn_readlane(cnt, l);
            const unsigned long long qp = readlane_u64(qpre, l);
            const size_t seg = (size_t)(b0 + l) * BLK;
            for (int e0 = 0; e0 < c; e0 += WAVE) {            // 64 entries per coalesced load, chained through readlane
                const int ne = c - e0 < WAVE ? c - e0 : WAVE;
                const double pj = lane < ne ? walk_p[seg + e0 + lane] : 0.0;
                const unsigned long long qj = lane < ne ? walk_q[seg + e0 + lane] : 0ull;
                double sj = 0.0;
                for (int e = 0; e < ne; ++e) {
                    const double pl = readlane_f64(pj, e);
                    const unsigned long long ql = qp + readlane_u64(qj, e);
                    if (nwalk == 0) {
                        s = pl;                               // cumsum[0] = p[0]
                    } else {
                        const int qe = cdf_qexp(cdf_binade(s));
                        const double before = s + ldexp((double)(ql - qprev), qe);   // exact: multiples of one quantum inside one binade
                        s = before + pl;                                              // the reference's rounding
                    }
                    qprev = ql;
                    ++nwalk;
                    sj = e == lane ? s : sj;
                }
                if (lane < ne) walk_s[seg + e0 + lane] = sj;
            }
        }
        if (b < nblk) { X.ref_s[b] = rs; X.ref_q[b] = rq; }
        Qbase += readlane_u64(qin, WAVE - 1);
    }
    const int qe = cdf_qexp(cdf_binade(s));
    const double slast = s + ldexp((double)(Qbase - qprev), qe);       // cumsum[-1]
    if (lane == 0) {
        X.tail[0] = slast;
        X.tail[1] = (double)nwalk;
        if ((double)nwalk > head[GH_WALK]) head[GH_WALK] = (double)nwalk;
    }
}

template <int MODEL, typename REAL>
__global__ __launch_bounds__(CDF_NT) void pfg_grid_cdf_apply_kernel(const pfg_dev_problem *__restrict__ probs, int t) {
    constexpr int NT = CDF_NT, PPT = CDF_PPT, BLK = CDF_BLK;
    __shared__ int widx[BLK];
    const pfg_dev_problem &P = probs[blockIdx.y];
    if (t >= P.T) return;
    const int N = P.N, blk = blockIdx.x, tid = threadIdx.x;
    if ((long long)blk * BLK >= N) return;
    const GridLayout L = grid_layout<MODEL, REAL>(N, true);
    char *base = static_cast<char *>(P.scratch);
    gptr<double> A = global_ptr(reinterpret_cast<double *>(base + L.cdf));
    gptr<unsigned long long> AQ = (gptr<unsigned long long>)A;
    gptr<double> coarse = global_ptr(reinterpret_cast<double *>(base + L.coarse));
    gptr<const int> walk_i = global_ptr(reinterpret_cast<const int *>(base + L.walk_i));
    gptr<const unsigned long long> walk_q = global_ptr(reinterpret_cast<const unsigned long long *>(base + L.walk_q));
    gptr<const double> walk_s = global_ptr(reinterpret_cast<const double *>(base + L.walk_s));
    const CdfX X = cdfx_of(base, L);
    const int nin = (int)X.wcount[blk];
    const size_t seg = (size_t)blk * BLK;
    const unsigned long long qpre = X.qprefix[blk], refq = X.ref_q[blk];
    const double refs = X.ref_s[blk], slast = X.tail[0];
    for (int q = tid; q < nin; q += NT) widx[q] = walk_i[seg + q];
    __syncthreads();
    const int S = L.S;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int g = blk * BLK + k * NT + tid;
        if (g < N) {
            // last walk element at or before g: inside this block, or the one the chain recorded for the block
            int lo = 0, hi = nin;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (widx[mid] <= g) lo = mid + 1; else hi = mid; }
            const double sb = lo > 0 ? walk_s[seg + lo - 1] : refs;
            const unsigned long long qb = lo > 0 ? qpre + walk_q[seg + lo - 1] : refq;
            const double s = sb + ldexp((double)((qpre + AQ[g]) - qb), cdf_qexp(cdf_binade(sb)));
            const double cv = s / slast;
            A[g] = cv;
            if (((g + 1) % S) == 0 || g == N - 1) coarse[g / S] = cv;
        }
    }
}

}  // namespace pfg
