// libpfgrad device code: pf_reg_kernel, the LDS-resident particle filter (N <= 1024) incl. its
// PaRIS, systematic-resampling and O(N^2) instantiations.
#pragma once
#include "pfg_models.hpp"

namespace pfg {

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
__host__ __device__ __forceinline__ constexpr int cdf_phys(int i) { return i + (i >> 5); }
// FAST layout = LDS math tables + sentinel-padded, bank-conflict-free cdf with an unrolled search.
// Everything except the 1024-thread single-buffer variant (which spends all LDS on particles).
__host__ __device__ constexpr bool fast_layout(int NT, bool PP) { return PP || NT <= 512 || NT == 1024; }

// State arrays x[NS][.], stats[H][.] of a FAST layout are NL + pad elements apart (pad = 8 bytes).  With a stride of
// exactly NL = NT * PPT elements (a multiple of 512 bytes) the compiler fuses the gathers / stores of one particle's
// entries in two arrays into ds_read2st64_b64 / ds_write2st64_b64, which the LDS serves at HALF the rate of two
// ds_read_b64 (MI355X_MICROARCH.md, LDS table: 8 cycles per wave-instruction against 2 + 2; with the random
// addresses of a gather about 24 against 14).  A stride that is no multiple of 512 bytes keeps them apart.
// Device-generator kernels only: they are bound by LDS-array cycles (c2: 51.4 -> 47.0 ms with the pad).  The REPLAY
// kernels wait on their HBM streams instead and run 17 % SLOWER with twice the LDS instructions (768 windows of
// T = N = 1000: 6.8 ms fused, 8.0 ms padded), so they keep the fused form.  -DPFG_OPT_PADSTATE=0 restores it everywhere (A/B).
#ifndef PFG_OPT_PADSTATE
#define PFG_OPT_PADSTATE 1
#endif
template <typename REAL, int RNG> __host__ __device__ constexpr int state_pad() {
    return (PFG_OPT_PADSTATE && RNG == PFG_RNG_DEVICE) ? (int)(8 / sizeof(REAL)) : 0;
}

// workgroup barrier; a one-wave workgroup only needs its own LDS accesses kept in program order (the LDS
// executes a wave's instructions in order): no s_barrier, no drain of the LDS queue
template <int NW> __device__ __forceinline__ void block_sync() {
    if (NW == 1) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        __syncthreads();
    }
}
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
    // device generator (plain smoothers): 32-bit fixed-point CDF, see pf_reg_kernel
    constexpr bool BLK = FAST && RNG == PFG_RNG_DEVICE && MODE == MODE_PLAIN && (PPT & (PPT - 1)) == 0;
    const size_t NLS = NL + (FAST ? state_pad<REAL, RNG>() : 0);
    return (NC * (BLK ? 4 : 8) + 15) / 16 * 16 + (PP ? 2 : 1) * NLS * (ModelDims<MODEL>::NS + ModelDims<MODEL>::H) * sizeof(REAL) +
           (size_t)RegLayout<NT, PPT>::RED * 8 + tab_bytes<REAL, RNG, FAST>() +
           (PARIS ? NL * 8 + NL * 4 + 3 * NL * 4 : 0);   // PaRIS: parents' log-weights, fallback queue,
                                                         // two wave-queue arrays, accepted parents
}

// waves per SIMD the register allocator should aim for: what LDS lets a CU hold anyway.
// 256x4 fp64: ping-pong state is 80 KB/workgroup -> 2 workgroups (2 waves/SIMD); the single
// buffer is 50 KB -> 3, which is worth a few spilled registers (measured +15 %).
// One-wave workgroups (NT = 64): LDS admits many windows per CU, the register budget decides how many waves a SIMD
// holds (-DPFG_OCC64=n for A/B builds)
#ifndef PFG_OCC64
#define PFG_OCC64 4
#endif
// 4096 particle slots in fewer than 1024 threads (N <= 4096 state fills the LDS of a CU: ONE workgroup per CU
// whatever its thread count): 512 threads = 2 waves per SIMD and 256 VGPRs, 256 threads = 1 wave per SIMD and 512.
// (A/B instantiations of round 3, slower than 1024 x 4 and no longer built: see the variant table in pfgrad.hip.)
__host__ __device__ constexpr bool occ_lds4096(int NT, int PPT) { return NT * PPT == 4096 && NT < 1024; }
__host__ __device__ constexpr int occ_max(int NT, int PPT, size_t real, bool PP, bool dev4 = false) {
    if (occ_lds4096(NT, PPT)) return NT == 512 ? 2 : 1;
    if (NT == 64) return PFG_OCC64;
    return (NT >= 512 || PPT == 1 || dev4) ? 4 : ((PP && real == 8) ? 2 : 3);
}
__host__ __device__ constexpr int occ_min(int NT, int PPT, size_t real, bool PP, bool dev4 = false) {
    if (occ_lds4096(NT, PPT)) return NT == 512 ? 2 : 1;
    if (NT == 512) return 4;        // two 8-wave workgroups per CU
    return dev4 ? 4 : ((NT == 256 && PPT == 4 && !PP) ? 3 : 1);
}
// Device-generator units only (-DPFG_FAST_ALGEBRA; the REPLAY units keep the reference's operation
// order and phase structure).  Each can be switched off for A/B timing (-DPFG_OPT_x=0):
//  PFG_OPT_LAZYLL  the log-likelihood increment  w (m + log(W/N))  used to cost wave 0 an fp64 log and
//                  a division per timestep while the other waves waited at the next barrier; now wave 0
//                  parks (W, m, w) of step t in lane t % 64 and evaluates 64 steps at once (one table log
//                  per lane + one wave sum);
//  PFG_OPT_RCPW    1/W by v_rcp_f64 + two Newton steps instead of the IEEE division sequence.
// Measured and NOT kept (profiles/r02b_knockouts.txt): a wave-local maximum with the rescaling
// exp(m_w - m) folded into the prefix-sum exchange (drops the max barrier, lengthens the chain behind
// barrier 2: +4 %); the step's generator calls and Box-Muller issued between the search probes (E grows by
// what G shrinks: +4 %); a per-workgroup start-up stagger (0 %); jsf32 instead of xoshiro128++ (0 %); the
// model's closed-form upper bound of the log-weights as the shift (no max reduction, no max barrier, exact
// maximum only on underflow: SVM +10 %, GARCH 0 % -- the retry path costs 9 more spilled registers); the
// step's words / normals drawn behind barrier 2, next to the serial offsets chain (+10 %: 4-8 more live
// registers); search levels 1-3 compared in registers against 7 broadcast pivots (7 instead of 10 dependent
// LDS round trips: 0 % -- the search is issue-bound, not LDS-latency-bound); fewer, wider threads for the same
// four LDS-bound workgroups per CU (128 threads x 8 particles at 2 waves per SIMD and 227 VGPRs, no spills:
// +18 %; 64 x 16 at one wave per SIMD, no barriers left: +51 %) -- thread-level parallelism hides the LDS and
// fp64 latencies better than the same independent work inside one wave; the particle arrays on a 512-byte
// boundary so that their base folds into the ds_read2st64 / ds_write2st64 offsets (0 %); descriptor-field tests
// hoisted out of the loop and the observations held 64 steps at a time in the lanes of a register instead of a
// scalar load per step (+0.5 % / +2 %: 15 more spilled registers); the CDF as an implicit 4-ary search tree (a node
// = three pivots read with ds_read2_b32 + ds_read_b32, five dependent LDS round trips instead of ten, no padding to
// undo; bit-identical ancestors): SVM +6 %, GARCH +3 %, N = 100 0 %, N = 4000 -2 % -- three compares and two selects per
// level cost more than the round trips they save; the children of a step kept in registers and written behind
// barrier 1 of the NEXT step (every wave is past its gathers by then: three barriers per timestep on one buffer, the
// gathers' latency overlapped with the generator calls): SVM +5.5 % (32 spilled registers), GARCH +1 %, N = 4000 +2 %.
// TRACE (template parameter of pf_reg_kernel): the instantiation honours the trace_* / rec_* buffers of its
// descriptors (save_all trajectories, recorded generator draws: tests, elementwise statistics).  TRACE = false is
// the production twin of the plain device-generator kernels: the same code with every trace / record test compiled
// out of the T-loop -- each was a scalar load of a descriptor field plus a wait on the critical path of every
// timestep, and their address registers cost spills (measured: -5 % kernel time on BASELINE configs[1], -12 % on
// config 3, -6 % on config 1, -2 % on config 4).  tests/test_gpu_device_replay.py replays the TRACE = true twin from its
// recorded draws and asserts that the TRACE = false twin returns bitwise the same statistics for the same key.
// A/B experiment switch (diagnostic builds only; default = production):
//  PFG_EXP_PLAIN    compile the filter / lambda != 1 / no-statistic cases out (Poyiadjis O(N) score only)
#ifndef PFG_EXP_PLAIN
#define PFG_EXP_PLAIN 0
#endif
#define PFG_TR(p) (TRACE && (p))
#ifdef PFG_FAST_ALGEBRA
#ifndef PFG_OPT_LAZYLL
#define PFG_OPT_LAZYLL 1
#endif
#ifndef PFG_OPT_RCPW
#define PFG_OPT_RCPW 1
#endif
#ifndef PFG_OPT_SEL32
#define PFG_OPT_SEL32 1
#endif
#ifndef PFG_OPT_PIVOTS
#define PFG_OPT_PIVOTS 0
#endif
#else
#undef PFG_OPT_LAZYLL
#undef PFG_OPT_RCPW
#define PFG_OPT_LAZYLL 0
#define PFG_OPT_RCPW 0
#undef PFG_OPT_SEL32
#define PFG_OPT_SEL32 0
#undef PFG_OPT_PIVOTS
#define PFG_OPT_PIVOTS 0
#endif
// The device-generator SVM single-buffer workgroup needs 39.5 KB of LDS with the 32-bit CDF:
// FOUR workgroups fit a CU if the kernel stays within 128 VGPRs (34 spilled registers; measured
// +3.6 % workgroups per ms over occupancy 3).  -DPFG_OCC4=0 restores occupancy 3.
#ifndef PFG_OCC4
#define PFG_OCC4 1
#endif
// GARCH fp64 single buffer: six state arrays = 56.8 KB of LDS -> two workgroups per CU whatever
// the registers; give the allocator the 256 VGPRs that occupancy leaves (168 -> 32 spills)
__host__ __device__ constexpr bool occ_two(int MODEL, int NT, int PPT, size_t real, bool PP) {
    return MODEL == PFG_MODEL_GARCH && real == 8 && NT == 256 && PPT == 4 && !PP;
}
__host__ __device__ constexpr bool occ_dev4(int MODEL, int NT, int PPT, int RNG, bool PP, int MODE) {
    return PFG_OCC4 && MODEL == PFG_MODEL_SVM && NT == 256 && PPT == 4 && !PP && RNG == PFG_RNG_DEVICE && MODE == MODE_PLAIN;
}

template <int MODEL, int KERNEL, typename REAL, int NT, int PPT, int RNG, bool PP, int MODE = 0, bool TRACE = true, bool SCORE1 = false>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(occ_two(MODEL, NT, PPT, sizeof(REAL), PP) ? 2 : occ_min(NT, PPT, sizeof(REAL), PP, occ_dev4(MODEL, NT, PPT, RNG, PP, MODE)), occ_two(MODEL, NT, PPT, sizeof(REAL), PP) ? 2 : occ_max(NT, PPT, sizeof(REAL), PP, occ_dev4(MODEL, NT, PPT, RNG, PP, MODE))))) void pf_reg_kernel(const pfg_dev_problem *__restrict__ probs) {
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
    // SCORE1 (PFG_SMOOTHER_POYIADJIS_N launches of the 1024 x 4 unit): the Poyiadjis O(N) score only -- the filter, the
    // lambda != 1 shrinkage and the other statistics compiled out (config 4: 10.56 -> 10.07 ms, no spilled VGPR; the
    // 256 x 4 unit gains 0.5 %, the 512 x 2 one loses 2 %: profiles/r04_ab_score1_twin.txt).  A descriptor that is not
    // (NEMETH, lambduh = 1, SCORE) gets NaNs, loudly, instead of another estimator's numbers.
    constexpr bool ONLY_SCORE1 = PFG_EXP_PLAIN || SCORE1;
    if constexpr (SCORE1) {
        if (P.smoother != PFG_SMOOTHER_NEMETH || P.lambduh != 1.0 || P.stat != PFG_STAT_SCORE) {
            if (threadIdx.x < PFG_OUT_DOUBLES && P.out) P.out[threadIdx.x] = __builtin_nan("");
            return;
        }
    }
    const bool is_filter = !ONLY_SCORE1 && (P.smoother == PFG_SMOOTHER_FILTER);
    const int stat = ONLY_SCORE1 ? (int)PFG_STAT_SCORE : P.stat;
    const double lam_d = ONLY_SCORE1 ? 1.0 : is_filter ? 0.0 : ((P.smoother == PFG_SMOOTHER_PARIS || N2) ? 1.0 : P.lambduh);
    const REAL lam = (REAL)lam_d, oml = (REAL)(1.0 - lam_d);
    const bool needS_every = is_filter || (lam_d != 1.0);
    const gptr<const double> yv = global_ptr(P.y);
    const gptr<const double> wv = global_ptr(P.weights);
    const gptr<const double> uv = global_ptr(P.u);
    const gptr<const double> zv = global_ptr(P.z);

    constexpr bool FAST = fast_layout(NT, PP);
    constexpr bool TAB = FAST;
    // Device RNG only: the CDF is built in THREAD-major order (position tid*PPT + k <-> particle
    // k*NT + tid).  Multinomial resampling does not care how particles are labelled, and in this
    // order a thread's PPT weights are contiguous: one in-register prefix + ONE wave scan per
    // thread instead of PPT wave scans.  REPLAY keeps the reference's index order (parity).
    constexpr bool BLK = FAST && RNG == PFG_RNG_DEVICE && MODE == MODE_PLAIN && (PPT & (PPT - 1)) == 0;
    // SORTED (the 1024-thread device-generator variant, N <= 4096): the N resampling uniforms of a timestep are drawn
    // as the ORDER STATISTICS of N i.i.d. uniforms -- exponential spacings e_r = -log u_r, U_(r) = sum_{q<=r} e_q /
    // sum_{q<=N+1} e_q, by a second prefix scan that rides on the weight scan's barriers -- and child r takes U_(r)
    // (multinomial resampling does not care which child gets which uniform; children are exchangeable).  CDF and
    // ranks both run in thread-major order, so neighbouring lanes search neighbouring keys (coherent probes: LDS
    // broadcasts instead of bank conflicts) and gather neighbouring parents.  One such workgroup fills a CU's LDS, so
    // nothing else hides its LDS stalls: the knock-out with evenly spaced words was worth 15 % there (4 % on the
    // 256-thread SVM kernel, where the second scan costs more than that).  -DPFG_OPT_SORTED1024=0 restores i.i.d. words.
#ifndef PFG_OPT_SORTED1024
#define PFG_OPT_SORTED1024 1
#endif
    constexpr bool SORTED = PFG_OPT_SORTED1024 && BLK && NT == 1024 && PPT == 4 && !systematic && NW > 1;
    // PRIO: wave issue priority (s_setprio) by phase.  A timestep alternates between phases that are mostly LDS round
    // trips (E search, F gather) and phases that are mostly VALU work (A-D, G, H); the arbiter of a SIMD otherwise picks
    // by age.  256 x 4, four workgroups per CU in different phases: the VALU phases at priority 2 and E, F at 0 -- a wave
    // that is about to wait for the LDS anyway gives way -- 45.6 -> 44.8 ms per bench launch (-1.9 %; the same with 3
    // instead of 2; nothing if only G, H are raised).  1024 x 4, ONE workgroup per CU whose 16 waves are in the same
    // phase: the other way round (E, F at 2: the waves that reach the search first get their probes out) 11.26 ->
    // 11.03 ms (-2.0 %), and +0.4 % with the 256 x 4 setting.  512 x 2 (GARCH) and the one-wave kernels: 0 ... +4 % with
    // either, so none (profiles/r03_ab_wave_priority.txt).  The REPLAY instantiation of 256 x 4 (768 windows, three per CU):
    // 5.85 -> 5.26 ms with the 256 x 4 setting (5.52 with the opposite one).  -DPFG_OPT_PRIO=0 builds without.
#ifndef PFG_OPT_PRIO
#define PFG_OPT_PRIO 1
#endif
    constexpr int PRIO = !(PFG_OPT_PRIO && (BLK || (RNG == PFG_RNG_REPLAY && MODE == MODE_PLAIN)) && !PP && PPT == 4) ? 0 : (NT == 1024 ? 1 : (NT == 256 ? 2 : 0));
    constexpr int LOG_PPT = PPT == 1 ? 0 : (PPT == 2 ? 1 : (PPT == 4 ? 2 : (PPT == 8 ? 3 : 4)));
    static_assert(PPT <= 16, "LOG_PPT covers 1, 2, 4, 8, 16 particles per thread");
    // PP: the cdf is stored at physical index i + (i >> 5) (one pad slot per 32 entries): the
    // binary search probes at power-of-two strides, which would otherwise all hit one LDS bank
    // (measured: 720 conflict cycles per wave-timestep, i.e. all of SQ_LDS_BANK_CONFLICT).
    const int NC = FAST ? NT * PPT + NT * PPT / 32 : NL;
    double *cdf = reinterpret_cast<double *>(smem);
    // BLK: the uniforms carry 32 random bits, so the CDF is kept as 32-bit fixed point
    // (floor(cdf * 2^32)) and searched with the raw generator word: integer compares, half the
    // LDS bytes per probe, no u32 -> f64 conversion of the uniform.
    uint32_t *cdfu = reinterpret_cast<uint32_t *>(smem);
    REAL *buf0 = reinterpret_cast<REAL *>(smem + ((size_t)NC * (BLK ? 4 : 8) + 15) / 16 * 16);
    // stride of the state arrays: NL + one 8-byte pad (FAST layouts) -- see state_pad()
    const int NLS = NL + (FAST ? state_pad<REAL, RNG>() : 0);
    const size_t bufsz = (size_t)(NS + H) * NLS;
    REAL *cur = buf0, *nxt = PP ? buf0 + bufsz : buf0;
    // PAIRED (round 4; the device-generator production kernels in fp64 with an even record length: SVM, GARCH): the state
    // is stored as (NS + H) / 2 arrays of 16-byte PAIRS {component 2p, component 2p + 1} instead of NS + H arrays of
    // doubles.  The parent gathers -- 16 random ds_read_b64 per lane-timestep of the SVM kernel, 56 % of its LDS
    // bank-conflict cycles (profiles/r04_lds_conflict_split.txt) -- become 8 ds_read_b128, which use the full width of
    // the LDS; the children's stores stay lane-contiguous (ds_write_b128).  -DPFG_OPT_PAIRSTATE=0 restores the arrays.
#ifndef PFG_OPT_PAIRSTATE
#define PFG_OPT_PAIRSTATE 1
#endif
    constexpr bool PAIRED = PFG_OPT_PAIRSTATE && BLK && sizeof(REAL) == 8 && ((NS + H) % 2 == 0);
    // element index of component d of particle i
    auto sidx = [&](int d, int i) -> size_t {
        return PAIRED ? (size_t)(d >> 1) * (2 * (size_t)NLS) + 2 * (size_t)i + (size_t)(d & 1) : (size_t)d * NLS + (size_t)i;
    };
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
    if (FAST && !BLK) {
#pragma unroll
        for (int k = 0; k < PPT; ++k)
            if (k * NT + tid >= N) cdf[cdf_phys(k * NT + tid)] = 2.0;      // sentinel: never <= u
    }
    __syncthreads();

    const Consts<REAL> c = make_consts<MODEL, REAL>(P.theta);
    int np2 = 1;
    while (np2 < N) np2 <<= 1;
    // measurement hooks (pfg_dev_problem.stamps): shader-clock / 100 MHz stamps at start and end;
    // per-phase cycle sums only in diagnostic builds
    if (P.stamps && tid == 0) { P.stamps[0] = __builtin_amdgcn_s_memtime(); P.stamps[1] = __builtin_amdgcn_s_memrealtime(); }
#ifdef PFG_PHASE_STAMPS
    unsigned long long ph_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long ph_prev = __builtin_amdgcn_s_memtime();
#define PFG_PH(i) { const unsigned long long ph_now = __builtin_amdgcn_s_memtime(); ph_acc[i] += ph_now - ph_prev; ph_prev = ph_now; }
#define PFG_MARK(s)
#elif defined(PFG_ISA_MARKERS)
    // tools/isa_histogram.py: phase boundaries and rarely-executed regions as comments in the ISA listing
#define PFG_PH(i) asm volatile("; PFG_PHASE " #i);
#define PFG_MARK(s) asm volatile("; PFG_MARK " s);
#else
#define PFG_PH(i)
#define PFG_MARK(s)
#endif

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
    long long paris_cursor = 0;            // PaRIS in the reference's stream order: doubles of P.paris_stream consumed so far
    bool paris_overflow = false;           // ... and whether the stream ran out (the host retries with a longer one)
    // RAW (PFG_FLAG_PARIS_RAW_STREAM, PaRIS REPLAY): P.paris_stream is the window's WHOLE np.random stream as doubles
    // (RandomState.random_sample): the kernel takes from it, in np.random's order, the N resampling uniforms of a timestep,
    // its N normals and the backward sampling's uniforms -- one launch for a window whose consumption is data dependent.
    // The normals are NumPy's legacy Gaussians (legacy_gauss: Marsaglia's polar method on pairs of doubles, the second
    // variate of a pair cached for the next call): accept / reject is exact fp64 arithmetic (no contraction in the REPLAY
    // units), so the CONSUMPTION is the reference's to the double; the values go through this device's log (<= 1 ulp from
    // the host libm's), within the REPLAY tolerance.
    constexpr bool RAWCAP = MODE == MODE_PARIS && RNG == PFG_RNG_REPLAY;
    const bool raw = RAWCAP && (P.flags & PFG_FLAG_PARIS_RAW_STREAM) != 0 && P.paris_stream != nullptr;
    bool carry_has = false;                // a cached second variate is pending (workgroup-uniform); its value sits in red_W0[1]
    double *const zbuf = reinterpret_cast<double *>(paris_queue);          // [N] normals of the current call (queue scratch is free then)
    long long *const raw_slots = reinterpret_cast<long long *>(red_W0 + 2);  // [0] cut-off attempt of a call, [1] stream position of the cached pair
    [[maybe_unused]] auto legacy_normals = [&]() {
        const gptr<const double> strm = global_ptr(P.paris_stream);
        const long long cap = P.paris_stream_len;
        const unsigned long long ltm = (1ull << lane) - 1ull;
        int *const wc = reinterpret_cast<int *>(red_scan);                 // [PPT][NW] accepted attempts per (slot, wave)
        const int produced0 = carry_has ? 1 : 0;
        __syncthreads();                                                   // queue scratch / red_scan of the previous phase are done with
        if (carry_has && tid == 0) zbuf[0] = red_W0[1];
        const int need = (N - produced0 + 1) >> 1;                         // pairs to accept
        const long long base = paris_cursor;
        int acc = 0;
        for (int round = 0; acc < need && !paris_overflow; ++round) {
            double x1[PPT], x2[PPT], r2[PPT];
            bool fl[PPT];
            int rank[PPT];
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const long long p = base + 2 * ((long long)round * (NT * PPT) + k * NT + tid);
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
            int S = 0;
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
#pragma unroll
                for (int w = 0; w < NW; ++w) {
                    if (w == wave) rank[k] += S;
                    S += wc[k * NW + w];
                }
            }
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const int q = acc + rank[k];
                if (fl[k] && q < need) {
                    const double f = sqrt(-2.0 * log(r2[k]) / r2[k]);
                    const int i0 = produced0 + 2 * q;
                    zbuf[i0] = f * x2[k];                                  // returned by this call of legacy_gauss
                    const long long j = (long long)round * (NT * PPT) + k * NT + tid;
                    if (i0 + 1 < N) zbuf[i0 + 1] = f * x1[k];              // the cached one, returned by the next call
                    else { red_W0[1] = f * x1[k]; raw_slots[1] = base + 2 * j; }
                    if (q == need - 1) raw_slots[0] = j;
                }
            }
            acc += S;
            if (acc < need && base + 2 * ((long long)(round + 1) * (NT * PPT)) + 1 >= cap) paris_overflow = true;
            __syncthreads();
        }
        if (need > 0 && !paris_overflow) paris_cursor = base + 2 * (raw_slots[0] + 1);
        carry_has = ((N - produced0) & 1) != 0;
    };
    if (RAWCAP && raw && (P.flags & PFG_FLAG_PARIS_RAW_CARRY)) {          // the generator came with a cached Gaussian: stream[0]
        carry_has = true;
        if (tid == 0) red_W0[1] = P.paris_stream[0];
        paris_cursor = 1;
    }
    // ---- x0 (kernels.py:83-100, garch/kernels.py:7-18) or warm start ------------------
    {
        double pv = P.prior_var;
        if (MODEL == PFG_MODEL_GARCH && (P.flags & PFG_FLAG_GARCH_STATIONARY_PRIOR))
            pv = (double)c.alpha / (1.0 - (double)c.beta - (double)c.gamma);
        const double sd = sqrt(pv);
        REAL z0[PPT];
        if (RNG == PFG_RNG_DEVICE) draw_normals(z0);
        if constexpr (RAWCAP) { if (raw && !P.init_x) legacy_normals(); }
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
                    const double z = (RNG == PFG_RNG_REPLAY) ? ((RAWCAP && raw) ? zbuf[i] : P.z0[i]) : (double)z0[k];
                    x[0] = (REAL)(P.prior_mean + sd * z);
                    if (RNG == PFG_RNG_DEVICE && PFG_TR(P.trace_x) && P.rec_z0) P.rec_z0[i] = z;
                }
#pragma unroll
                for (int d = 0; d < NS; ++d) cur[sidx(d, i)] = x[d];
#pragma unroll
                for (int h = 0; h < H; ++h) cur[sidx((NS + h), i)] = s[h];
                if (PFG_TR(P.trace_x)) {
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
    constexpr bool LAZYLL = PFG_OPT_LAZYLL && TAB && sizeof(REAL) == 8;
    double ll_W = 1.0, ll_w = 0.0;          // LAZYLL: lane t % 64 of wave 0 holds step t's (W, w, m)
    float ll_m = 0.0f;
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
        [[maybe_unused]] double uu_raw[PPT];
        [[maybe_unused]] REAL zz_raw[PPT];
        if constexpr (RAWCAP) {
            if (raw && t < T) {
                // np.random's order within a timestep: N uniforms (np.random.choice), N normals (Kernel.rv), then the
                // backward sampling's draws (at the end of this iteration)
                const gptr<const double> strm = global_ptr(P.paris_stream);
                if (paris_cursor + N > P.paris_stream_len) paris_overflow = true;
#pragma unroll
                for (int k = 0; k < PPT; ++k) uu_raw[k] = paris_overflow ? 0.5 : strm[paris_cursor + own[k]];
                if (!paris_overflow) paris_cursor += N;
                legacy_normals();
#pragma unroll
                for (int k = 0; k < PPT; ++k) zz_raw[k] = (REAL)zbuf[own[k]];
            }
        }
        // ---- (A) block max of the current log weights  (log_normalize, pf.py:374-377) ----
        float ml = (float)lw[0];
#pragma unroll
        for (int k = 1; k < PPT; ++k) ml = fmaxf(ml, (float)lw[k]);
        ml = wave_max(ml);
        if (NW > 1 && lane == 0) red_maxf[wave] = ml;
        PFG_PH(0)
        block_sync<NW>();                                                       // barrier 1
        PFG_PH(1)
        {
            float mm = NW > 1 ? red_maxf[0] : ml;     // one wave: its maximum IS the block maximum, no LDS round trip
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
            PFG_MARK("cold needS")
#pragma unroll
            for (int h = 0; h < H; ++h) {
                double part = 0.0;
#pragma unroll
                for (int k = 0; k < PPT; ++k) part += (double)cur[sidx((NS + h), own[k])] * cs[k];
                part = wave_sum(part);
                if (lane == 0) red_S[h * NW + wave] = part;
            }
        }
        double wave_inc = 0.0;          // BLK: this wave's inclusive scan (lane 63 = the wave total)
        float es[PPT];                  // SORTED: inclusive prefix of the exponential spacings of this thread's children (in-wave
                                        // sums < 2^9 in f32: good to 1e-5 of a spacing, half the registers across barrier 2 and a
                                        // 6-instruction wave scan; the cross-wave offsets and the normalisation are f64)
        uint32_t us[PPT];               // SORTED: the sorted uniforms as 32-bit fixed point
        if (BLK) {
#pragma unroll
            for (int k = 1; k < PPT; ++k) cs[k] += cs[k - 1];
            const double inc = wave_incl_scan(cs[PPT - 1]);
            const double exc = inc - cs[PPT - 1];
#pragma unroll
            for (int k = 0; k < PPT; ++k) cs[k] += exc;
            if (NW > 1 && lane == WAVE - 1) red_scan[wave] = inc;
            wave_inc = inc;
            if (SORTED && t < T) {
                // exponential spacings of this thread's (valid) children, thread-local prefix + one wave scan; the
                // (N+1)-th spacing rides in the last wave's total
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    const float e = spacing_f32(rng.next());
                    es[k] = (valid[k] ? e : 0.0f) + (k > 0 ? es[k - 1] : 0.0f);
                }
                const float eincf = wave_incl_scan_f32(es[PPT - 1]);
                const float eexc = eincf - es[PPT - 1];
#pragma unroll
                for (int k = 0; k < PPT; ++k) es[k] += eexc;
                if (lane == WAVE - 1) {
                    double einc = (double)eincf;
                    if (wave == NW - 1) einc += (double)spacing_f32(rng.next());
                    red_scan[NW + wave] = einc;
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                cs[k] = wave_incl_scan(cs[k]);
                if (lane == WAVE - 1) red_scan[k * NW + wave] = cs[k];
            }
        }
        if (RNG != PFG_RNG_REPLAY && systematic && tid == 0) red_W0[0] = u01_32(rng.next());
        // this step's randomness: REPLAY loads are issued here so that their latency overlaps the
        // barrier; device draws happen right before their use (keeps register pressure down).  (Loading a timestep
        // ahead was measured, round 3: a lone window 2.90 -> 2.85 ms, 768 windows 6.85 -> 7.60 ms -- the registers
        // cost more than the latency; the REPLAY kernel's distance to the device units is its fp64 CDF and search.)
        double uu[PPT];
        REAL zz[PPT];
        if (t < T && RNG == PFG_RNG_REPLAY) {
            if (RAWCAP && raw) {
#pragma unroll
                for (int k = 0; k < PPT; ++k) { uu[k] = uu_raw[k]; zz[k] = zz_raw[k]; }
            } else {
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    uu[k] = uv[(size_t)t * N + own[k]];
                    zz[k] = (REAL)zv[(size_t)t * N + own[k]];
                }
            }
        }
        PFG_PH(2)
        block_sync<NW>();                                                       // barrier 2
        PFG_PH(3)
        if (BLK && NW == 1) {
            // one wave: no other totals to add, W is the scan's last lane
            W = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(wave_inc), WAVE - 1),
                                 __builtin_amdgcn_readlane(__double2loint(wave_inc), WAVE - 1));
        } else if (BLK) {
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
            if (SORTED && t < T) {
                const double totE = (lane < NW) ? red_scan[NW + lane] : 0.0;
                double incE = totE;
                incE += dpp_shr0_f64<0x111>(incE);
                incE += dpp_shr0_f64<0x112>(incE);
                incE += dpp_shr0_f64<0x114>(incE);
                incE += dpp_shr0_f64<0x118>(incE);
                const double excE = incE - totE;
                const double offE = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(excE), wave),
                                                     __builtin_amdgcn_readlane(__double2loint(excE), wave));
                const double Etot = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(incE), NW - 1),
                                                     __builtin_amdgcn_readlane(__double2loint(incE), NW - 1));
                double r = __builtin_amdgcn_rcp(Etot);
                r = fma(fma(-Etot, r, 1.0), r, r);
                r = fma(fma(-Etot, r, 1.0), r, r);
                const double fe = uniform_f64(r * 4294967296.0);
#pragma unroll
                for (int k = 0; k < PPT; ++k) us[k] = cvt_u32_sat(((double)es[k] + offE) * fe);
            }
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
        double invW;
        if (PFG_OPT_RCPW) {
            double r = __builtin_amdgcn_rcp(W);
            r = fma(fma(-W, r, 1.0), r, r);
            r = fma(fma(-W, r, 1.0), r, r);
            invW = uniform_f64(r);
        } else {
            invW = uniform_f64(1.0 / W);
        }
        if (needS) {
            PFG_MARK("cold needS")
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
            PFG_MARK("w0.25 wave0-loglik")
            const bool counts = t > 0 && (t - 1) >= t1 && (t - 1) < tL;
            if (LAZYLL) {
                const bool mine = lane == (t & (WAVE - 1));
                ll_W = mine ? W : ll_W;
                ll_w = mine ? (counts ? wt_prev : 0.0) : ll_w;
                ll_m = mine ? (float)m : ll_m;          // m is an f32 value
                if ((t & (WAVE - 1)) == WAVE - 1 || t == T || PFG_TR(P.trace_ll)) {
                    PFG_MARK("cold loglik-flush")
                    const double term = ll_w * ((double)ll_m + (double)mth.log((REAL)(ll_W * invN)));
                    ll = uniform_f64(ll + wave_sum(term));
                    ll_w = 0.0;
                }
            } else if (counts) {
                ll = uniform_f64(ll + wt_prev * (m + log(W / (double)N)));
            }
            if (PFG_TR(P.trace_ll) && tid == 0) P.trace_ll[t] = ll;
        }
        if (is_filter && t > 0) {
            PFG_MARK("cold filter")
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
            {
                const double fixs = invW * 4294967296.0;
#pragma unroll
                for (int k = 0; k < PPT; ++k)
                    cdfu[cdf_phys(tid * PPT + k)] = cvt_u32_sat(cs[k] * fixs);
            }
        } else {
#pragma unroll
            for (int k = 0; k < PPT; ++k)
                if (valid[k]) {
                    cdf[FAST ? cdf_phys(k * NT + tid) : k * NT + tid] = cs[k] * invW;
                    if (PARIS || N2) lwL[k * NT + tid] = lw[k];
                }
        }
        PFG_PH(4)
        block_sync<NW>();                                                       // barrier 3
        PFG_PH(5)
        if (PRIO == 1) __builtin_amdgcn_s_setprio(2);                           // the LDS-heavy phases E, F: see PRIO
        if (PRIO == 2) __builtin_amdgcn_s_setprio(0);

        // ---- (E) ancestors: smallest j with cdf[j] > u (searchsorted 'right').  Branch-free:
        // probes past the end read cdf[N-1] (= 1 > u), the final clamp covers rounding.
        if (RNG != PFG_RNG_REPLAY) {
            if (systematic) {
                // extension: one uniform per timestep (drawn by thread 0 before barrier 2)
                const double u0 = red_W0[0];
#pragma unroll
                for (int k = 0; k < PPT; ++k) uu[k] = ((double)(k * NT + tid) + u0) * invN;
            } else if (!BLK) {
#pragma unroll
                for (int k = 0; k < PPT; ++k) uu[k] = u01_32(rng.next());
            }
        }
        int anc[PPT];
#pragma unroll
        for (int k = 0; k < PPT; ++k) anc[k] = 0;
        if (BLK) {
            uint32_t ua[PPT];
#pragma unroll
            for (int k = 0; k < PPT; ++k) ua[k] = SORTED ? us[k] : rng.next();
#ifdef PFG_EXP_EVENWORDS
            // knock-out (timing only, NOT a valid resampler): evenly spaced words in CDF order -- what sorted uniforms
            // would do to the LDS bank conflicts of the search and the gathers, without their cost
#pragma unroll
            for (int k = 0; k < PPT; ++k) ua[k] = (uint32_t)(((uint64_t)(tid * PPT + k) << 32) / (uint64_t)(NT * PPT)) + (ua[k] >> 12);
#endif
            if (PFG_TR(P.trace_x) && P.rec_u) {           // test instrumentation: the words this launch searched with
#pragma unroll
                for (int k = 0; k < PPT; ++k)
                    if (valid[k]) P.rec_u[(size_t)t * N + k * NT + tid] = ua[k];
            }
            // the search runs on BYTE offsets into the CDF (the LDS address itself: probe offsets fold into the
            // ds_read immediates and a level costs compare + select + add, no address arithmetic)
            using lds_u32 = const __attribute__((address_space(3))) uint32_t;
            const uint32_t cdf_base = (uint32_t)(uintptr_t)(lds_u32 *)cdfu;
            uint32_t off[PPT];
#pragma unroll
            for (int k = 0; k < PPT; ++k) off[k] = cdf_base;
            // PIVOTS (A/B, -DPFG_OPT_PIVOTS=1; 1024 slots): the three entries the first two levels of every search
            // compare with (positions 511, 255, 767) are read ONCE per wave and timestep (broadcast reads) and held in
            // scalar registers: two dependent LDS round trips and eight ds_read per lane-timestep less
            constexpr bool PIVOTS = PFG_OPT_PIVOTS && NT * PPT == 1024;
            constexpr int FIRST_STEP = PIVOTS ? 128 : (NT * PPT) >> 1;
            if constexpr (PIVOTS) {
                const uint32_t p511 = __builtin_amdgcn_readfirstlane(cdfu[cdf_phys(511)]);
                const uint32_t p255 = __builtin_amdgcn_readfirstlane(cdfu[cdf_phys(255)]);
                const uint32_t p767 = __builtin_amdgcn_readfirstlane(cdfu[cdf_phys(767)]);
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    const bool g1 = p511 <= ua[k];
                    const uint32_t t2 = g1 ? p767 : p255;
                    const bool g2 = t2 <= ua[k];
                    off[k] += (g1 ? 4u * (512 + 16) : 0u) + (g2 ? 4u * (256 + 8) : 0u);
                }
            }
#pragma unroll
            for (int step = FIRST_STEP; step >= 1; step >>= 1) {
                const int probe = step - 1 + (step >= 32 ? (step >> 5) - 1 : 0);
                const int adv = step + (step >> 5);
                uint32_t cv[PPT];
#pragma unroll
                for (int k = 0; k < PPT; ++k) cv[k] = *(lds_u32 *)(uintptr_t)(off[k] + 4u * probe);
                if constexpr (PFG_OPT_SEL32 && NT >= 512) {
                // the advanced offset is formed while the probe is in flight; compare + select in the VOP2 forms
                // (v_cmp_le_u32 vcc / v_cndmask_b32 with the implicit vcc).  The compiler's own lowering selects
                // between 0 and the advance with the VOP3 form, which reads vcc as an SGPR operand: two wait states
                // behind the compare, 22 s_nop per lane-timestep in this loop.  Measured (ms per bench launch, with /
                // without): 512 x 2 GARCH 2.86 / 2.95, 1024 x 4 12.68 / 12.76, one wave 1.73 / 1.73, but 256 x 4 SVM
                // 52.3 / 51.5 -- there the glued pairs keep the scheduler from weaving the generator's integer work
                // into the search, which hides more than the wait states cost; so only for NT >= 512.
                uint32_t cand[PPT];
#pragma unroll
                for (int k = 0; k < PPT; ++k) cand[k] = off[k] + 4u * adv;
#pragma unroll
                for (int k = 0; k < PPT; ++k)
                    asm("v_cmp_le_u32_e32 vcc, %1, %2\n\tv_cndmask_b32_e32 %0, %3, %4, vcc"
                        : "=v"(off[k]) : "v"(cv[k]), "v"(ua[k]), "v"(off[k]), "v"(cand[k]) : "vcc");
                } else {
#pragma unroll
                for (int k = 0; k < PPT; ++k) off[k] += (cv[k] <= ua[k]) ? 4u * adv : 0u;
                }
            }
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                uint32_t p = (off[k] - cdf_base) >> 2;
                p -= __umul24(p, 993u) >> 15;                                   // physical -> CDF position (p < 2^15)
                anc[k] = (int)(((p << (NT == 256 ? 8 : (NT == 512 ? 9 : (NT == 1024 ? 10 : 6)))) & (uint32_t)((PPT - 1) * NT)) | (p >> LOG_PPT));   // -> particle index
            }
        } else if (FAST) {
            // sentinel-padded cdf, physical positions: log2(NT*PPT) fixed probes whose offsets
            // fold into the ds_read immediates; logical index recovered once at the end
            // on byte offsets, like the 32-bit search above (the LDS address is the search variable)
            using lds_f64 = const __attribute__((address_space(3))) double;
            const uint32_t cdf_base = (uint32_t)(uintptr_t)(lds_f64 *)cdf;
            uint32_t off[PPT];
#pragma unroll
            for (int k = 0; k < PPT; ++k) off[k] = cdf_base;
#pragma unroll
            for (int step = (NT * PPT) >> 1; step >= 1; step >>= 1) {
                const int probe = step - 1 + (step >= 32 ? (step >> 5) - 1 : 0);
                const int adv = step + (step >> 5);
                double cv[PPT];
#pragma unroll
                for (int k = 0; k < PPT; ++k) cv[k] = *(lds_f64 *)(uintptr_t)(off[k] + 8u * probe);
#pragma unroll
                for (int k = 0; k < PPT; ++k) off[k] += (cv[k] <= uu[k]) ? 8u * adv : 0u;
            }
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const uint32_t p = (off[k] - cdf_base) >> 3;
                anc[k] = (int)(p - (__umul24(p, 993u) >> 15));                  // p - p/33 (exact for p < 32768)
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
#ifdef PFG_EXP_OWNGATHER
        // knock-out (timing / counters only, NOT a valid resampler): the real search runs (its result is kept alive), but
        // every child gathers its OWN slot -- conflict-free lane <-> particle reads: what is left of SQ_LDS_BANK_CONFLICT is
        // the search's share (profiles/r04_lds_conflict_split.txt)
#pragma unroll
        for (int k = 0; k < PPT; ++k) { asm volatile("" :: "v"(anc[k])); anc[k] = valid[k] ? k * NT + tid : 0; }
#endif
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
        PFG_PH(6)
        // ---- (F) gather parents, (G) propose / weight / statistic, (H) publish children ---
        auto slots = [&](auto stat_tag) {
            constexpr int STAT = decltype(stat_tag)::value;
            if (STAT != PFG_STAT_SCORE) { PFG_MARK("cold sufficient-statistics") }
            REAL xp[PPT][NS], sp[PPT][H];
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                if constexpr (PAIRED) {
                    typedef double dv2 __attribute__((ext_vector_type(2)));
                    REAL rec[NS + H];
#pragma unroll
                    for (int p = 0; p < (NS + H) / 2; ++p) {
                        const dv2 v = *reinterpret_cast<const dv2 *>(&cur[sidx(2 * p, anc[k])]);
                        rec[2 * p] = (REAL)v.x; rec[2 * p + 1] = (REAL)v.y;
                    }
#pragma unroll
                    for (int d = 0; d < NS; ++d) xp[k][d] = rec[d];
#pragma unroll
                    for (int h = 0; h < H; ++h) sp[k][h] = rec[NS + h];
                } else {
#pragma unroll
                for (int d = 0; d < NS; ++d) xp[k][d] = cur[sidx(d, anc[k])];
#pragma unroll
                for (int h = 0; h < H; ++h) sp[k][h] = cur[sidx((NS + h), anc[k])];
                }
            }
            PFG_PH(7)
            if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
            if (PRIO == 2) __builtin_amdgcn_s_setprio(2);
            if (!PP) block_sync<NW>();                                          // barrier 4 (single buffer)
            PFG_PH(8)
            if (RNG != PFG_RNG_REPLAY) {
                draw_normals(zz);
                if (PFG_TR(P.trace_x) && P.rec_z) {
#pragma unroll
                    for (int k = 0; k < PPT; ++k)
                        if (valid[k]) P.rec_z[(size_t)t * N + k * NT + tid] = (double)zz[k];
                }
            }
            // One straight-line block per case: the case is uniform, so it is decided once per timestep and
            // not per particle, and the stores of a FAST layout are unconditional (a slot beyond N has its own
            // LDS cell and weight 0) -- the PPT particle chains stay in one basic block for the scheduler.
            auto children = [&](auto upd_tag) {
                constexpr int UPD = decltype(upd_tag)::value;      // 0 plain + statistic, 1 plain, no statistic, 2 general
                if (UPD == 1) { PFG_MARK("cold children-outside-window") }
                if (UPD == 2) { PFG_MARK("cold children-general") }
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    REAL xn[NS], add[H], lwn;
                    particle_step<MODEL, KERNEL, STAT, REAL>(c, mth, xp[k], (REAL)y_t, zz[k], xn, lwn, add);
                    lw[k] = valid[k] ? lwn : (REAL)(-INFINITY);
                    if (UPD == 0) {
                        // Poyiadjis O(N), lambda = 1: 1*s[a] + 0*S + w_t h = s[a] + w_t h exactly
#pragma unroll
                        for (int h = 0; h < H; ++h) sp[k][h] = sp[k][h] + add[h] * (REAL)wt;
                    } else if (UPD == 2) {
#pragma unroll
                        for (int h = 0; h < H; ++h) {
                            const REAL a = use_stat ? add[h] * (REAL)wt : (REAL)0;
                            // pf.py:175-179 / :78-80
                            const REAL sm = (lam * sp[k][h] + oml * (REAL)S[h]) + a;
                            sp[k][h] = is_filter ? a : sm;
                        }
                    }
                    // (1024 threads: the guarded store keeps the particles in separate blocks -- 19 instead of 27
                    // spilled registers at the 128-VGPR cap, 13.1 instead of 14.5 ms per config-4 launch)
                    if ((FAST && NT < 1024) || valid[k]) {
                        const int i = k * NT + tid;
                        if constexpr (PAIRED) {
                            typedef double dv2 __attribute__((ext_vector_type(2)));
                            REAL rec[NS + H];
#pragma unroll
                            for (int d = 0; d < NS; ++d) rec[d] = xn[d];
#pragma unroll
                            for (int h = 0; h < H; ++h) rec[NS + h] = sp[k][h];
#pragma unroll
                            for (int p = 0; p < (NS + H) / 2; ++p) {
                                dv2 v;
                                v.x = (double)rec[2 * p]; v.y = (double)rec[2 * p + 1];
                                *reinterpret_cast<dv2 *>(&nxt[sidx(2 * p, i)]) = v;
                            }
                        } else {
#pragma unroll
                        for (int d = 0; d < NS; ++d) nxt[sidx(d, i)] = xn[d];
#pragma unroll
                        for (int h = 0; h < H; ++h) nxt[sidx((NS + h), i)] = sp[k][h];
                        }
                    }
                }
            };
            if (!plain) children(std::integral_constant<int, 2>{});
            else if (use_stat) children(std::integral_constant<int, 0>{});
            else children(std::integral_constant<int, 1>{});
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
                for (int d = 0; d < NS; ++d) xp[d] = cur[sidx(d, anc[k])];
                particle_step<MODEL, KERNEL, STAT, REAL>(c, mth, xp, (REAL)y_t, zz[k], xn[k], lwn[k], add);
                aux[k] = (MODEL == PFG_MODEL_SVM) ? mth.exp(-xn[k][0]) : (REAL)0;
#pragma unroll
                for (int h = 0; h < H; ++h) sacc[k][h] = (REAL)0;
                if (valid[k]) {
#pragma unroll
                    for (int d = 0; d < NS; ++d) nxt[sidx(d, k * NT + tid)] = xn[k][d];
                }
            }
            // wave-local work queues of pending children (this wave's NT*PPT/NW slots of two [NL]
            // arrays) and the accepted parent of every child of the current backward draw
            int *const wq0 = paris_queue + NL + wave * (PPT * WAVE);
            int *const wq1 = paris_queue + 2 * NL + wave * (PPT * WAVE);
            int *const Jres = paris_queue + 3 * NL;
            const unsigned long long ltmask = (1ull << lane) - 1ull;
            const bool ordered = RNG == PFG_RNG_REPLAY && P.paris_stream != nullptr;
            for (int j = 0; j < Nt; ++j) {
                // ---- 2. accept-reject against the filter weights, up to R rounds per child ------
                // Pending children sit compacted in a wave-local queue.  While more than half a
                // wave is pending each lane tries one candidate for one child per pass; below that
                // a child gets K = 2^k <= 64/pending CONSECUTIVE rounds in one pass (K lanes, the
                // first accepting round wins -- exactly the sequential outcome, also on replayed
                // pools), so the long tail of rounds costs a handful of passes.
                int *qa = wq0, *qb = wq1;
                int cnt = 0;
                if (!ordered) {
#pragma unroll
                for (int k = 0; k < PPT; ++k) {
                    const unsigned long long mk = __ballot(valid[k]);
                    if (valid[k]) qa[cnt + __popcll(mk & ltmask)] = k * NT + tid;
                    cnt += __popcll(mk);
                }
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
                    for (int d = 0; d < NS; ++d) { xI[d] = cur[sidx(d, I)]; xc[d] = nxt[sidx(d, child)]; }
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
                if (ordered) {
                    // ---- 2'. the REFERENCE's consumption order (pf.py:260-341): one sequential stream of uniforms.  Round
                    // r of draw j takes len(L) doubles for np.random.choice and len(L) for np.random.rand, the k-th
                    // pending child IN INDEX ORDER gets the k-th of each; once at most manual_sample_threshold children
                    // are left (or after max_accept_reject rounds) each of them takes one double, in index order, for its
                    // exact categorical draw.  The rank of a pending child = a workgroup-wide exclusive count of the
                    // pending flags in particle-index order (slot-major: particle k * NT + tid).  accept_reject = False
                    // (pf.py:226-236): no rounds, child i's draw j takes double i * Ntilde + j of the timestep's N * Ntilde (the cursor
                    // moves on by N * Ntilde behind the last draw).
                    const gptr<const double> strm = global_ptr(P.paris_stream);       // global_load: see global_ptr
                    const long long cap = P.paris_stream_len;
                    const bool noar = (P.flags & PFG_FLAG_PARIS_NO_ACCEPT_REJECT) != 0;
                    const int mthr = P.paris_manual_threshold;
                    int *const wcnt = paris_queue + NL;                 // [PPT][NW] pending children per (slot, wave)
                    bool pend[PPT];
#pragma unroll
                    for (int k = 0; k < PPT; ++k) pend[k] = valid[k];
                    int S = 0, rank[PPT];
                    for (int round = 0;; ++round) {
#pragma unroll
                        for (int k = 0; k < PPT; ++k) {
                            const unsigned long long mk = __ballot(pend[k]);
                            rank[k] = __popcll(mk & ltmask);
                            if (lane == 0) wcnt[k * NW + wave] = __popcll(mk);
                        }
                        __syncthreads();
                        S = 0;
#pragma unroll
                        for (int k = 0; k < PPT; ++k) {
#pragma unroll
                            for (int w = 0; w < NW; ++w) {
                                if (w == wave) rank[k] += S;
                                S += wcnt[k * NW + w];
                            }
                        }
                        __syncthreads();                                // wcnt is rewritten by the next round
                        if (S == 0 || noar || S <= mthr || round >= R || paris_overflow) break;
                        if (paris_cursor + 2ll * S > cap) { paris_overflow = true; break; }
                        // A round is a dependent chain: HBM latency of the two uniforms, ten LDS probes, gather, exp.  All four
                        // slots of a thread walk it TOGETHER (loads first, the searches level by level, branch-free; a slot
                        // that is not pending searches with u = -1 and ends at parent 0): one after the other, behind an
                        // `if (!pend[k]) continue`, a round cost 8 us.
                        double u1v[PPT], u2v[PPT];
#pragma unroll
                        for (int k = 0; k < PPT; ++k) {
                            u1v[k] = pend[k] ? strm[paris_cursor + rank[k]] : -1.0;
                            u2v[k] = pend[k] ? strm[paris_cursor + S + rank[k]] : 2.0;
                        }
                        int Iv[PPT];
#pragma unroll
                        for (int k = 0; k < PPT; ++k) Iv[k] = 0;
#pragma unroll
                        for (int step = (NT * PPT) >> 1; step >= 1; step >>= 1) {
                            const int probe = step - 1 + (step >= 32 ? (step >> 5) - 1 : 0);
#pragma unroll
                            for (int k = 0; k < PPT; ++k) Iv[k] += (cdf[Iv[k] + probe] <= u1v[k]) ? step + (step >> 5) : 0;
                        }
#pragma unroll
                        for (int k = 0; k < PPT; ++k) {
                            int I = Iv[k];
                            I -= (I * 993) >> 15;
                            I = I < last ? I : last;
                            REAL xI[NS];
#pragma unroll
                            for (int d = 0; d < NS; ++d) xI[d] = cur[sidx(d, I)];
                            const double thr = (double)mth.exp(backward_log_ratio<MODEL, REAL>(c, mth, xI, xn[k]));
                            if (pend[k] && u2v[k] <= thr) { Jres[k * NT + tid] = I; pend[k] = false; }
                        }
                        paris_cursor += 2ll * S;
                    }
                    // the children still pending, in index order: queue position = rank
                    if (S > 0 && !noar && paris_cursor + S > cap) paris_overflow = true;
                    if (S > 0 && noar && paris_cursor + (long long)N * Nt > cap) paris_overflow = true;
                    if (!paris_overflow) {
#pragma unroll
                        for (int k = 0; k < PPT; ++k) {
                            if (!pend[k]) continue;
                            const int i = k * NT + tid;
                            queue[rank[k]] = i;
                            const double um = noar ? strm[paris_cursor + (long long)i * Nt + j] : strm[paris_cursor + rank[k]];
                            nxt[sidx(NS, i)] = (REAL)um;
                        }
                        if (tid == 0) *qcount = S;
                        if (!noar) paris_cursor += S;
                    } else {
#pragma unroll
                        for (int k = 0; k < PPT; ++k)
                            if (pend[k]) Jres[k * NT + tid] = 0;            // the host discards an overflowed window
                    }
                }
                for (int e0 = 0; e0 < cnt; e0 += WAVE) {
                    const int e = e0 + lane;
                    if (e < cnt) {
                        const int i = qa[e];
                        queue[atomicAdd(qcount, 1)] = i;
                        // the child's fallback uniform rides in its (still unused) statistic slot
                        const double um = (RNG == PFG_RNG_REPLAY) ? pman[((size_t)t * Nt + j) * N + i]
                                                                  : u01_32(rng.next());
                        nxt[sidx(NS, i)] = (REAL)um;
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
                    for (int d = 0; d < NS; ++d) xc[d] = nxt[sidx(d, ci)];
                    const double um = (double)nxt[sidx(NS, ci)];
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
                            for (int d = 0; d < NS; ++d) xq[d] = cur[sidx(d, qq)];
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
                        for (int d = 0; d < NS; ++d) xq[d] = cur[sidx(d, qq)];
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
                    if (PFG_TR(P.trace_x) && P.trace_paris_J && valid[k])
                        P.trace_paris_J[((size_t)t * Nt + j) * N + k * NT + tid] = Jk;
                    REAL xJ[NS], aj[H];
#pragma unroll
                    for (int d = 0; d < NS; ++d) xJ[d] = cur[sidx(d, Jk)];
                    additive_stat<MODEL, STAT, REAL>(c, xJ, xn[k], (REAL)y_t, aux[k], aj);
#pragma unroll
                    for (int h = 0; h < H; ++h) {
                        const REAL a = use_stat ? aj[h] * (REAL)wt : (REAL)0;
                        sacc[k][h] += cur[sidx((NS + h), Jk)] + a;
                    }
                }
                __syncthreads();                // queue / statistic-slot scratch free for the next j
            }
            if (ordered && (P.flags & PFG_FLAG_PARIS_NO_ACCEPT_REJECT) && !paris_overflow) paris_cursor += (long long)N * Nt;
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                lw[k] = valid[k] ? lwn[k] : (REAL)(-INFINITY);
                if (valid[k]) {
#pragma unroll
                    for (int h = 0; h < H; ++h) nxt[sidx((NS + h), k * NT + tid)] = sacc[k][h] / (REAL)Nt;
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
                for (int d = 0; d < NS; ++d) xp[d] = cur[sidx(d, anc[k])];
                particle_step<MODEL, KERNEL, STAT, REAL>(c, mth, xp, (REAL)y_t, zz[k], xn[k], lwn[k], add);
                aux[k] = (MODEL == PFG_MODEL_SVM) ? mth.exp(-xn[k][0]) : (REAL)0;
                if (valid[k]) {
#pragma unroll
                    for (int d = 0; d < NS; ++d) nxt[sidx(d, k * NT + tid)] = xn[k][d];
                }
            }
            REAL mx[PPT];
#pragma unroll
            for (int k = 0; k < PPT; ++k) mx[k] = (REAL)(-INFINITY);
#pragma unroll 2
            for (int j = 0; j < N; ++j) {
                REAL xj[NS];
#pragma unroll
                for (int d = 0; d < NS; ++d) xj[d] = cur[sidx(d, j)];
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
                for (int d = 0; d < NS; ++d) xj[d] = cur[sidx(d, j)];
#pragma unroll
                for (int h = 0; h < H; ++h) sj[h] = cur[sidx((NS + h), j)];
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
                    for (int h = 0; h < H; ++h) nxt[sidx((NS + h), k * NT + tid)] = num[k][h] / den[k];
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
        if (PFG_TR(P.trace_x)) {
            // own children back from LDS (written by this thread: no barrier needed)
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                if (valid[k]) {
                    const int i = k * NT + tid;
                    const size_t row = (size_t)(t + 1) * N + i;
                    if (P.trace_anc) P.trace_anc[(size_t)t * N + i] = anc[k];
#pragma unroll
                    for (int d = 0; d < NS; ++d) P.trace_x[row * NS + d] = (double)nxt[sidx(d, i)];
                    P.trace_logw[row] = (double)lw[k];
                    if (P.trace_stats && !is_filter) {
#pragma unroll
                        for (int h = 0; h < H; ++h)
                            P.trace_stats[row * H + h] = (double)nxt[sidx((NS + h), i)];
                    }
                }
            }
        }
        if (PP) { REAL *tmp = cur; cur = nxt; nxt = tmp; }
        wt_prev = wt;
        PFG_PH(9)
    }
    if (P.stamps && tid == 0) { P.stamps[2] = __builtin_amdgcn_s_memtime(); P.stamps[3] = __builtin_amdgcn_s_memrealtime(); }
#ifdef PFG_PHASE_STAMPS
    if (P.stamps && lane == 0) {
#pragma unroll
        for (int q = 0; q < 10; ++q) atomicAdd(reinterpret_cast<unsigned long long *>(P.stamps) + 4 + q, ph_acc[q]);
    }
#endif
#undef PFG_PH
#undef PFG_MARK

    // ---- outputs --------------------------------------------------------------------
    if (RNG == PFG_RNG_REPLAY && P.out) {
        tie = -wave_max(-tie);
        if (lane == 0) red_max[wave] = tie;
        __syncthreads();
        tie = red_max[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) tie = red_max[w] < tie ? red_max[w] : tie;
    }
    if (PARIS && tid == 0 && P.paris_consumed) {
        P.paris_consumed[0] = paris_overflow ? -1ll : paris_cursor;
        // RAW: a cached Gaussian is pending -> how far back from the end of the consumption its pair of doubles starts
        P.paris_consumed[1] = (RAWCAP && raw && carry_has && !paris_overflow) ? paris_cursor - raw_slots[1] : 0ll;
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
                for (int d = 0; d < NS; ++d) P.final_x[(size_t)i * NS + d] = (double)cur[sidx(d, i)];
                if (P.final_logw) P.final_logw[i] = (double)lw[k];
                if (P.final_stats && !is_filter) {
#pragma unroll
                    for (int h = 0; h < H; ++h)
                        P.final_stats[(size_t)i * H + h] = (double)cur[sidx((NS + h), i)];
                }
            }
        }
    }
}

}  // namespace pfg
