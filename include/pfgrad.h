/*
 * pfgrad.h -- C ABI of libpfgrad.so: the buffered particle-filter score / log-likelihood
 * estimator of sgmcmc_ssm, implemented as hand-written HIP kernels for gfx950 (MI355X).
 *
 * The reference is pure Python and has no FFI; this ABI is what a ctypes binding on the
 * reference side would bind in place of the call
 *     buffered_pf_wrapper(pf=..., observations=..., parameters=..., N=..., kernel=...,
 *                         additive_statistic_func=..., t1=..., tL=..., weights=...,
 *                         prior_mean=..., prior_var=...)
 * (sgmcmc_ssm/particle_filters/buffered_smoother.py:156-199, loop :12-149) as issued by
 * Helper.pf_gradient_estimate / pf_loglikelihood_estimate
 * (models/svm/helper.py:67-185, models/garch/helper.py:59-170, models/lgssm/helper.py:1016-1143).
 * Python callables (Kernel objects, additive_statistic_func) cannot cross to the device,
 * so (Kernel, statistic) pairs are replaced by enumerated ids.  See INTEGRATION.md for the
 * reference-side ctypes stub.
 *
 * Conventions: plain C, no torch types.  Every function returns 0 on success or a negative
 * pfg_status code and never throws; the message is available from pfg_last_error().
 * A pfg_ctx is not thread-safe (one per host thread / process).  Host pointers are owned by
 * the caller and must stay valid for the duration of the call; device memory allocated by
 * the library is owned by the pfg_ctx.
 */
#ifndef PFGRAD_H
#define PFGRAD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PFG_VERSION 125          /* 0.1.25 */
#define PFG_MAX_STAT 4           /* widest additive statistic (GARCH / LGSSM score) */
#define PFG_MAX_THETA 4          /* raw parameters per model */
#define PFG_OUT_DOUBLES 8        /* doubles in one result record (see pfg_dev_problem.out) */
#define PFG_STAMP_WORDS 16       /* uint64 words behind pfg_dev_problem.stamps */
#define PFG_MAX_PRED 16          /* predictive log-likelihood leads k = 0..num_steps_ahead (<= 15) */

typedef struct pfg_ctx pfg_ctx;

/* models: theta layout = Parameters.var_dict order of the reference
 *   SVM   [A, LQinv, LRinv]                       models/svm/parameters.py:21-25
 *   GARCH [log_mu, logit_phi, logit_lambduh, LRinv] models/garch/parameters.py:19-22
 *   LGSSM [A, C, LQinv, LRinv]                    models/lgssm/parameters.py:20-25
 * score statistic column order (what pf_gradient_estimate unpacks):
 *   SVM   [LRinv, LQinv, A]                       models/svm/helper.py:121-126
 *   GARCH [LRinv, log_mu, logit_phi, logit_lambduh] models/garch/helper.py:109-115
 *   LGSSM [LRinv, LQinv, C, A]                    models/lgssm/helper.py:1136-1142        */
enum pfg_model { PFG_MODEL_SVM = 0, PFG_MODEL_GARCH = 1, PFG_MODEL_LGSSM = 2 };
/* proposal kernels: models/{svm,garch,lgssm}/kernels.py ("prior" bootstrap, "optimal") */
enum pfg_kernel { PFG_KERNEL_PRIOR = 0, PFG_KERNEL_OPTIMAL = 1 };
/* smoothers: particle_filters/pf.py:138-181 (nemeth; poyiadjis_N = lambduh 1.0), :40-82 (filter),
 * :183-341 (PaRIS: Ntilde backward-sampled parents per child by accept-reject, exact
 * categorical fallback after max_accept_reject rounds; N <= 1024 LDS-resident, up to 16384 in
 * the large-N kernel through pfg_run / pfg_run_batch, which size its scratch) */
enum pfg_smoother { PFG_SMOOTHER_NEMETH = 0, PFG_SMOOTHER_FILTER = 1, PFG_SMOOTHER_PARIS = 2,
                    /* EXTENSION (not in the reference, which resamples multinomially every step,
                     * pf.py:26-30): NEMETH with systematic resampling, u_i = (i + u0)/N with ONE
                     * uniform u0 per timestep.  DEVICE rng, N <= 1024; parity-unpinned, checked
                     * statistically.  Its own kernel instantiation (keeps the hot path untouched). */
                    PFG_SMOOTHER_NEMETH_SYSTEMATIC = 3,
                    /* poyiadjis_smoother, the O(N^2) algorithm (pf.py:84-136): every child averages
                     * stats_j + w_t h(x_j, child) over ALL parents j with the backward weights
                     * w_j q(child | x_j), normalised per child.  N <= 1024. */
                    PFG_SMOOTHER_POYIADJIS_N2 = 4,
                    /* LAUNCH-LEVEL id only (pfg_launch_device_smoother; never in a descriptor): the caller states that
                     * every descriptor of the batch is the Poyiadjis O(N) score -- smoother = NEMETH, lambduh = 1.0,
                     * stat = SCORE (what pf = 'poyiadjis_N' means, pf.py:139-180).  Same kernels and the same numbers
                     * as NEMETH; the 1024 x 4 (1024 < N <= 4096) and the one-wave x 2 (N <= 128, batches) fp64
                     * device-generator units, and the 256 x 4 / one-wave x 2 REPLAY units of SVM and LGSSM, run a twin
                     * with the filter, the lambda != 1 shrinkage and the other statistics compiled out (BASELINE config
                     * 4: -3.9 %, config 1: -3.8 %; seed-compatible arithmetic -4.6 % / -6.2 %, bitwise the same numbers).
                     * A descriptor that breaks the statement gets out[0..7] = NaN from that twin.  pfg_run_batch
                     * chooses it by itself when every window of the batch qualifies. */
                    PFG_SMOOTHER_POYIADJIS_N = 5 };
/* additive statistic: *_complete_data_loglike_gradient (score), *_sufficient_statistics, zero */
enum pfg_stat { PFG_STAT_SCORE = 0, PFG_STAT_SUFF = 1, PFG_STAT_NONE = 2,
                /* k-step-ahead predictive log-likelihoods accumulated with the filter's
                 * logsumexp update (pf.py:72-76; statistic functions svm/helper.py:352-395,
                 * lgssm/helper.py:1281-1336, garch/helper.py:374-412).  FILTER smoother only. */
                PFG_STAT_PREDICTIVE = 3 };
/* particle-state arithmetic type.  Weight normalisation, CDF and search are always f64. */
enum pfg_dtype { PFG_F64 = 0, PFG_F32 = 1 };
/* REPLAY: caller supplies the NumPy legacy stream (z0[N], u[T*N], z[T*N]) -> results
 * reproduce the reference on the same seed.  DEVICE: generated in the kernel: one jsf32
 * generator per lane keyed by Philox4x32-10(seed; lane, stream, *step_ctr); 32-bit uniforms,
 * Box-Muller normals evaluated on the f32 transcendental units and widened. */
enum pfg_rng { PFG_RNG_REPLAY = 0, PFG_RNG_DEVICE = 1 };

enum pfg_status {
    PFG_OK = 0,
    PFG_ERR_INVALID = -1,      /* bad argument (message says which) */
    PFG_ERR_UNSUPPORTED = -2,  /* combination not built (e.g. SVM optimal kernel) */
    PFG_ERR_DEVICE = -3,       /* HIP runtime error */
    PFG_ERR_NOMEM = -4,
    PFG_ERR_NUMERIC = -5       /* |A| > 1 for SVM (models/svm/kernels.py:6-11) */
};

/* flags of pfg_problem / pfg_dev_problem */
#define PFG_FLAG_GARCH_STATIONARY_PRIOR 1u /* prior_var = alpha/(1-beta-gamma) (garch/helper.py:324-327) */
#define PFG_FLAG_PARIS_NO_ACCEPT_REJECT 2u /* paris_smoother(accept_reject=False), pf.py:226-236: every child draws its Ntilde
                                            * parents from the exact backward categorical (REPLAY + paris_stream only) */
#define PFG_FLAG_PARIS_RAW_STREAM 4u       /* paris_stream is the window's WHOLE np.random stream (see pfg_problem.paris_stream) */
#define PFG_FLAG_PARIS_RAW_CARRY 8u        /* ... and its first double is the generator's cached Gaussian (has_gauss = 1) */

/* One buffered PF window, host side (all pointers are HOST pointers, C-contiguous f64). */
typedef struct pfg_problem {
    int32_t model, kernel, smoother, stat, dtype, rng;
    int32_t N;               /* particles */
    int32_t T;               /* timesteps of the buffered window */
    int32_t t1, tL;          /* statistic / log-likelihood accumulate on [t1, tL) */
    uint32_t flags;
    int32_t reserved;
    double lambduh;          /* Nemeth shrinkage; 1.0 = Poyiadjis O(N) */
    double prior_mean, prior_var;
    const double *y;         /* [T] observations (m = 1) */
    const double *weights;   /* [tL-t1] importance weights or NULL (= 1) */
    const double *theta;     /* raw parameters, model layout above */
    const double *z0, *u, *z;/* REPLAY streams: [N], [T*N], [T*N]; NULL for DEVICE */
    uint64_t seed, stream;   /* DEVICE rng: key and stream id (global chain id) */
    const double *init_x, *init_logw, *init_stats; /* optional warm start: [N*n],[N],[N*h] */
    /* PaRIS only.  REPLAY pools of uniforms addressed by (timestep, j, round, particle):
     * paris_idx_u / paris_acc_u [T][Ntilde][max_accept_reject][N] (index draw, accept draw),
     * paris_man_u [T][Ntilde][N] (fallback draw); NULL with the DEVICE rng. */
    int32_t Ntilde, max_accept_reject;
    const double *paris_idx_u, *paris_acc_u, *paris_man_u;
    /* PFG_STAT_PREDICTIVE only: leads 0..num_steps_ahead; REPLAY pool of the standard normals the
     * statistic draws, pred_z [T][num_steps_ahead+1][N] (SVM, GARCH; unused for LGSSM). */
    int32_t num_steps_ahead;
    /* != 0: also compute the ELEMENTWISE sufficient statistics of the window -- the reference's
     * `elementwise_statistic=True` run (buffered_smoother.py:64-65, 201-210) behind
     * Helper.pf_latent_var_distr (svm/helper.py:249-294): one 3-column block [x', x'^2, x x'] (GARCH:
     * [x', x'^2, x'^4]) per window timestep, 3 (tL - t1) columns per particle, carried through the
     * smoother's recursion.  The filter runs once (its trajectory does not depend on the statistic) and
     * records what the recursion needs on the device; a second pass streams the [N][3 (tL-t1)]
     * statistic matrix through HBM step by step.  NEMETH (any lambduh) and PARIS. */
    int32_t elementwise;
    const double *pred_z;
    /* PaRIS in the REFERENCE's np.random order (REPLAY): ONE sequential stream of uniforms, consumed as
     * accept_reject_based_backward_sampling does (pf.py:260-341): per draw j and round, len(L) doubles for
     * np.random.choice then len(L) for np.random.rand, the k-th pending child in index order taking the k-th of each;
     * once <= paris_manual_threshold children are left (or after max_accept_reject rounds) one double per child for
     * its exact draw.  Replaces the addressed pools paris_idx_u / acc_u / man_u (which must then be NULL).  The number
     * of doubles consumed is data dependent: pfg_result.paris_consumed returns it (-1: the stream was too short, run
     * again with a longer one -- the consumption of a window is bounded by (T + 1) (4 N + 2 N Ntilde max_accept_reject +
     * N Ntilde) doubles, a caller that still sees -1 beyond that has a different problem).  The filter's own draws of a timestep (u, z) precede them in np.random's order, so a
     * caller that reproduces np.random.seed() runs ONE timestep per call (warm start init_x / init_logw / init_stats),
     * as sgmcmc_ssm_amd.particle_filters does.  A window of several timesteps carries the cursor from one timestep to
     * the next (PARIS_NO_ACCEPT_REJECT: child i's draw j at timestep t reads double (t N + i) Ntilde + j): given the
     * concatenation of what the single-timestep calls consumed it repeats them in one launch (with `elementwise`).
     * PFG_FLAG_PARIS_RAW_STREAM (dtype f64; N <= 16384): z0 / u / z are NULL and paris_stream holds what RandomState.random_sample
     * delivers from the generator's current state: the kernel takes EVERYTHING from it in np.random's order -- N normals
     * for x0, then per timestep N uniforms, N normals and the backward sampling's uniforms -- so a whole window is one
     * launch.  The normals are NumPy's legacy Gaussians (Marsaglia's polar method on pairs of doubles, second variate of
     * a pair cached for the next draw): acceptance is exact fp64 arithmetic, so the consumption is the reference's to
     * the double; the values go through the device's log (<= 1 ulp from the host libm's; REPLAY tolerance).  With
     * PFG_FLAG_PARIS_RAW_CARRY paris_stream[0] is the generator's pending cached Gaussian and the doubles start at [1].
     * pfg_result.paris_consumed counts the doubles taken (incl. that slot); pfg_result.paris_carry_back != 0: a cached
     * Gaussian is pending at the end, the second variate of the pair of doubles that starts paris_carry_back doubles
     * before the end of the consumption (the caller recomputes it with its own libm and hands the generator on). */
    const double *paris_stream;
    int64_t paris_stream_len;
    int32_t paris_manual_threshold, reserved2;
    /* DEVICE rng: SGLD step the window belongs to, mixed into the generator key exactly as a resident chain's
     * device-side counter (*pfg_dev_problem.step_ctr) is: a window run through pfg_run_batch with
     * (seed, stream = global chain id, step) draws what that chain draws at that step through pfg_launch_device. */
    uint64_t step;
} pfg_problem;

/* Result of one window; optional arrays are caller-allocated HOST buffers or NULL. */
typedef struct pfg_result {
    double mean_stat[PFG_MAX_STAT]; /* sum_i stats_i softmax(logw)_i (average_statistic); FILTER: running m */
    double loglik;                  /* loglikelihood_estimate */
    double *x_T;        /* [N*n]  final particles          */
    double *logw_T;     /* [N]    final log weights        */
    double *stats_T;    /* [N*h]  final per-particle statistics (NEMETH only) */
    double *trace_x;    /* [(T+1)*N*n]  = save_all 'all_x_t'            */
    double *trace_logw; /* [(T+1)*N]    = 'all_log_weights'             */
    double *trace_stats;/* [(T+1)*N*h]  = 'all_statistics' (NEMETH)     */
    double *trace_ll;   /* [T+1]        = 'all_loglikelihood_estimate'  */
    int32_t status;
    int32_t paris_carry_back;   /* PFG_FLAG_PARIS_RAW_STREAM: see pfg_problem.paris_stream; else 0 */
    int32_t *trace_anc; /* [T*N] ancestor index of every particle at every step (with trace_x):
                           the genealogy, from which smoothed marginals are traced back */
    double pred[PFG_MAX_PRED]; /* PFG_STAT_PREDICTIVE: out['statistics'][k] of the reference */
    /* DEVICE rng + trace_x only (test instrumentation of the device-generator kernels): the random
     * inputs the kernel drew, so that a CPU oracle can replay the SAME launch deterministically.
     * rec_u [T*N]: the raw 32-bit word child i searched the resampling CDF with at step t;
     * rec_z [T*N]: its standard normal (as widened to f64);  rec_z0 [N]: the x0 normals. */
    uint32_t *rec_u;
    double *rec_z, *rec_z0;
    double *rec_ud;     /* [T*N]: large-N device kernel (pf_big_kernel), whose resampling uniforms are not raw
                           generator words but SORTED uniforms built from exponential spacings: the uniform in
                           (0,1) child i searched the CDF with at step t (rec_u stays 0 there) */
    /* pfg_problem.elementwise: ew_mean [3 (tL-t1)] = average_statistic of the elementwise run (required);
     * ew_stats [N * 3 (tL-t1)] = its per-particle statistics, row-major (optional) */
    double *ew_mean, *ew_stats;
    int64_t paris_consumed;  /* doubles of pfg_problem.paris_stream the window consumed; -1 = stream too short */
} pfg_result;

/* Device-side descriptor: one per workgroup, resident in HBM.  All pointers are DEVICE
 * pointers.  Used by pfg_launch_device() for callers that keep everything resident
 * (multi-chain SGLD engine, bench).  Layout is part of the ABI (Python mirrors it). */
typedef struct pfg_dev_problem {
    const double *y;
    const double *weights;
    const double *theta;
    const double *z0, *u, *z;
    const double *init_x, *init_logw, *init_stats;
    double *out;             /* [PFG_OUT_DOUBLES]: mean_stat[0..3], loglik, sum of weights W_T,
                                max log weight m_T, min |u - cdf| tie margin (REPLAY) */
    double *final_x, *final_logw, *final_stats;
    double *trace_x, *trace_logw, *trace_stats, *trace_ll;
    const uint64_t *step_ctr;/* optional: *step_ctr is mixed into the DEVICE rng key */
    void *scratch;           /* large-N variant: per-problem state buffers, else NULL */
    double prior_mean, prior_var, lambduh;
    uint64_t seed, stream;
    int32_t T, t1, tL, N;
    int32_t smoother, stat;
    uint32_t flags;
    int32_t reserved;
    const double *paris_idx_u, *paris_acc_u, *paris_man_u;   /* PaRIS REPLAY pools (see pfg_problem) */
    int32_t Ntilde, max_accept_reject;
    int32_t *trace_anc;      /* [T*N] or NULL */
    const double *pred_z;    /* PREDICTIVE, REPLAY: [T][num_steps_ahead+1][N] */
    double *pred_out;        /* PREDICTIVE: [PFG_MAX_PRED] */
    void *pred_scratch;      /* PREDICTIVE: [N][PFG_MAX_PRED] of the state type */
    int32_t num_steps_ahead, reserved3;
    uint32_t *rec_u;         /* [T*N] or NULL: see pfg_result.rec_u (DEVICE rng, with trace_x) */
    double *rec_z, *rec_z0;  /* [T*N], [N] or NULL */
    double *rec_ud;          /* [T*N] or NULL: see pfg_result.rec_ud */
    int32_t *trace_paris_J;  /* [T][Ntilde][N] or NULL (PARIS, with trace_x): the backward-sampled parent of
                                every child and draw: the structure the elementwise statistics are carried through */
    const double *paris_stream;      /* see pfg_problem.paris_stream */
    int64_t paris_stream_len;
    int64_t *paris_consumed;         /* [2] or NULL: doubles consumed (-1: stream too short), carry-back distance */
    int32_t paris_manual_threshold, reserved4;
    uint64_t *stamps;        /* [PFG_STAMP_WORDS] or NULL (measurement): wave 0 of the workgroup writes
                                s_memtime / s_memrealtime (100 MHz) at kernel start [0],[1] and end [2],[3]
                                -> in-kernel shader clock = ([2]-[0]) / ([3]-[1]) * 100 MHz; [4..15]: per-phase
                                cycle sums, only in diagnostic builds (-DPFG_PHASE_STAMPS), else untouched */
} pfg_dev_problem;

int pfg_version(void);
/* sizeof() of an ABI struct: 0 pfg_problem, 1 pfg_result, 2 pfg_dev_problem, 3 pfg_prior_hyper
 * (bindings assert their mirror has the same size) */
int pfg_struct_size(int which);
int pfg_create(pfg_ctx **out, int device_id);
void pfg_destroy(pfg_ctx *ctx);
const char *pfg_last_error(pfg_ctx *ctx);   /* ctx may be NULL: last error of pfg_create */

/* Host-buffer entry points: stage inputs to HBM, run, copy results back, synchronise.
 * pfg_run_batch requires all problems to share model/kernel/dtype/rng; N, T, windows,
 * parameters and seeds may differ per problem.  One workgroup per problem. */
int pfg_run(pfg_ctx *ctx, const pfg_problem *p, pfg_result *r);
int pfg_run_batch(pfg_ctx *ctx, int B, const pfg_problem *ps, pfg_result *rs);

/* The context's own (non-blocking) stream, as a hipStream_t. */
void *pfg_ctx_stream(pfg_ctx *ctx);

/* Resident entry point: `dev_probs` is a DEVICE array of B descriptors; launches on
 * `hip_stream` (a hipStream_t used as is: NULL is HIP's default stream; pass
 * pfg_ctx_stream(ctx) for the context's own) and returns without synchronising.
 * n_max = the largest N in the batch (selects the kernel variant). */
int pfg_launch_device(pfg_ctx *ctx, int model, int kernel, int dtype, int rng, int n_max,
                      int B, const pfg_dev_problem *dev_probs, void *hip_stream);
/* pfg_launch_device / pfg_launch_device_smoother are the PRODUCTION launches: for the plain (NEMETH / FILTER)
 * LDS-resident kernels (DEVICE generator, and REPLAY) they run an instantiation with the trace instrumentation compiled
 * out, which IGNORES the trace_x / trace_logw / trace_stats / trace_ll / trace_anc / rec_* fields of the
 * descriptors (out / final_* / stamps are honoured).  pfg_launch_device_traced runs the twin that honours them
 * (save_all trajectories, recorded generator draws); same arguments as pfg_launch_device_smoother.  Both twins
 * return bitwise the same `out` record for the same descriptor (tests/test_gpu_device_replay.py).
 * pfg_last_traced: 1 if the latest launch through the context honoured trace buffers, 0 if not. */
int pfg_launch_device_traced(pfg_ctx *ctx, int model, int kernel, int dtype, int rng, int smoother,
                             int n_max, int B, const pfg_dev_problem *dev_probs, void *hip_stream);
int pfg_last_traced(pfg_ctx *ctx);
/* as pfg_launch_device for a batch whose descriptors all have smoother = `smoother`
 * (PFG_SMOOTHER_PARIS, _NEMETH_SYSTEMATIC and _POYIADJIS_N2 have their own kernel instantiations;
 * the plain entry point serves NEMETH / FILTER) */
int pfg_launch_device_smoother(pfg_ctx *ctx, int model, int kernel, int dtype, int rng, int smoother,
                               int n_max, int B, const pfg_dev_problem *dev_probs, void *hip_stream);
/* N above the one-workgroup kernels' maximum (16384 < N <= 4194304; the reference has no limit and its bias experiments
 * call pf_gradient_estimate with N = 1000000: gradient_error_fig_scripts/svm_grad_compare.py:68-82): every window of the
 * batch runs as a WHOLE-GPU window -- the particle axis cut into tiles of 1024 (N <= 524288) / 2048 particles, one 256-thread workgroup each, one
 * kernel launch per timestep (T_max + 2 launches on `hip_stream`; REPLAY: 5 T_max + 2, the extra four per timestep build
 * the reference's CDF -- NumPy's sequential cumsum, bit for bit -- with the particle axis spread over the GPU).  T_max = the largest T of
 * the batch (the host issues the launches, so it has to know; shorter windows leave theirs at once).  NEMETH / FILTER
 * with the score, sufficient or no statistic.  Every descriptor needs `scratch` of pfg_scratch_bytes(model, dtype, rng, N)
 * bytes (256-byte aligned); windows of one batch must all have N <= 524288 or all N > 524288 (the tile size differs).
 * DEVICE rng: the resampling uniforms are SORTED uniforms (exponential spacings, rec_ud), generator lane = particle
 * index mod tile; out[7] = 1.  REPLAY: u / z as for every kernel, out[7] = the smallest |u - cdf| margin of the run.
 * pfg_run / pfg_run_batch route N > 16384 here by themselves (PFGRAD_VARIANT=grid forces it for smaller N). */
int pfg_launch_device_grid(pfg_ctx *ctx, int model, int kernel, int dtype, int rng, int n_max, int T_max, int B,
                           const pfg_dev_problem *dev_probs, void *hip_stream);
/* The same window, one piece per call, for callers that put events, graph nodes or their own work between the launches:
 * phase = PFG_GRID_PHASE_INIT (x0 / warm start, the partials of timestep 0), then every timestep t = 0 .. T_max - 1 in order
 * (phase = t; REPLAY: the CDF kernel + the step kernel), then PFG_GRID_PHASE_FINISH (out / final_* / trace_ll[T]).  The
 * sequence INIT, 0, 1, ..., T_max - 1, FINISH on one stream is exactly pfg_launch_device_grid. */
#define PFG_GRID_PHASE_INIT (-2)
#define PFG_GRID_PHASE_FINISH (-3)
int pfg_launch_device_grid_phase(pfg_ctx *ctx, int model, int kernel, int dtype, int rng, int n_max, int phase, int B,
                                 const pfg_dev_problem *dev_probs, void *hip_stream);
/* Both of the above with the batch's smoother stated (as pfg_launch_device_smoother does for N <= 16384): NEMETH / FILTER
 * run what the entry points above run; PFG_SMOOTHER_POYIADJIS_N -- every window is (NEMETH, lambduh = 1, score) -- runs a
 * twin of the device-generator timestep kernel specialised to that estimator (fewer registers: g1 +2.2 %, four windows of
 * 4 10^5 particles +5.7 %; same numbers to the last place or two).  A window that breaks the statement gets out[0..7] = NaN.
 * phase = PFG_GRID_PHASE_ALL: the whole window (T_max + 2 launches); otherwise one piece, T_max is ignored.
 * pfg_run_batch states POYIADJIS_N by itself when every window of a whole-GPU batch qualifies. */
#define PFG_GRID_PHASE_ALL (-1)
int pfg_launch_device_grid_smoother(pfg_ctx *ctx, int model, int kernel, int dtype, int rng, int smoother, int n_max,
                                    int T_max, int phase, int B, const pfg_dev_problem *dev_probs, void *hip_stream);

/* Buffered-subsequence sampling for resident chains, on the device: for every descriptor b draw
 * a window start as SGMCMCSampler._random_subsequence_and_buffers does (sgmcmc_sampler.py:259-288;
 * 'uniform' style: start ~ U{0..T-S}; strict != 0: start = S * U{0..T/S-1}), keyed by
 * Philox(seed; chain_offset + b, *step_ctr), and point y / T / t1 / tL / weights at
 * [start - buffer, start + S + buffer) clipped to the series.  weights_table: [T-S+1][S] row per
 * start (random_subsequence_and_weights, :1969-2017) or NULL.  With this the whole SGLD step is
 * three launches and no host work: capturable in a hipGraph.  Asynchronous on `hip_stream`. */
int pfg_sample_windows_device(pfg_ctx *ctx, int B, pfg_dev_problem *dev_probs, const double *y_dev,
                              const double *weights_table_dev, int T, int S, int buffer, int strict,
                              uint64_t seed, uint64_t chain_offset, const uint64_t *step_ctr,
                              void *hip_stream);

/* bytes of per-problem HBM scratch (pfg_dev_problem.scratch, 256-byte aligned) the large-N
 * kernel (N <= 16384) or the whole-GPU window (N <= 4194304) needs for (model, dtype, rng, N); 0 when an LDS-resident
 * variant serves this size, -1 when N is above the supported maximum */
int64_t pfg_scratch_bytes(int model, int dtype, int rng, int N);
/* name of the kernel variant pfg_launch_device would pick (for profiles / logs) */
const char *pfg_variant_name(int model, int kernel, int dtype, int rng, int n_max);
/* tag of the kernel variant the latest launch through this context ran ("wg256x4s", "wg1024x1",
 * "wg1024x4s", "wg64x2", "big4096", "big16384", "mem1024", ...): the batch size takes part in the
 * choice, so tests and profiles read it back instead of predicting it */
const char *pfg_last_variant(pfg_ctx *ctx);
int pfg_synchronize(pfg_ctx *ctx);

/* SGLD parameter update for B resident chains (sgmcmc_sampler.py:427-464, 529-566, 650-656):
 *   theta_v += eps*(grad_logprior_v(theta) + ghat_v)/Tscale + sqrt(2 eps) N(0, 1/Tscale)
 * followed by project_parameters.  `outs` is the [B][PFG_OUT_DOUBLES] result array the
 * PF kernel wrote (score column order), `theta` [B][PFG_MAX_THETA] is updated in place.
 * hyper: model-specific prior hyper-parameters, see pfg_prior_hyper.  The noise of chain b is
 * keyed by (seed, chain_offset + b, *step_ctr): a chain's trajectory depends on its GLOBAL
 * index only, not on how chains are spread over GPUs.  *step_ctr is incremented afterwards. */
typedef struct pfg_prior_hyper {
    double df_Qinv, scale_Qinv, df_Rinv, scale_Rinv;  /* Wishart on Qinv/Rinv (covariance.py:252-284) */
    double mean_A, var_col_A, mean_C, var_col_C;      /* matrix normal (matrices.py:597-607) */
    double scale_mu, shape_mu, alpha_phi, beta_phi, alpha_lambduh, beta_lambduh; /* garch_var.py:152-165 */
} pfg_prior_hyper;
int pfg_sgld_update_device(pfg_ctx *ctx, int model, int B, double *theta, const double *outs,
                           const pfg_prior_hyper *hyper, double epsilon, double Tscale,
                           uint64_t seed, uint64_t chain_offset, uint64_t *step_ctr,
                           void *hip_stream);

/* EXTENSION (not in the reference: its samplers are SGD / ADAGRAD / SGLD / SGRLD / Gibbs,
 * sgmcmc_sampler.py:467-648): SGHMC update with friction alpha in (0, 1] for resident chains,
 *   v <- (1 - alpha) v + eps (grad_logprior + ghat)/Tscale + N(0, 2 alpha eps / Tscale),  theta <- theta + v,
 * then project_parameters.  momentum [B][PFG_MAX_THETA] is updated in place; alpha = 1 is exactly
 * pfg_sgld_update_device (same noise stream).  Parity-unpinned. */
int pfg_sghmc_update_device(pfg_ctx *ctx, int model, int B, double *theta, double *momentum,
                            const double *outs, const pfg_prior_hyper *hyper, double epsilon, double alpha,
                            double Tscale, uint64_t seed, uint64_t chain_offset, uint64_t *step_ctr,
                            void *hip_stream);

/* Inverse-multiquadric kernel Stein discrepancy of K points x[K][d] with score estimates
 * g[K][d] (HOST pointers, d <= 8): sqrt(sum_{i,j} k0(x_i,x_j)) / K for
 * k(x,y) = (c^2 + |x-y|^2)^(-beta) -- IMQ_KSD of sgmcmc_ssm/trace_metric_functions.py:20-81
 * (the O(K^2) pass of the KSD evaluation, nonlinear_ssm_pf_experiment_scripts/svm/driver.py:906-1090). */
int pfg_imq_ksd(pfg_ctx *ctx, int K, int d, const double *x, const double *g, double c, double beta,
                double *ksd_out);

/* Host helper of the REPLAY path (no GPU involved): NumPy's legacy RandomState stream of one particle-filter
 * window, bit-identical to
 *     z0 = rs.normal(size=N);  for t in range(T): u[t] = rs.random_sample(N); z[t] = rs.normal(size=N)
 * -- the order in which the reference's filter consumes the global np.random state
 * (particle_filters/pf.py:26-38, kernels.py:83-100) -- generated natively (MT19937 + polar method in one tight
 * loop, the sqrt/log of the accepted pairs on `threads` worker threads; 0 = the default, 2).
 * The state is RandomState.get_state()'s (key[624], pos, has_gauss, cached_gaussian), advanced in place. */
int pfg_legacy_streams(uint32_t *key, int32_t *pos, int32_t *has_gauss, double *gauss, int N, int T,
                       double *z0, double *u, double *z, int threads);

/* Page-lock a caller-owned host buffer (hipHostRegister) so that pfg_run_batch stages it with a DMA straight
 * from where it lies: an input array of a pfg_problem that falls inside a registered range skips the pack copy
 * into the library's own pinned arena (the 2 x 8 MB replay streams of a T = N = 1000 window: 1.2 ms of a 5.6 ms
 * step).  The buffer must stay allocated and registered until pfg_host_unregister; registering a range twice
 * is an error.  Returns PFG_OK or a negative code (no message channel: there is no context here). */
int pfg_host_register(void *ptr, size_t bytes);
int pfg_host_unregister(void *ptr);

#ifdef __cplusplus
}
#endif
#endif /* PFGRAD_H */
