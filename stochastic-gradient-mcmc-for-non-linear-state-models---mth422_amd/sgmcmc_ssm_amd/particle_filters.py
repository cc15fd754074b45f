"""Buffered particle filter / smoother: Python face of the HIP kernels.

`buffered_pf_wrapper` keeps the call shape of the reference function of the same name
(particle_filters/buffered_smoother.py:156-199) but takes enumerated model / kernel /
statistic ids instead of Python callables (callables cannot cross to the device), and runs
the whole T-loop (:93-133) as ONE kernel launch in libpfgrad.so.

RNG modes
  rng='replay' (default): the N + 2*T*N draws the reference's loop would take from the global
      legacy `np.random` state are drawn here, in the same order (N normals for x0, then per
      timestep N uniforms for np.random.choice and N normals for Kernel.rv), and handed to the
      kernel.  Results then reproduce the reference on identical seeds (fp64).
  rng='device' (alias 'philox'): generated inside the kernel (one jsf32 per lane keyed
      by Philox4x32-10 of (seed, stream, step)): no host stream, no H2D traffic; the fast path,
      statistically equivalent, not seed-compatible with the reference.

There is no NumPy fallback: without libpfgrad.so and an MI355X every entry point raises.
"""
import os

import numpy as np

from . import _capi

PF_NAMES = ("nemeth", "poyiadjis_N", "poyiadjis_N2", "filter", "paris")


def _smoother_of(pf, kwargs):
    """pf name -> (smoother id, lambduh), as buffered_smoother.py:170-197 dispatches."""
    if pf == "nemeth":
        return "nemeth", float(kwargs.pop("lambduh", 0.95))
    if pf == "poyiadjis_N":
        kwargs.pop("lambduh", None)
        return "nemeth", 1.0
    if pf == "filter":
        kwargs.pop("lambduh", None)
        return "filter", 1.0
    if pf == "paris":
        kwargs.pop("lambduh", None)
        return "paris", 1.0
    if pf == "poyiadjis_N2":
        kwargs.pop("lambduh", None)
        return "poyiadjis_n2", 1.0
    raise ValueError("Unrecognized pf = {0}".format(pf))


# Stream buffers are recycled between calls: a fresh 16 MB ndarray costs ~45 ms of first-touch
# page faults, more than generating its contents (T = N = 1000).
_NATIVE_STREAM_MIN = 4096      # N*T from which the native generator is used (below: call overhead dominates)
_SPECULATE_MIN = 200000        # N*T from which the next stream is prefetched on a worker thread (below, starting
                               # and joining the thread costs more than the ~4 ns per draw it hides: measured
                               # 0.49 -> 0.79 ms per step at N*T = 24000, 8.5 -> 5.6 ms at 1e6)
_stream_pool = {}
_STREAM_POOL_MAX_BYTES = 2048 << 20     # a T = 48, N = 10^6 window (the reference's bias experiments) is 2 x 384 MB
# the prefetch worker (speculation) and the caller's thread both take and return buffers: one lock for the pool and
# the page-locked bookkeeping below
import threading
_pool_lock = threading.RLock()


# Large stream buffers are page-locked once (pfg_host_register) and then live in the pool for the rest of the
# process -- registered pages must never go back to the allocator -- so that pfg_run_batch stages them by DMA
# from where the generator wrote them instead of copying 16 MB into its own pinned arena first.
_PIN_MIN_BYTES = 1 << 20
_PIN_MAX_BYTES = 2048 << 20
_pinned = []                   # (u, z) pairs kept alive for good
_pinned_bytes = 0


def _stream_buffers(N, T):
    global _pinned_bytes
    with _pool_lock:
        free = _stream_pool.get((N, T))
        if free:
            return free.pop()
        u, z = np.empty((T, N)), np.empty((T, N))
        want_pin = u.nbytes >= _PIN_MIN_BYTES and _pinned_bytes + 2 * u.nbytes <= _PIN_MAX_BYTES
        if want_pin:
            _pinned_bytes += 2 * u.nbytes      # reserved under the lock: two threads cannot both pass the cap
    if want_pin:
        u.fill(0.0); z.fill(0.0)               # touch the pages before locking them
        ok = _capi.host_register(u)
        if ok and not _capi.host_register(z):
            _capi.host_unregister(u)
            ok = False
        with _pool_lock:
            if ok:
                _pinned.append((u, z))
            else:
                _pinned_bytes -= 2 * u.nbytes
    return u, z


def _is_pinned(bufs):
    return any(bufs[0] is p[0] for p in _pinned)


def _recycle_bufs(bufs):
    """Return one (u, z) pair to the pool: page-locked pairs always (registered pages must not go back to the
    allocator), pageable ones while the pool is below its cap."""
    if bufs is None:
        return
    with _pool_lock:
        held = sum(len(v) * 2 * k[0] * k[1] * 8 for k, v in _stream_pool.items())
        if held < _STREAM_POOL_MAX_BYTES or _is_pinned(bufs):
            _stream_pool.setdefault(bufs[0].shape[::-1], []).append(bufs)


def _recycle_streams(problems):
    """Return the u/z buffers of finished problems to the pool (the C call has copied them)."""
    for q in problems:
        _recycle_bufs(q.pop("_stream_bufs", None))


# ----------------------------------------------------------------------------------------------------
# Speculative prefetch of the NEXT window's replay stream.
# What a filter run takes from np.random does not depend on the gradient it returns, so while the GPU runs
# window k a worker thread can already produce window k+1's stream -- on a CLONE of the generator, advanced by
# whatever the caller says happens in between (SGLD noise, the next window's start).  The stream is adopted
# only if, when window k+1 is actually requested, the real np.random state is bit for bit the state the clone
# was in before it drew: any other use of np.random in between simply voids the speculation.
# ----------------------------------------------------------------------------------------------------
class _Speculation(object):
    def __init__(self):
        self.thread = None
        self.result = None
        self.adopted = self.discarded = 0       # statistics (tests, tuning)

    @staticmethod
    def _same_state(a, b):
        return (a[0] == b[0] and a[2] == b[2] and a[3] == b[3] and a[4] == b[4] and np.array_equal(a[1], b[1]))

    def start(self, between):
        """between(rs) -> (N, T) of the next stream or None; called on a clone of np.random's state."""
        self.cancel()
        state = np.random.get_state()

        def work():
            rs = np.random.RandomState()
            rs.set_state(state)
            shape = between(rs)
            if shape is None or shape[0] * shape[1] < _SPECULATE_MIN:
                return
            N, T = shape
            before = rs.get_state()
            bufs = _stream_buffers(N, T)
            z0 = np.empty(N)
            _capi.legacy_streams(rs, N, T, z0, bufs[0], bufs[1])
            self.result = (before, N, T, z0, bufs, rs.get_state())

        import threading
        self.result = None
        self.thread = threading.Thread(target=work, daemon=True)
        self.thread.start()

    def take(self, N, T):
        """The prefetched (z0, u, z) if it is exactly what np.random would deliver now, else None."""
        if self.thread is None:
            return None
        self.thread.join()
        self.thread, res, self.result = None, self.result, None
        if res is None:
            return None
        before, n, t, z0, bufs, after = res
        if n == N and t == T and self._same_state(before, np.random.get_state()):
            np.random.set_state(after)
            self.adopted += 1
            return z0, bufs
        self.discarded += 1
        _recycle_bufs(bufs)
        return None

    def cancel(self):
        if self.thread is not None:
            self.thread.join()
            if self.result is not None:
                _recycle_bufs(self.result[4])
            self.thread, self.result = None, None


speculation = _Speculation()
# a prefetch still writing into its buffers must not outlive the interpreter's teardown of those buffers
import atexit
atexit.register(speculation.cancel)


def draw_replay_streams(N, T, random_state=None, buffers=None):
    """Take from `random_state` (default: the global legacy np.random) exactly what one
    reference PF run takes, in its order.  Returns z0 (N,), u (T,N), z (T,N)."""
    rs = np.random if random_state is None else random_state
    if rs is np.random:
        got = speculation.take(N, T)
        if got is not None:
            _recycle_bufs(buffers)
            return got[0], got[1][0], got[1][1]
    u, z = buffers if buffers is not None else (np.empty((T, N)), np.empty((T, N)))
    if N * T >= _NATIVE_STREAM_MIN and (rs is np.random or isinstance(rs, np.random.RandomState)):
        # the same numbers, generated natively (libpfgrad's pfg_legacy_streams): NumPy's row-by-row
        # calls were 14 of the 16.9 ms of a T = N = 1000 step
        z0 = np.empty(N)
        _capi.legacy_streams(rs, N, T, z0, u, z)
        return z0, u, z
    z0 = rs.normal(size=N)
    for t in range(T):
        u[t] = rs.random_sample(N)
        z[t] = rs.normal(size=N)
    return z0, u, z


def draw_replay_streams_predictive(model, N, T, t1, tL, num_steps_ahead, random_state=None):
    """The legacy-stream consumption of one pf_predictive_loglikelihood_estimate run: as
    draw_replay_streams, plus -- inside the window, after the step's N normals -- N normals per lead
    k with t + k < T (svm/helper.py:380 `np.random.normal` inside svm_predictive_loglikelihood;
    garch/helper.py:405 prior_kernel.rv; LGSSM draws none).  Returns z0, u, z, pred_z [T][K+1][N]."""
    rs = np.random if random_state is None else random_state
    K1 = num_steps_ahead + 1
    z0 = rs.normal(size=N)
    u, z = np.empty((T, N)), np.empty((T, N))
    pred_z = None if model == "lgssm" else np.zeros((T, K1, N))
    for t in range(T):
        u[t] = rs.random_sample(N)
        z[t] = rs.normal(size=N)
        if pred_z is not None and t1 <= t < tL:
            for k in range(min(K1, T - t)):
                pred_z[t, k] = rs.normal(size=N)
    return z0, u, z, pred_z


def make_problem(model, kernel, pf, observations, theta, N, t1=0, tL=None, weights=None,
                 prior_mean=0.0, prior_var=1.0, stat="score", dtype="f64", rng="replay",
                 seed=None, stream=None, flags=0, random_state=None, ctx=None, **kwargs):
    """Build one problem dict for _capi.Context.run_batch.  `ctx`: the context a PaRIS window in np.random's order runs
    on (rng='replay', pf='paris': that window has to run HERE, while its draws are due -- see below; every other problem
    is only described, and runs where the caller hands it to run_batch)."""
    kwargs = dict(kwargs)
    kwargs.pop("tqdm", None)
    kwargs.pop("tqdm_name", None)
    resampling = kwargs.pop("resampling", "multinomial")
    if resampling == "systematic":
        # extension (the reference only resamples multinomially): one uniform per timestep
        if rng == "replay":
            raise ValueError("resampling='systematic' needs rng='device' (no reference stream to replay)")
    elif resampling != "multinomial":
        raise ValueError("Unrecognized resampling = {0}".format(resampling))
    smoother, lambduh = _smoother_of(pf, kwargs)
    if resampling == "systematic":
        if smoother != "nemeth":
            raise NotImplementedError("systematic resampling is built for pf = 'poyiadjis_N' | 'nemeth'")
        smoother = "nemeth_systematic"
    y = np.ascontiguousarray(observations, dtype=float)
    if y.ndim == 2:
        if y.shape[1] != 1:
            raise ValueError("the HIP particle filter supports m = 1 observations only")
        y = y[:, 0]
    T = y.shape[0]
    paris_kw = {}
    if smoother == "paris":
        # PaRIS (pf.py:183-341).  Its backward-sampling draws are data dependent in number and interleave with the
        # filter's draws in np.random's order.  rng='replay' (N <= 16384) reproduces that order exactly: the
        # window runs here and now, one kernel launch per timestep (_paris_replay_window), so that np.random.seed(s)
        # gives the reference's numbers and leaves the generator where the reference leaves it.  Explicit uniform
        # pools (tests) use the addressed form; rng='device' uses the device generator, keyed from
        # np.random (reproducible under np.random.seed, statistically equivalent to the reference).
        accept_reject = bool(kwargs.pop("accept_reject", True))
        mst = kwargs.pop("manual_sample_threshold", None)
        mar = kwargs.pop("max_accept_reject", None)
        paris_kw["Ntilde"] = int(kwargs.pop("Ntilde", 2))
        pools = [kwargs.pop(k, None) for k in ("paris_idx_u", "paris_acc_u", "paris_man_u")]
        if rng == "replay" and pools[2] is None and int(N) <= 16384 and stat in ("score", "suff", "none") and dtype == "f64":
            q = dict(model=model, kernel=kernel, smoother="paris", stat=stat, dtype=dtype, rng="replay", N=int(N),
                     t1=int(t1), tL=(T if tL is None else int(tL)), lambduh=1.0, Ntilde=paris_kw["Ntilde"],
                     prior_mean=float(np.asarray(prior_mean).reshape(-1)[0]),
                     prior_var=float(np.asarray(prior_var).reshape(-1)[0]), y=y, weights=weights, theta=theta, flags=flags)
            # the whole window in one launch (round 4: also 1024 < N <= 16384, the large-N kernel; one launch per timestep
            # -- _paris_replay_window -- stays available: PFGRAD_PARIS_PER_TIMESTEP=1, A/B and tests)
            if os.environ.get("PFGRAD_PARIS_PER_TIMESTEP") and int(N) > 1024:
                q["_result"] = _paris_replay_window(q, accept_reject, mar, mst, random_state, ctx=ctx)
            else:
                q["_result"] = _paris_raw_window(q, accept_reject, mar, mst, random_state, ctx=ctx)
            return q
        # default rounds: the reference stops accept-reject once <= 10 log10(N/10) children are
        # left and draws those exactly (its own cap is 100 log10(N/10) rounds); a fixed number of
        # rounds followed by the exact draw is the device equivalent.  The exact draw costs O(N)
        # per child; pending children sit in wave-local queues and the tail of rounds is taken
        # several rounds per pass, so extra rounds are nearly free: measured (tools/paris_perf.py,
        # SVM N = 1000) 61 / 47 / 43 / 44 us per timestep with 16 / 32 / 64 / 256 rounds; N = 10000
        # (exact draw = 10000 parents): 2196 / 659 / 400 us with 16 / 64 / 256 rounds.
        default_rounds = 64 if int(N) <= 1024 else 256
        paris_kw["max_accept_reject"] = default_rounds if mar is None else max(0, int(mar))
        if not accept_reject:
            # paris_smoother(accept_reject=False), pf.py:226-236: every child draws its parents from the exact backward
            # categorical -- no accept-reject rounds, the exact draw for every child
            paris_kw["max_accept_reject"] = 0
        if rng == "replay" and pools[2] is None:
            rng = "device"
        elif rng == "replay":
            paris_kw.update(paris_idx_u=pools[0], paris_acc_u=pools[1], paris_man_u=pools[2])
    q = dict(model=model, kernel=kernel, smoother=smoother, stat=stat, dtype=dtype, rng=rng,
             N=int(N), t1=int(t1), tL=(T if tL is None else int(tL)), lambduh=lambduh, **paris_kw,
             prior_mean=float(np.asarray(prior_mean).reshape(-1)[0]),
             prior_var=float(np.asarray(prior_var).reshape(-1)[0]),
             y=y, weights=weights, theta=theta, flags=flags)
    if stat == "predictive":
        if smoother != "filter":
            raise ValueError("Only can use pf = 'filter' since we are filtering")
        q["num_steps_ahead"] = int(kwargs.pop("num_steps_ahead", 5))
    if rng == "replay" and stat == "predictive":
        q["z0"], q["u"], q["z"], q["pred_z"] = draw_replay_streams_predictive(
            model, int(N), T, q["t1"], min(q["tL"], T), q["num_steps_ahead"], random_state)
    elif rng == "replay":
        bufs = _stream_buffers(int(N), T)
        q["z0"], q["u"], q["z"] = draw_replay_streams(int(N), T, random_state, buffers=bufs)
        q["_stream_bufs"] = (q["u"], q["z"])      # = bufs, or a prefetched pair (bufs then went back to the pool)
    elif rng in ("device", "philox"):
        # derive the device key AND stream from the host generator: (np.random state) -> (seed, stream)
        # is a pure function, so np.random.seed() reproduces device-generator runs within and
        # across processes
        rs = np.random if random_state is None else random_state
        if seed is None:
            seed = int(rs.randint(0, 2 ** 31 - 1)) | (int(rs.randint(0, 2 ** 31 - 1)) << 31)
        if stream is None:
            stream = int(rs.randint(0, 2 ** 31 - 1))
        q["seed"], q["stream"] = int(seed), int(stream)
    else:
        raise ValueError("Unrecognized rng = {0}".format(rng))
    return q


_paris_block_hint = {}          # (N, Ntilde) -> doubles a timestep's backward sampling consumed last time (stream block size)
_paris_raw_hint = {}            # (N, Ntilde, accept_reject) -> doubles per timestep a whole window consumed last time
_RAW_CHUNK = 1 << 16            # the raw stream is drawn in chunks with the generator state kept at every boundary


def _paris_raw_window(q, accept_reject=True, max_accept_reject=None, manual_sample_threshold=None, random_state=None, ctx=None):
    """One PaRIS window (N <= 16384) consuming the legacy generator EXACTLY as the reference does, in ONE launch
    (pfgrad.h: PFG_FLAG_PARIS_RAW_STREAM): the host hands the kernel what RandomState.random_sample delivers from the
    generator's current state, the kernel takes from it -- in np.random's order -- the normals of x0 and per timestep N
    uniforms, N normals (NumPy's legacy polar method on pairs of doubles, the second variate of a pair cached) and the
    backward sampling's uniforms, and reports how many doubles it consumed.  The generator is then put where the
    reference's stands: advanced by that many doubles, its cached Gaussian (if one is pending) recomputed here from the
    pair of doubles the kernel points at, with the host libm NumPy uses.  Returns what _paris_replay_window returns
    minus the all_* traces (buffered_pf_wrapper(save_all=True) re-runs the launch on the same doubles for those)."""
    import math
    ctx = ctx or _capi.default_context()
    rs = np.random if random_state is None else random_state
    N, Nt, y = q["N"], q["Ntilde"], q["y"]
    T = y.shape[0]
    h = _capi.STAT_DIM[q["model"]] if q["stat"] == "score" else 3
    mar = int(100 * np.log10(N / 10)) if max_accept_reject is None else int(max_accept_reject)
    mst = int(10 * np.log10(N / 10)) if manual_sample_threshold is None else int(manual_sample_threshold)
    mar, mst = max(mar, 0), max(mst, 0)
    state0 = rs.get_state()
    has_gauss, cached = int(state0[3]), float(state0[4])
    flags = int(q.get("flags", 0)) | _capi.FLAG_PARIS_RAW_STREAM
    flags |= 0 if accept_reject else _capi.FLAG_PARIS_NO_ACCEPT_REJECT
    flags |= _capi.FLAG_PARIS_RAW_CARRY if has_gauss else 0
    key = (N, Nt, bool(accept_reject))
    normals = int(1.36 * N) + 64                     # doubles of one call: N / 2 pairs accepted with probability pi / 4
    per_step = _paris_raw_hint.get(key, 0) * 1.2 or (N + normals + (N * Nt if not accept_reject else 8 * N * Nt))
    ahead = 4096 if N <= 1024 else 16384                  # the kernel's look-ahead: one round of attempts (1024 | 4096 pairs of doubles)
    L = int(normals + T * per_step) + 4 * 1024 + ahead
    while True:
        L = (L + _RAW_CHUNK - 1) // _RAW_CHUNK * _RAW_CHUNK
        stream = np.empty(L + 1)
        stream[0] = cached
        states = []
        for c0 in range(0, L, _RAW_CHUNK):
            states.append(rs.get_state())
            stream[1 + c0:1 + c0 + _RAW_CHUNK] = rs.random_sample(_RAW_CHUNK)
        s0 = 0 if has_gauss else 1                  # without a cached Gaussian the doubles start at stream[1]
        run = dict(model=q["model"], kernel=q["kernel"], smoother="paris", stat=q["stat"], dtype="f64", rng="replay", N=N,
                   t1=q["t1"], tL=q["tL"], lambduh=1.0, theta=q["theta"], prior_mean=q["prior_mean"], prior_var=q["prior_var"],
                   y=y, weights=q.get("weights", None), Ntilde=Nt, max_accept_reject=mar, paris_manual_threshold=mst,
                   paris_stream=stream[s0:], flags=flags)
        o = ctx.run_batch([run], want_final=True)[0]
        used = o["paris_consumed"]
        if used >= 0:
            break
        rs.set_state(state0)
        if used != -1:
            raise RuntimeError("PaRIS whole-window stream: the kernel reported {0} doubles consumed".format(used))
        # -1 = the stream ran out: the same window again on a longer one.  The consumption is bounded: per timestep N
        # uniforms, < 2.6 N doubles of normal attempts with overwhelming probability, and at most
        # 2 N Ntilde max_accept_reject + N Ntilde doubles of backward sampling
        bound = int((T + 1) * (4 * N + 2 * N * Nt * max(mar, 1) + N * Nt)) + 2 * _RAW_CHUNK
        if L > 4 * bound:
            raise RuntimeError("PaRIS whole-window stream: {0} doubles were not enough for a window whose consumption is "
                               "bounded by {1}".format(L, bound))
        L *= 2
    drawn = used - (1 if has_gauss else 0)          # doubles taken from the generator
    _paris_raw_hint[key] = drawn / max(T, 1)
    c = min(drawn // _RAW_CHUNK, len(states) - 1)
    rs.set_state(states[c])
    if drawn - c * _RAW_CHUNK > 0:
        rs.random_sample(drawn - c * _RAW_CHUNK)    # the generator now stands where the reference's stands ...
    st = rs.get_state()
    back = o["paris_carry_back"]
    if back > 0:                                    # ... including legacy_gauss' cached second variate
        d0, d1 = stream[s0 + used - back], stream[s0 + used - back + 1]
        x1, x2 = 2.0 * d0 - 1.0, 2.0 * d1 - 1.0
        r2 = x1 * x1 + x2 * x2
        rs.set_state((st[0], st[1], st[2], 1, math.sqrt(-2.0 * math.log(r2) / r2) * x1))
    else:
        rs.set_state((st[0], st[1], st[2], 0, 0.0))
    stats = o["statistics"][:, :h]
    return dict(mean_stat=np.asarray(o["mean_stat"])[:h], loglik=o["loglik"], x_t=o["x_t"], log_weights=o["log_weights"],
                statistics=stats, _draws=dict(paris_stream=stream[s0:s0 + used + ahead].copy(), flags=flags, max_accept_reject=mar,
                                              paris_manual_threshold=mst, _consumed=used))     # (+ the kernel's look-ahead)


def _paris_replay_window(q, accept_reject=True, max_accept_reject=None, manual_sample_threshold=None, random_state=None, ctx=None):
    """One PaRIS window consuming the legacy generator EXACTLY as the reference does (paris_smoother +
    accept_reject_based_backward_sampling, pf.py:183-341): N normals for x0, then per timestep N uniforms
    (np.random.choice), N normals (Kernel.rv) and the data-dependent run of uniforms of the backward sampling.
    One kernel launch per timestep, warm-started from the previous step's particles: the host draws the filter's
    u / z, hands the kernel a block of further uniforms (pfg_problem.paris_stream), and afterwards rewinds the
    generator to the end of the filter's draws and advances it by exactly what the kernel reports as consumed.
    Returns the dict run_batch would (mean_stat, loglik, x_t, log_weights, statistics, all_* traces)."""
    ctx = ctx or _capi.default_context()
    rs = np.random if random_state is None else random_state
    N, Nt, y = q["N"], q["Ntilde"], q["y"]
    T, t1, tL = y.shape[0], q["t1"], min(q["tL"], y.shape[0])
    ns, h = _capi.STATE_DIM[q["model"]], (_capi.STAT_DIM[q["model"]] if q["stat"] == "score" else 3)
    # pf.py:282-285 (range() of a negative count is empty, a negative threshold is never reached)
    mar = int(100 * np.log10(N / 10)) if max_accept_reject is None else int(max_accept_reject)
    mst = int(10 * np.log10(N / 10)) if manual_sample_threshold is None else int(manual_sample_threshold)
    mar, mst = max(mar, 0), max(mst, 0)
    base = dict(model=q["model"], kernel=q["kernel"], stat=q["stat"], dtype="f64", rng="replay", N=N, lambduh=1.0,
                theta=q["theta"], prior_mean=q["prior_mean"], prior_var=q["prior_var"])
    flags = int(q.get("flags", 0))
    # x0 (LatentGaussianKernel.sample_x0, kernels.py:83-100): a T = 0 window returns the initial particle system
    z0 = rs.normal(size=N)
    o = ctx.run_batch([dict(base, smoother="nemeth", flags=flags, y=np.zeros(0), t1=0, tL=0, z0=z0, u=np.zeros(0), z=np.zeros(0))],
                      want_final=True)[0]
    x, logw, stats = o["x_t"], o["log_weights"], np.zeros((N, _capi.STAT_DIM[q["model"]]))
    all_x, all_lw, all_st, all_ll = [x.copy()], [logw.copy()], [stats[:, :h].copy()], [0.0]
    ll, mean_stat = 0.0, np.zeros(h)
    pflags = flags | (0 if accept_reject else _capi.FLAG_PARIS_NO_ACCEPT_REJECT)
    us, zs, consumed = [], [], []       # the draws of this window, for a second launch on the same numbers (elementwise pass)
    for t in range(T):
        u = rs.random_sample(N)
        z = rs.normal(size=N)
        inside = t1 <= t < tL
        w = None
        if inside and q.get("weights", None) is not None:
            w = np.array([float(np.asarray(q["weights"]).reshape(-1)[t - t1])])
        M = N * Nt if not accept_reject else max(_paris_block_hint.get((N, Nt), 0) * 2, 8 * N * Nt) + 64
        after_filter = rs.get_state()
        while True:
            block = rs.random_sample(M)
            step = dict(base, smoother="paris", flags=pflags, y=y[t:t + 1], t1=0, tL=1 if inside else 0, weights=w,
                        init_x=x, init_logw=logw, init_stats=stats, u=u.reshape(1, N), z=z.reshape(1, N),
                        Ntilde=Nt, max_accept_reject=mar, paris_stream=block, paris_manual_threshold=mst)
            o = ctx.run_batch([step], want_final=True)[0]
            used = o["paris_consumed"]
            rs.set_state(after_filter)
            if used >= 0:
                break
            if used != -1 or M > 64 * (2 * N * Nt * max(mar, 1) + N * Nt + 64):
                raise RuntimeError("PaRIS replay: the kernel reported {0} for a block of {1} uniforms".format(used, M))
            M *= 4                      # the block ran out: the same step again with a longer one
        if used > 0:
            rs.random_sample(used)      # the generator now stands where the reference's stands
        us.append(u); zs.append(z); consumed.append(block[:max(used, 0)].copy())     # (a copy: a view would keep the whole block alive)
        _paris_block_hint[(N, Nt)] = max(int(used), 1)
        x, logw = o["x_t"], o["log_weights"]
        stats = np.zeros((N, _capi.STAT_DIM[q["model"]]))
        stats[:, :o["statistics"].shape[1]] = o["statistics"]
        ll += o["loglik"]
        mean_stat = o["mean_stat"]
        all_x.append(x.copy()); all_lw.append(logw.copy()); all_st.append(stats[:, :h].copy()); all_ll.append(ll)
    return dict(mean_stat=np.asarray(mean_stat)[:h], loglik=ll, x_t=x, log_weights=logw, statistics=stats[:, :h],
                all_x_t=np.array(all_x), all_log_weights=np.array(all_lw), all_statistics=np.array(all_st),
                all_loglikelihood_estimate=np.array(all_ll),
                _draws=dict(z0=z0, u=np.array(us).reshape(T, N), z=np.array(zs).reshape(T, N),
                            paris_stream=np.concatenate(consumed) if consumed else np.zeros(0), flags=pflags,
                            max_accept_reject=mar, paris_manual_threshold=mst))


def paris_replay_again(q):
    """The problem dict that re-runs a window _paris_replay_window has run, in ONE launch on the same random
    numbers: the filter's z0 / u / z and, as paris_stream, exactly the uniforms the backward sampling consumed
    (the kernel carries its stream cursor across timesteps).  Same ancestors, same parents, same numbers; the
    caller adds what it wants recorded on top (Helper.pf_latent_var_distr: the elementwise statistics)."""
    d = q["_result"]["_draws"]
    return dict({k: v for k, v in q.items() if k != "_result"}, **d)


def buffered_pf_wrapper(pf, model, kernel, observations, theta, N, ctx=None,
                        save_all=False, want_final=True, **kwargs):
    """Run one buffered PF window on the GPU.

    Returns the reference's dict: x_t (N,n), log_weights (N,), statistics ((N,h), or (h,) for
    pf='filter'), loglikelihood_estimate, plus mean_statistic (= average_statistic(out)) and,
    with save_all=True, the all_* traces of buffered_smoother.py:128-142."""
    ctx = ctx or _capi.default_context()
    q = make_problem(model, kernel, pf, observations, theta, N, ctx=ctx, **kwargs)
    if "_result" in q:                  # PaRIS in np.random's order: the window ran while its draws were due
        if save_all and "all_x_t" not in q["_result"]:      # (one launch, no traces kept: the same launch again with them)
            q2 = paris_replay_again(q)
            o = ctx.run_batch([q2], want_final=True, want_trace=True)[0]
            expected = q2.get("_consumed", q2["paris_stream"].shape[0])
            if o["paris_consumed"] != expected:       # the traces must be those of the trajectory that advanced np.random
                raise RuntimeError("PaRIS replay consumed {0} of {1} uniforms".format(o["paris_consumed"], expected))
            return _to_reference_dict(o, q)
        return _to_reference_dict(q["_result"], q)
    o = ctx.run_batch([q], want_final=want_final or save_all, want_trace=save_all)[0]
    out = _to_reference_dict(o, q)
    _recycle_streams([q])
    return out


def _to_reference_dict(o, q):
    out = dict(loglikelihood_estimate=o["loglik"])
    if "predictive" in o:
        out["statistics"] = o["predictive"]
    elif q["smoother"] == "filter":
        out["statistics"] = o["mean_stat"]
    else:
        out["mean_statistic"] = o["mean_stat"]
        if "statistics" in o:
            out["statistics"] = o["statistics"]
    for src, dst in (("x_t", "x_t"), ("log_weights", "log_weights"), ("all_x_t", "all_x_t"),
                     ("all_log_weights", "all_log_weights"), ("all_statistics", "all_statistics"),
                     ("all_loglikelihood_estimate", "all_loglikelihood_estimate")):
        if src in o:
            out[dst] = o[src]
    return out


def run_windows(problems, ctx=None, want_final=False):
    """Many independent windows (same model/kernel/dtype/rng) in ONE launch, one workgroup each."""
    ctx = ctx or _capi.default_context()
    todo = [q for q in problems if "_result" not in q]
    outs = iter(ctx.run_batch(todo, want_final=want_final) if todo else [])
    res = [_to_reference_dict(q["_result"] if "_result" in q else next(outs), q) for q in problems]
    _recycle_streams(problems)
    return res


def average_statistic(out):
    """sum_i statistics_i * softmax(log_weights)_i  (buffered_smoother.py:151-154); the kernel
    already returns it as out['mean_statistic']."""
    if "mean_statistic" in out:
        return out["mean_statistic"]
    lw = out["log_weights"]
    p = np.exp(lw - np.max(lw))
    p /= np.sum(p)
    return np.sum(out["statistics"].T * p, axis=1)
