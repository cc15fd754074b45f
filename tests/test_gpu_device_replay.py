"""Deterministic parity of the DEVICE-generator kernels -- the instantiations bench.py times.

The REPLAY kernels are pinned to the reference through NumPy's legacy stream; the device-generator
kernels live in other translation units (-ffp-contract=fast -DPFG_FAST_ALGEBRA: fused multiply-adds,
a cubic expm1, a 32-bit fixed-point resampling CDF in thread-major order searched with the raw
generator word -- or, in the large-N kernel, sorted uniforms built from exponential spacings --,
Gaussian draws from the f32 transcendental units), so REPLAY parity says nothing
about them.  Here the SAME instantiation (same template arguments, same code object) additionally
writes out the random inputs it consumed -- per step and child the 32-bit word it searched the CDF
with and its standard normal, plus the x0 normals (pfg_result.rec_u / rec_z / rec_z0) -- and the
CPU oracle replays the launch on those numbers: `po.pf_window` (the reference's T-loop, pinned to
the reference by tests/test_oracle_golden.py) with its resampling step replaced by
`po.device_ancestors` (the kernels' CDF layout).  Everything downstream of the random inputs is
deterministic, so trajectories, ancestors, statistics, log-likelihood and gradient must agree.

Tolerance: rtol 1e-8 (fused multiply-adds ~1e-16, cubic expm1 < 3e-12 relative, parallel prefix
sums).  An ancestor flips only if a 32-bit word lands within ~3e-12 * 2^32 of a CDF entry:
probability ~1e-5 per 1e6 draws; asserted exactly.
"""
import numpy as np
import pytest

from oracle import pf_oracle as po

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-8, 1e-8

THETA = {
    "svm": np.array([0.95, 0.5 ** -0.5, 0.5 ** -0.5]),
    "lgssm": np.array([0.9, 1.0, 0.7 ** -0.5, 1.0]),
    "garch": None,      # filled below from (alpha, beta, gamma) = (.1, .8, .05), R = .3
}


def _garch_theta():
    alpha, beta, gamma, R = 0.1, 0.8, 0.05, 0.3
    phi = beta + gamma
    lam = beta / phi
    mu = alpha / (1.0 - phi)
    logit = lambda p: np.log(p / (1.0 - p))
    return np.array([np.log(mu), logit(phi), logit(lam), R ** -0.5])


THETA["garch"] = _garch_theta()


def _series(model, T, seed):
    """Synthetic observations of roughly the model's scale (parity does not need model-exact data)."""
    rs = np.random.RandomState(seed)
    if model == "svm":
        x = np.zeros(T)
        for t in range(1, T):
            x[t] = 0.95 * x[t - 1] + np.sqrt(0.5) * rs.normal()
        return np.exp(x / 2) * np.sqrt(0.5) * rs.normal(size=T)
    if model == "lgssm":
        x = np.zeros(T)
        for t in range(1, T):
            x[t] = 0.9 * x[t - 1] + np.sqrt(0.7) * rs.normal()
        return x + rs.normal(size=T)
    return 0.8 * rs.normal(size=T)


# model, kernel, pf, lambduh, N, T, (t1, tL, weights?), forced variant, (NT, PPT, cdf)
CASES = [
    # BASELINE configs[1] = the bench workload, on the bench instantiation pf_reg_kernel<0,0,double,256,4,1,false,0>
    ("svm", "prior", "poyiadjis_N", 1.0, 1000, 1000, None, "wg256x4s", (256, 4, "fixed32")),
    ("svm", "prior", "nemeth", 0.95, 1000, 120, (10, 100, True), "wg256x4s", (256, 4, "fixed32")),
    ("svm", "prior", "filter", 1.0, 777, 60, None, "wg256x4s", (256, 4, "fixed32")),
    ("svm", "prior", "poyiadjis_N", 1.0, 1000, 200, None, "wg1024x1", (1024, 1, "fixed32")),   # one chain alone
    ("svm", "prior", "poyiadjis_N", 1.0, 900, 80, None, "wg256x4", (256, 4, "fixed32")),
    ("svm", "prior", "poyiadjis_N", 1.0, 200, 100, None, "wg256x1", (256, 1, "fixed32")),
    # config 1: LGSSM T=200 N=100 (one wave per window)
    ("lgssm", "optimal", "poyiadjis_N", 1.0, 100, 200, None, "wg64x2", (64, 2, "fixed32")),
    ("lgssm", "prior", "nemeth", 0.9, 128, 50, None, "wg64x2", (64, 2, "fixed32")),
    # 128 < N <= 256 in batches: one wave per window, four particles per lane
    ("lgssm", "optimal", "poyiadjis_N", 1.0, 200, 80, None, "wg64x4", (64, 4, "fixed32")),
    ("svm", "prior", "nemeth", 0.95, 256, 40, (5, 30, True), "wg64x4", (64, 4, "fixed32")),
    # config 3: GARCH N=1000, S=16 B=4 window
    ("garch", "optimal", "poyiadjis_N", 1.0, 1000, 24, (4, 20, True), "wg512x2s", (512, 2, "fixed32")),
    ("garch", "prior", "poyiadjis_N", 1.0, 1000, 24, (4, 20, True), "wg512x2s", (512, 2, "fixed32")),
    ("garch", "optimal", "nemeth", 0.9, 1024, 40, None, "wg512x2s", (512, 2, "fixed32")),
    ("garch", "optimal", "poyiadjis_N", 1.0, 1000, 24, (4, 20, True), "wg256x4s", (256, 4, "fixed32")),
    # config 4: SVM N=4000, LDS-resident 1024 x 4
    ("svm", "prior", "poyiadjis_N", 1.0, 4000, 1000, None, "wg1024x4s", (1024, 4, "fixed32")),   # full size
    # one wave per window on a single state buffer
    ("lgssm", "optimal", "poyiadjis_N", 1.0, 100, 200, None, "wg64x2s", (64, 2, "fixed32")),
    ("lgssm", "prior", "nemeth", 0.9, 128, 50, (5, 40, True), "wg64x2s", (64, 2, "fixed32")),
    ("garch", "optimal", "poyiadjis_N", 1.0, 77, 40, None, "wg64x2s", (64, 2, "fixed32")),
    ("lgssm", "optimal", "poyiadjis_N", 1.0, 200, 80, None, "wg64x4s", (64, 4, "fixed32")),
    ("svm", "prior", "filter", 1.0, 256, 40, None, "wg64x4s", (64, 4, "fixed32")),
    # config 5: SVM N=10000, S=16 B=4 window, large-N kernel (fp64 CDF)
    ("svm", "prior", "poyiadjis_N", 1.0, 10000, 24, (4, 20, True), "big16384", (16384, 1, "f64_uniform")),
    ("svm", "prior", "poyiadjis_N", 1.0, 9001, 30, None, "big16384", (16384, 1, "f64_uniform")),
    ("garch", "optimal", "poyiadjis_N", 1.0, 4000, 40, None, "big4096", (4096, 1, "f64_uniform")),
    ("lgssm", "optimal", "nemeth", 0.95, 3000, 40, None, "big4096", (4096, 1, "f64_uniform")),
    # edges: every slot used, one slot used beyond a power of two, ragged N, a single timestep, the maximum N
    ("svm", "prior", "poyiadjis_N", 1.0, 1024, 60, None, "wg256x4s", (256, 4, "fixed32")),
    ("svm", "prior", "poyiadjis_N", 1.0, 129, 40, None, "wg256x1", (256, 1, "fixed32")),
    ("lgssm", "prior", "poyiadjis_N", 1.0, 65, 30, None, "wg64x2", (64, 2, "fixed32")),
    ("svm", "prior", "poyiadjis_N", 1.0, 1000, 1, None, "wg256x4s", (256, 4, "fixed32")),
    ("svm", "prior", "poyiadjis_N", 1.0, 4096, 20, None, "wg1024x4s", (1024, 4, "fixed32")),
    ("svm", "prior", "poyiadjis_N", 1.0, 1025, 20, None, "wg1024x4s", (1024, 4, "fixed32")),
    ("garch", "prior", "nemeth", 0.9, 4097, 12, (2, 9, True), "big16384", (16384, 1, "f64_uniform")),
    ("svm", "prior", "poyiadjis_N", 1.0, 16384, 6, None, "big16384", (16384, 1, "f64_uniform")),
    ("lgssm", "optimal", "filter", 1.0, 1500, 10, None, "big4096", (4096, 1, "f64_uniform")),
]


@pytest.fixture(scope="module")
def ctx():
    from sgmcmc_ssm_amd import _capi
    return _capi.default_context(0)


def _assert_twin_statistics(production, traced, variant, where=None):
    """The production launch against the traced kernel the oracle has just replayed, same key.  The trace-free twin is the
    SAME code with the instrumentation compiled out: bitwise.  A Poyiadjis-score-only twin (`*_score1`) is another
    specialisation of the unit: same draws, same ancestors, but the compiler fuses multiply-adds of the score differently
    where the general branch is gone -- equal to a few units in the last place (1e-13), not always bitwise."""
    if variant.endswith("_score1"):
        np.testing.assert_allclose(production, traced, rtol=1e-13, atol=1e-13 * max(1.0, float(np.abs(traced).max())), err_msg=str(where))
    else:
        assert np.array_equal(production, traced), where


@pytest.mark.parametrize("case", CASES, ids=lambda c: "{0}-{1}-{2}-N{4}-T{5}-{7}".format(*c))
def test_device_kernel_replayed_by_oracle(ctx, monkeypatch, case):
    model, kernel, pf, lam, N, T, window, variant, (NT, PPT, cdf) = case
    theta = THETA[model]
    y = _series(model, T, seed=N + T)
    t1, tL, weights = 0, T, None
    if window is not None:
        t1, tL = window[0], window[1]
        weights = np.linspace(20.0, 30.0, tL - t1) if window[2] else None
    if model == "garch":
        pm, pv = po.garch_prior_x(theta)
        pv = float(np.asarray(pv).reshape(-1)[0])
    else:
        pm, pv = 0.0, 10.0
    smoother = "filter" if pf == "filter" else "nemeth"
    q = dict(model=model, kernel=kernel, smoother=smoother, stat="score", dtype="f64", rng="device", N=N,
             t1=t1, tL=tL, lambduh=lam, prior_mean=pm, prior_var=pv, y=y, weights=weights, theta=theta,
             seed=20241004 + N, stream=T)
    monkeypatch.setenv("PFGRAD_VARIANT", variant)
    o = ctx.run_batch([q], want_trace=True, want_draws=True)[0]
    assert ctx.last_variant() == variant          # the instantiation under test really ran
    assert ctx.last_traced()
    # the production twin (trace instrumentation compiled out of the T-loop; what bench.py and every resident run
    # launch): same key -> bitwise the same statistics (the log-likelihood sum is flushed every 64 steps instead of
    # every step: same terms, other rounding)
    plain = ctx.run_batch([dict(q)])[0]
    # (a window that is the Poyiadjis O(N) score runs the 1024 x 4 unit's specialised twin, PFG_SMOOTHER_POYIADJIS_N)
    score = smoother == "nemeth" and lam == 1.0
    twin = variant
    if score and (variant in ("wg1024x4s", "wg64x2s") or (variant == "wg1024x1" and model != "garch")):
        twin = variant + "_score1"
    assert ctx.last_variant() == twin and ctx.last_traced() == (not variant.startswith("wg"))
    _assert_twin_statistics(plain["mean_stat"], o["mean_stat"], twin)
    assert abs(plain["loglik"] - o["loglik"]) <= 1e-12 * abs(o["loglik"])

    words, z, z0 = o["rec_u"], o["rec_z"], o["rec_z0"]
    if cdf == "f64_uniform":
        # large-N kernel: the recorded resampling inputs are the sorted uniforms themselves (CDF, ranks and
        # storage all in particle order); per step they must be what they claim to be: in (0, 1) and
        # increasing in child order
        words = o["rec_ud"]
        assert np.all((words > 0.0) & (words < 1.0))
        child = np.arange(N)
        rank = (child % NT) * PPT + child // NT
        assert np.all(np.diff(words[:, np.argsort(rank)], axis=1) >= 0.0)
    assert np.all(np.isfinite(z)) and np.all(np.isfinite(z0)) and np.any(words != 0)
    ref = po.pf_window(model, theta, y, N, z0, None, z, kernel=kernel, pf=pf, lambduh=lam, stat="score",
                       t1=t1, tL=tL, weights=weights, prior_mean=pm, prior_var=pv, save_all=True,
                       resampler=lambda t, logw: po.device_ancestors(logw, words[t], NT, PPT, cdf))

    flips = int(np.sum(o["all_ancestors"] != ref["all_ancestors"]))
    assert flips == 0, "{0} ancestor indices differ".format(flips)
    np.testing.assert_allclose(o["all_x_t"], ref["all_x_t"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(o["all_log_weights"], ref["all_log_weights"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(o["all_loglikelihood_estimate"], ref["all_loglikelihood_estimate"], rtol=RTOL, atol=ATOL)
    if pf != "filter":
        np.testing.assert_allclose(o["all_statistics"], ref["all_statistics"], rtol=RTOL, atol=1e-7)
        np.testing.assert_allclose(o["mean_stat"], ref["mean_statistic"], rtol=RTOL, atol=1e-7)
        # the north-star bar, on the timed kernel: gradient L2 error < 1e-4
        assert np.linalg.norm(o["mean_stat"] - ref["mean_statistic"]) < 1e-6 * max(1.0, np.linalg.norm(ref["mean_statistic"]))
    else:
        np.testing.assert_allclose(o["mean_stat"], ref["statistics"], rtol=RTOL, atol=1e-7)
    np.testing.assert_allclose(o["loglik"], ref["loglikelihood_estimate"], rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("model,kernel,variant,N", [("svm", "prior", "wg256x4s", 1000), ("garch", "optimal", "wg256x4s", 600),
                                                  ("lgssm", "optimal", "wg64x2", 100), ("svm", "prior", "wg1024x1", 1000)])
def test_outlier_observations_on_the_device_units(ctx, monkeypatch, model, kernel, variant, N):
    """Observations no particle explains (tens to hundreds of sigma out): log-weights of -1e2 .. -1e5.  The
    device units must give the same trajectory as the oracle (exact-max log_normalize, pf.py:374-377).
    Where the reference's own log-likelihood would be -inf (it is not max-stabilised,
    buffered_smoother.py:124-126) the kernel's stays finite (DESIGN deviation (i))."""
    T = 40
    y = _series(model, T, seed=77)
    y[11] = 400.0 if model != "svm" else 60.0
    y[25] = -300.0 if model != "svm" else -45.0
    theta = THETA[model]
    if model == "garch":
        pm, pv = po.garch_prior_x(theta)
        pv = float(np.asarray(pv).reshape(-1)[0])
    else:
        pm, pv = 0.0, 10.0
    NT, PPT = {"wg256x4s": (256, 4), "wg64x2": (64, 2), "wg1024x1": (1024, 1)}[variant]
    q = dict(model=model, kernel=kernel, smoother="nemeth", stat="score", dtype="f64", rng="device", N=N, t1=0, tL=T,
             lambduh=1.0, prior_mean=pm, prior_var=pv, y=y, theta=theta, seed=31337, stream=N)
    monkeypatch.setenv("PFGRAD_VARIANT", variant)
    o = ctx.run_batch([q], want_trace=True, want_draws=True)[0]
    assert ctx.last_variant() == variant
    words = o["rec_u"]
    with np.errstate(divide="ignore"):
        ref = po.pf_window(model, theta, y, N, o["rec_z0"], None, o["rec_z"], kernel=kernel, pf="poyiadjis_N", stat="score",
                           prior_mean=pm, prior_var=pv, save_all=True,
                           resampler=lambda t, logw: po.device_ancestors(logw, words[t], NT, PPT, "fixed32"))
    assert int(np.sum(o["all_ancestors"] != ref["all_ancestors"])) == 0
    np.testing.assert_allclose(o["all_x_t"], ref["all_x_t"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(o["all_log_weights"], ref["all_log_weights"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(o["mean_stat"], ref["mean_statistic"], rtol=RTOL, atol=1e-7)
    assert np.isfinite(o["loglik"]) and np.all(np.isfinite(o["mean_stat"]))
    assert np.max(o["all_log_weights"][12]) < -100.0           # the outlier step really is one


@pytest.mark.parametrize("model,kernel,N,T,variant,layout", [
    ("svm", "prior", 1000, 24, "wg256x4", (256, 4, "fixed32")),          # f32 default: ping-pong
    ("garch", "optimal", 1000, 24, "wg256x4", (256, 4, "fixed32")),
    ("lgssm", "optimal", 100, 40, "wg64x2", (64, 2, "fixed32")),
    ("svm", "prior", 4000, 16, "wg1024x4s", (1024, 4, "fixed32"))])
def test_f32_state_device_kernels_replayed(ctx, monkeypatch, model, kernel, N, T, variant, layout):
    """dtype='f32' (particle state and statistics in f32, weights / CDF / search in f64): the same recorded draws,
    checked TEACHER-FORCED step by step at f32 tolerance -- from the kernel's own traced particles and log-weights of
    step t the oracle's resampling (on the recorded words), proposal, weight and statistic give step t + 1.  A whole-
    trajectory replay is not meaningful in f32: an f32 weight differs from the oracle's by ~1e-6, sooner or later one
    ancestor flips, that child's weight changes by O(1) and the two trajectories part (SURVEY finding 2); per step a
    handful of flips in T x N draws is what that rounding allows."""
    NT, PPT, cdf = layout
    theta = THETA[model]
    y = _series(model, T, seed=3 * N + T)
    if model == "garch":
        pm, pv = po.garch_prior_x(theta)
        pv = float(np.asarray(pv).reshape(-1)[0])
    else:
        pm, pv = 0.0, 10.0
    q = dict(model=model, kernel=kernel, smoother="nemeth", stat="score", dtype="f32", rng="device", N=N, t1=0, tL=T,
             lambduh=1.0, prior_mean=pm, prior_var=pv, y=y, theta=theta, seed=777 + N, stream=T)
    monkeypatch.setenv("PFGRAD_VARIANT", variant)
    o = ctx.run_batch([q], want_trace=True, want_draws=True)[0]
    assert ctx.last_variant() == variant
    words, z = o["rec_u"], o["rec_z"]
    d = po.derived(model, theta)
    flips = 0
    for t in range(T):
        x, lw, st = o["all_x_t"][t], o["all_log_weights"][t], o["all_statistics"][t]
        anc = po.device_ancestors(lw, words[t], NT, PPT, cdf)
        got = o["all_ancestors"][t]
        flips += int(np.sum(anc != got))
        yt = np.array([y[t]])
        xp = x[got]
        xn = po.kernel_rv(model, kernel, d, xp, yt, z[t])
        np.testing.assert_allclose(o["all_x_t"][t + 1], xn, rtol=2e-5, atol=2e-5)
        np.testing.assert_allclose(o["all_log_weights"][t + 1], po.kernel_reweight(model, kernel, d, xp, xn, yt), rtol=2e-4, atol=2e-4)
        ref_st = st[got] + po.score_statistic(model, d, xp, xn, yt)
        scale = np.maximum(1.0, np.abs(ref_st).max())
        assert np.max(np.abs(o["all_statistics"][t + 1] - ref_st)) < 2e-4 * scale
    assert flips <= max(3, int(2e-4 * T * N)), flips
    ref0 = pm + np.sqrt(pv) * o["rec_z0"]
    np.testing.assert_allclose(o["all_x_t"][0][:, 0], ref0, rtol=2e-6, atol=2e-6)


def test_recording_does_not_change_the_launch(ctx, monkeypatch):
    """The recorded launch is the timed launch: with and without the record / trace buffers the
    same (seed, stream) gives bitwise the same gradient (the log-likelihood sum is flushed every step
    instead of every 64 steps when its running value is traced: same terms, other rounding)."""
    monkeypatch.setenv("PFGRAD_VARIANT", "wg256x4s")
    y = _series("svm", 300, seed=5)
    q = dict(model="svm", kernel="prior", smoother="nemeth", stat="score", dtype="f64", rng="device", N=1000,
             t1=0, tL=300, lambduh=1.0, prior_mean=0.0, prior_var=10.0, y=y, theta=THETA["svm"], seed=11, stream=3)
    a = ctx.run_batch([dict(q)])[0]
    b = ctx.run_batch([dict(q)], want_trace=True, want_draws=True)[0]
    assert np.array_equal(a["mean_stat"], b["mean_stat"])
    assert abs(a["loglik"] - b["loglik"]) <= 1e-12 * abs(a["loglik"])


def test_recorded_draws_are_standard(ctx, monkeypatch):
    """The recorded inputs themselves: 1e6 normals (f32-unit Box-Muller) and 1e6 32-bit words."""
    from scipy import stats
    monkeypatch.setenv("PFGRAD_VARIANT", "wg256x4s")
    T, N = 1000, 1000
    y = _series("svm", T, seed=9)
    q = dict(model="svm", kernel="prior", smoother="nemeth", stat="score", dtype="f64", rng="device", N=N,
             t1=0, tL=T, lambduh=1.0, prior_mean=0.0, prior_var=10.0, y=y, theta=THETA["svm"], seed=99, stream=1)
    o = ctx.run_batch([q], want_trace=True, want_draws=True)[0]
    z = o["rec_z"].reshape(-1)
    u = (o["rec_u"].reshape(-1).astype(np.float64) + 0.5) / 2.0 ** 32
    n = z.shape[0]
    assert abs(z.mean()) < 5 / np.sqrt(n) and abs(z.var() - 1.0) < 5 * np.sqrt(2.0 / n)
    assert abs(np.mean(z ** 4) - 3.0) < 5 * np.sqrt(96.0 / n)
    assert stats.kstest(z, "norm").pvalue > 1e-4
    assert stats.kstest(u, "uniform").pvalue > 1e-4
    # no serial structure between a child's word and its normal, or along the particle axis
    assert abs(np.corrcoef(u, z)[0, 1]) < 5 / np.sqrt(n)
    assert abs(np.corrcoef(z[:-1], z[1:])[0, 1]) < 5 / np.sqrt(n)


def _theta_grid():
    """(model, tag, theta) of tests/golden/theta_grid.npz -- the grid the reference fixtures were generated on."""
    import json
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "theta_grid.npz"))
    seen, out = set(), []
    for m in json.loads(str(z["meta"])):
        if (m["model"], m["tag"]) not in seen:
            seen.add((m["model"], m["tag"]))
            out.append((m["model"], m["tag"], z[m["key"] + "/theta"], z[m["key"] + "/y"] if not m["traced"] else None, m))
    # observations: the N = 1000 window of the same (model, tag) (generated by the reference from that theta)
    ys = {}
    for m in json.loads(str(z["meta"])):
        if not m["traced"]:
            ys[(m["model"], m["tag"])] = (z[m["key"] + "/y"], m["prior_mean"], m["prior_var"])
    return [(model, tag, theta) + ys[(model, tag)] for model, tag, theta, _, _ in out]


GRID_VARIANTS = {
    # model: (kernel, N, forced variant, (NT, PPT, cdf))
    "svm": [("prior", 1000, "wg256x4s", (256, 4, "fixed32")), ("prior", 100, "wg64x2s", (64, 2, "fixed32")),
            ("prior", 4000, "wg1024x4s", (1024, 4, "fixed32")), ("prior", 10000, "big16384", (16384, 1, "f64_uniform"))],
    "lgssm": [("optimal", 100, "wg64x2s", (64, 2, "fixed32")), ("prior", 100, "wg64x2s", (64, 2, "fixed32")),
              ("optimal", 1000, "wg256x4s", (256, 4, "fixed32")), ("optimal", 10000, "big16384", (16384, 1, "f64_uniform"))],
    "garch": [("optimal", 1000, "wg512x2s", (512, 2, "fixed32")), ("prior", 1000, "wg512x2s", (512, 2, "fixed32")),
              ("optimal", 100, "wg64x2s", (64, 2, "fixed32")), ("optimal", 10000, "big16384", (16384, 1, "f64_uniform"))],
}


@pytest.mark.parametrize("model,tag,theta,y,pm,pv", _theta_grid(), ids=lambda v: v if isinstance(v, str) else "")
def test_theta_grid_device_kernels_replayed(ctx, monkeypatch, model, tag, theta, y, pm, pv):
    """The timed instantiations over the reference's parameter grid (theta_grid.npz: C != 1, |A| = 0.9999, Cholesky
    factors 0.1 / 10, phi = 0.999, lambduh 0.01 / 0.99): recorded draws replayed by the oracle, S = 16 / B = 4 windows
    with importance weights, on wg256x4s / wg512x2s / wg64x2s / wg1024x4s / big16384."""
    T, t1, tL = y.shape[0], 4, 20
    weights = np.linspace(40.0, 61.0, tL - t1)
    for kernel, N, variant, (NT, PPT, cdf) in GRID_VARIANTS[model]:
        monkeypatch.setenv("PFGRAD_VARIANT", variant)
        q = dict(model=model, kernel=kernel, smoother="nemeth", stat="score", dtype="f64", rng="device", N=N, t1=t1, tL=tL,
                 lambduh=1.0, prior_mean=pm, prior_var=pv, y=y, weights=weights, theta=theta, seed=4242 + N, stream=len(tag))
        o = ctx.run_batch([q], want_trace=True, want_draws=True)[0]
        assert ctx.last_variant() == variant
        words = o["rec_ud"] if cdf == "f64_uniform" else o["rec_u"]
        with np.errstate(divide="ignore"):
            ref = po.pf_window(model, theta, y, N, o["rec_z0"], None, o["rec_z"], kernel=kernel, pf="poyiadjis_N", lambduh=1.0,
                               stat="score", t1=t1, tL=tL, weights=weights, prior_mean=pm, prior_var=pv, save_all=True,
                               resampler=lambda t, logw: po.device_ancestors(logw, words[t], NT, PPT, cdf))
        where = (model, tag, kernel, variant)
        assert int(np.sum(o["all_ancestors"] != ref["all_ancestors"])) == 0, where
        np.testing.assert_allclose(o["all_x_t"], ref["all_x_t"], rtol=RTOL, atol=ATOL, err_msg=str(where))
        np.testing.assert_allclose(o["all_log_weights"], ref["all_log_weights"], rtol=RTOL, atol=ATOL, err_msg=str(where))
        scale = max(1.0, np.abs(ref["all_statistics"]).max())
        np.testing.assert_allclose(o["all_statistics"], ref["all_statistics"], rtol=RTOL, atol=1e-8 * scale, err_msg=str(where))
        np.testing.assert_allclose(o["mean_stat"], ref["mean_statistic"], rtol=RTOL, atol=1e-8 * scale, err_msg=str(where))
        # log-likelihood: the reference's log(mean(exp(logw))) is not max-stabilised (buffered_smoother.py:124-126); with
        # Rinv = 100 the log-weights sit near -1e3 and exp() lands in the denormal range (a few significant bits) or at
        # zero, so the reference's OWN value is inexact there.  The kernel's is the max-stabilised form of the same sum:
        # compare with that, computed from the oracle's log-weights, and with the reference form wherever it is exact.
        lw = ref["all_log_weights"][1:]
        mx = lw.max(axis=1)
        steps = np.arange(T)
        inside = (steps >= t1) & (steps < tL)
        wt = np.where(inside, np.concatenate([np.zeros(t1), weights, np.zeros(T - tL)]), 0.0)
        stable = float(np.sum(wt * (mx + np.log(np.mean(np.exp(lw - mx[:, None]), axis=1)))))
        np.testing.assert_allclose(o["loglik"], stable, rtol=RTOL, atol=ATOL, err_msg=str(where))
        if np.all(mx[inside] > -600.0):
            np.testing.assert_allclose(o["loglik"], ref["loglikelihood_estimate"], rtol=RTOL, atol=ATOL, err_msg=str(where))
        assert np.isfinite(o["loglik"])
        plain = ctx.run_batch([dict(q)])[0]
        _assert_twin_statistics(plain["mean_stat"], o["mean_stat"], ctx.last_variant(), where)


@pytest.mark.parametrize("N,variant,NT", [(10000, "big16384", 16384), (4000, "big4096", 4096), (16384, "big16384", 16384)])
def test_sorted_uniforms_of_the_large_n_kernel_are_uniform_order_statistics(ctx, monkeypatch, N, variant, NT):
    """The large-N kernel resamples with SORTED uniforms built from exponential spacings and per-(chunk, wave)
    offsets; the replay tests feed the recorded values back to the oracle, so a wrong offset or total would pass
    them.  Here the recorded uniforms are tested for what they must be: per step the order statistics of N i.i.d.
    U(0,1) -- pooled over steps a KS test against U(0,1), the normalised spacings (N+1)(U_(r) - U_(r-1)), incl. both
    ends, against Exp(1), no correlation between neighbouring spacings, and mean U_(r) = r/(N+1) along the ranks."""
    from scipy import stats
    monkeypatch.setenv("PFGRAD_VARIANT", "big")            # the large-N kernel also where an LDS-resident variant would fit
    T = 24
    y = _series("svm", T, seed=N)
    q = dict(model="svm", kernel="prior", smoother="nemeth", stat="score", dtype="f64", rng="device", N=N, t1=0, tL=T,
             lambduh=1.0, prior_mean=0.0, prior_var=10.0, y=y, theta=THETA["svm"], seed=777 + N, stream=5)
    o = ctx.run_batch([q], want_trace=True, want_draws=True)[0]
    assert ctx.last_variant() == variant
    child = np.arange(N)
    rank = (child % NT) * 1 + child // NT                    # PPT = 1: child index = rank
    ud = o["rec_ud"][:, np.argsort(rank)]                    # [T, N] in rank order
    assert np.all(np.diff(ud, axis=1) >= 0.0) and np.all((ud > 0.0) & (ud < 1.0))
    pooled = ud.reshape(-1)
    assert stats.kstest(pooled, "uniform").pvalue > 1e-4
    ends = np.concatenate([np.zeros((T, 1)), ud, np.ones((T, 1))], axis=1)
    sp = np.diff(ends, axis=1) * (N + 1)                     # [T, N+1] normalised spacings
    flat = sp.reshape(-1)
    n = flat.size
    assert stats.kstest(flat, "expon").pvalue > 1e-4
    assert abs(flat.mean() - 1.0) < 1e-9                     # spacings of a step sum to one
    assert abs(flat.var() - 1.0) < 6 * np.sqrt(8.0 / n)      # Var of Exp(1) = 1 (fourth central moment 9)
    assert abs(np.corrcoef(sp[:, :-1].reshape(-1), sp[:, 1:].reshape(-1))[0, 1]) < 5 / np.sqrt(n)
    # E[U_(r)] = r / (N + 1): the offsets between chunks and waves are right along the whole range
    dev = ud.mean(axis=0) - np.arange(1, N + 1) / (N + 1.0)
    sd = np.sqrt(np.arange(1, N + 1) * (N - np.arange(1, N + 1) + 1.0) / ((N + 1.0) ** 2 * (N + 2.0)) / T)
    assert np.max(np.abs(dev) / sd) < 5.5


@pytest.mark.parametrize("N", [4000, 4096, 1500])
def test_sorted_words_of_the_1024_thread_variant_are_uniform_order_statistics(ctx, monkeypatch, N):
    """wg1024x4s (1024 < N <= 4096, LDS-resident) also resamples with order statistics (exponential spacings scanned
    across the workgroup, child rank = thread-major position).  The recorded 32-bit words, taken in rank order over
    the VALID children, must be non-decreasing and distributed as the order statistics of N i.i.d. uniforms: pooled
    KS against U(0,1), normalised spacings against Exp(1), mean of U_(r) = r / (N + 1)."""
    from scipy import stats
    monkeypatch.setenv("PFGRAD_VARIANT", "wg1024x4s")
    T = 24
    y = _series("svm", T, seed=N)
    q = dict(model="svm", kernel="prior", smoother="nemeth", stat="score", dtype="f64", rng="device", N=N, t1=0, tL=T,
             lambduh=1.0, prior_mean=0.0, prior_var=10.0, y=y, theta=THETA["svm"], seed=31 + N, stream=2)
    o = ctx.run_batch([q], want_trace=True, want_draws=True)[0]
    assert ctx.last_variant() == "wg1024x4s"
    child = np.arange(N)
    rank = (child % 1024) * 4 + child // 1024
    ud = (o["rec_u"][:, np.argsort(rank)].astype(np.float64) + 0.5) / 2.0 ** 32
    assert np.all(np.diff(ud, axis=1) >= 0.0)
    assert stats.kstest(ud.reshape(-1), "uniform").pvalue > 1e-4
    ends = np.concatenate([np.zeros((T, 1)), ud, np.ones((T, 1))], axis=1)
    sp = np.diff(ends, axis=1) * (N + 1)
    n = sp.size
    assert stats.kstest(sp.reshape(-1), "expon").pvalue > 1e-4
    assert abs(sp.var() - 1.0) < 6 * np.sqrt(8.0 / n)
    assert abs(np.corrcoef(sp[:, :-1].reshape(-1), sp[:, 1:].reshape(-1))[0, 1]) < 5 / np.sqrt(n)
    r = np.arange(1, N + 1)
    dev = ud.mean(axis=0) - r / (N + 1.0)
    sd = np.sqrt(r * (N - r + 1.0) / ((N + 1.0) ** 2 * (N + 2.0)) / T)
    assert np.max(np.abs(dev) / sd) < 5.5
