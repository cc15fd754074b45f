"""GPU: PaRIS smoother.  Chain of trust: the oracle in np.random mode is
bit-exact vs the reference (tests/test_oracle_golden.py::test_paris_oracle_bit_exact); the same
oracle code with uniforms addressed by (timestep, j, round, particle) is what the kernel is
compared with here, on identical pools (rtol 1e-9).  The device-RNG mode is checked statistically."""
import numpy as np
import pytest

from oracle import pf_oracle as po
from test_host_logic import default_params, GEN

pytestmark = pytest.mark.gpu
RTOL, ATOL = 1e-9, 1e-9


@pytest.fixture(scope="module")
def ctx():
    from sgmcmc_ssm_amd import _capi
    return _capi.default_context(0)


CASES = [("svm", "prior"), ("garch", "optimal"), ("garch", "prior"), ("lgssm", "optimal"), ("lgssm", "prior")]


@pytest.mark.parametrize("model,kernel", CASES)
@pytest.mark.parametrize("N,Ntilde,R", [(32, 2, 6), (100, 3, 2), (300, 1, 0), (257, 2, 3)])
def test_paris_pool_parity(ctx, model, kernel, N, Ntilde, R):
    rs = np.random.RandomState(N * 7 + Ntilde)
    T, t1, tL = 7, 1, 6
    p = default_params(model)
    np.random.seed(3)
    y = GEN[model](T=T, parameters=p)["observations"].reshape(-1)
    w = rs.uniform(1.0, 5.0, size=tL - t1)
    z0, u, z = po.draw_streams(rs, N, T)
    idx_u = rs.random_sample((T, Ntilde, max(R, 1), N))[:, :, :R]
    acc_u = rs.random_sample((T, Ntilde, max(R, 1), N))[:, :, :R]
    man_u = rs.random_sample((T, Ntilde, N))
    pv = 1.3
    ref = po.pf_window(model, p.theta(), y, N, z0, u, z, kernel=kernel, pf="paris", stat="score", t1=t1, tL=tL,
                       weights=w, prior_mean=0.0, prior_var=pv, save_all=True, Ntilde=Ntilde,
                       max_accept_reject=R, manual_sample_threshold=0,
                       paris_draws=po.PoolDraws(idx_u, acc_u, man_u))
    q = dict(model=model, kernel=kernel, smoother="paris", stat="score", dtype="f64", rng="replay", N=N, t1=t1,
             tL=tL, prior_mean=0.0, prior_var=pv, y=y, weights=w, theta=p.theta(), z0=z0, u=u, z=z,
             Ntilde=Ntilde, max_accept_reject=R, paris_idx_u=np.ascontiguousarray(idx_u),
             paris_acc_u=np.ascontiguousarray(acc_u), paris_man_u=man_u)
    o = ctx.run_batch([q], want_trace=True)[0]
    np.testing.assert_allclose(o["all_x_t"], ref["all_x_t"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(o["all_log_weights"], ref["all_log_weights"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(o["all_statistics"], ref["all_statistics"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(o["mean_stat"], ref["mean_statistic"], rtol=RTOL, atol=1e-8)
    assert abs(o["loglik"] - ref["loglikelihood_estimate"]) <= ATOL + RTOL * abs(ref["loglikelihood_estimate"])


@pytest.mark.parametrize("model,kernel", CASES)
@pytest.mark.parametrize("N,Ntilde,R", [(1100, 2, 4), (2500, 1, 0), (4099, 2, 2)])
def test_paris_large_n_pool_parity(ctx, model, kernel, N, Ntilde, R):
    """N > 1024: the large-N kernel's PaRIS instantiation (state in the HBM scratch, chunked
    exact fallback) against the oracle on identical uniform pools."""
    rs = np.random.RandomState(N + Ntilde)
    T, t1, tL = 4, 1, 4
    p = default_params(model)
    np.random.seed(5)
    y = GEN[model](T=T, parameters=p)["observations"].reshape(-1)
    w = rs.uniform(1.0, 5.0, size=tL - t1)
    z0, u, z = po.draw_streams(rs, N, T)
    idx_u = rs.random_sample((T, Ntilde, max(R, 1), N))[:, :, :R]
    acc_u = rs.random_sample((T, Ntilde, max(R, 1), N))[:, :, :R]
    man_u = rs.random_sample((T, Ntilde, N))
    pv = 1.3
    ref = po.pf_window(model, p.theta(), y, N, z0, u, z, kernel=kernel, pf="paris", stat="score", t1=t1, tL=tL,
                       weights=w, prior_mean=0.0, prior_var=pv, save_all=True, Ntilde=Ntilde,
                       max_accept_reject=R, manual_sample_threshold=0,
                       paris_draws=po.PoolDraws(idx_u, acc_u, man_u))
    q = dict(model=model, kernel=kernel, smoother="paris", stat="score", dtype="f64", rng="replay", N=N, t1=t1,
             tL=tL, prior_mean=0.0, prior_var=pv, y=y, weights=w, theta=p.theta(), z0=z0, u=u, z=z,
             Ntilde=Ntilde, max_accept_reject=R, paris_idx_u=np.ascontiguousarray(idx_u),
             paris_acc_u=np.ascontiguousarray(acc_u), paris_man_u=man_u)
    o = ctx.run_batch([q], want_trace=True)[0]
    np.testing.assert_allclose(o["all_x_t"], ref["all_x_t"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(o["all_log_weights"], ref["all_log_weights"], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(o["all_statistics"], ref["all_statistics"], rtol=RTOL, atol=1e-8)
    np.testing.assert_allclose(o["mean_stat"], ref["mean_statistic"], rtol=RTOL, atol=1e-8)
    assert abs(o["loglik"] - ref["loglikelihood_estimate"]) <= ATOL + RTOL * abs(ref["loglikelihood_estimate"])


@pytest.mark.parametrize("model", ["svm", "garch"])
def test_paris_device_rng_statistics(ctx, model):
    """Device-RNG PaRIS (Helper API, default settings) vs the reference-order oracle:
    means of score and log-likelihood over repeated runs agree within 5 standard errors."""
    from sgmcmc_ssm_amd.particle_filters import make_problem
    T, N, B, R = 30, 150, 256, 64
    p = default_params(model)
    np.random.seed(11)
    y = GEN[model](T=T, parameters=p)["observations"].reshape(-1)
    kernel = "prior" if model == "svm" else "optimal"
    pm, pv = (0.0, 10.0) if model == "svm" else (0.0, float(po.garch_prior_x(p.theta())[1][0]))
    probs = [make_problem(model, kernel, "paris", y, p.theta(), N, prior_mean=pm, prior_var=pv, seed=5, stream=b, rng="device")
             for b in range(B)]
    assert probs[0]["rng"] == "device" and probs[0]["max_accept_reject"] == 64
    outs = ctx.run_batch(probs)
    got = np.array([np.append(o["mean_stat"], o["loglik"]) for o in outs])
    rs = np.random.RandomState(1)
    ref = []
    for _ in range(R):
        r = po.pf_window_paris_rng(model, p.theta(), y, N, rng=rs, kernel=kernel, stat="score",
                                   prior_mean=pm, prior_var=pv)
        ref.append(np.append(r["mean_statistic"], r["loglikelihood_estimate"]))
    ref = np.array(ref)
    se = np.sqrt(got.var(axis=0) / B + ref.var(axis=0) / R)
    zscore = np.abs(got.mean(axis=0) - ref.mean(axis=0)) / se
    assert np.all(zscore < 5.0), (zscore, got.mean(axis=0), ref.mean(axis=0))


def test_paris_through_sampler_api():
    """The demo's calls: noisy_logjoint(kind='pf', pf='paris', N=...) and a paris gradient."""
    from sgmcmc_ssm_amd.models.svm import SVMSampler
    np.random.seed(2)
    p = default_params("svm")
    y = GEN["svm"](T=60, parameters=p)["observations"]
    sampler = SVMSampler(n=1, m=1, observations=y, parameters=p)
    np.random.seed(4)
    lj = sampler.noisy_logjoint(kind="pf", pf="paris", N=200, return_loglike=True)
    assert np.isfinite(lj["logjoint"]) and np.isfinite(lj["loglikelihood"])
    g = sampler.noisy_gradient(kind="pf", pf="paris", N=200, subsequence_length=16, buffer_length=4)
    assert all(np.all(np.isfinite(v)) for v in g.values())
    g = sampler.noisy_gradient(kind="pf", pf="paris", N=5000, subsequence_length=16, buffer_length=4)
    assert all(np.all(np.isfinite(v)) for v in g.values())          # large-N kernel, PaRIS instantiation
    with pytest.raises(NotImplementedError):
        sampler.noisy_gradient(kind="pf", pf="paris", N=20000)


def test_paris_f32_and_filter_stat(ctx):
    """PaRIS with f32 state runs and agrees statistically with f64; suff-stat statistic works."""
    from sgmcmc_ssm_amd.particle_filters import make_problem
    p = default_params("lgssm")
    np.random.seed(13)
    y = GEN["lgssm"](T=25, parameters=p)["observations"].reshape(-1)
    res = {}
    for dtype in ("f64", "f32"):
        probs = [make_problem("lgssm", "optimal", "paris", y, p.theta(), 200, prior_var=10.0, seed=3, stream=b,
                              dtype=dtype, stat="suff", rng="device") for b in range(256)]
        outs = ctx.run_batch(probs)
        res[dtype] = np.array([np.append(o["mean_stat"], o["loglik"]) for o in outs])
        assert np.all(np.isfinite(res[dtype])) and res[dtype].shape[1] == 4
    se = np.sqrt(res["f64"].var(axis=0) / 256 + res["f32"].var(axis=0) / 256)
    assert np.all(np.abs(res["f64"].mean(axis=0) - res["f32"].mean(axis=0)) / se < 5.0)


def test_paris_randomised_pool_parity(ctx):
    """Randomised sweep of the accept-reject machinery: tiny and ragged N (single pending child,
    K = 64 consecutive rounds per pass), round caps that are not multiples of K, Ntilde 1..3, all
    models / kernels -- device vs oracle on identical pools, bit-level ancestry (rtol 1e-9)."""
    rs = np.random.RandomState(2024)
    for trial in range(36):
        model, kernel = CASES[trial % len(CASES)]
        N = int(rs.choice([1, 2, 3, 5, 17, 33, 63, 64, 65, 130, 255, 256, 257, 400, 1000]))
        Ntilde = int(rs.randint(1, 4))
        R = int(rs.choice([0, 1, 3, 7, 13, 33, 70]))
        T = int(rs.randint(2, 6))
        t1 = int(rs.randint(0, T))
        tL = int(rs.randint(t1 + 1, T + 1))
        p = default_params(model)
        y = rs.normal(size=T) * (1.5 if model != "garch" else 0.7)
        w = rs.uniform(0.5, 3.0, size=tL - t1) if rs.rand() < 0.5 else None
        z0, u, z = po.draw_streams(rs, N, T)
        idx_u = rs.random_sample((T, Ntilde, max(R, 1), N))[:, :, :R]
        acc_u = rs.random_sample((T, Ntilde, max(R, 1), N))[:, :, :R]
        man_u = rs.random_sample((T, Ntilde, N))
        stat = "score" if rs.rand() < 0.7 else "suff"
        ref = po.pf_window(model, p.theta(), y, N, z0, u, z, kernel=kernel, pf="paris", stat=stat, t1=t1, tL=tL,
                           weights=w, prior_mean=0.1, prior_var=1.7, save_all=True, Ntilde=Ntilde,
                           max_accept_reject=R, manual_sample_threshold=0,
                           paris_draws=po.PoolDraws(idx_u, acc_u, man_u))
        q = dict(model=model, kernel=kernel, smoother="paris", stat=stat, dtype="f64", rng="replay", N=N, t1=t1,
                 tL=tL, prior_mean=0.1, prior_var=1.7, y=y, weights=w, theta=p.theta(), z0=z0, u=u, z=z,
                 Ntilde=Ntilde, max_accept_reject=R, paris_idx_u=np.ascontiguousarray(idx_u),
                 paris_acc_u=np.ascontiguousarray(acc_u), paris_man_u=man_u)
        o = ctx.run_batch([q], want_trace=True)[0]
        tag = str((trial, model, kernel, N, Ntilde, R, T, t1, tL, stat))
        np.testing.assert_allclose(o["all_x_t"], ref["all_x_t"], rtol=RTOL, atol=ATOL, err_msg=tag)
        np.testing.assert_allclose(o["all_statistics"], ref["all_statistics"], rtol=RTOL, atol=1e-8, err_msg=tag)
        np.testing.assert_allclose(o["mean_stat"], ref["mean_statistic"], rtol=RTOL, atol=1e-8, err_msg=tag)
        assert abs(o["loglik"] - ref["loglikelihood_estimate"]) <= ATOL + RTOL * abs(ref["loglikelihood_estimate"]), tag


# ----------------------------------------------------------------------------------------------------------------------
# Round 3: PaRIS seed for seed through the public API (the demos' "LD" sampler is SGLD on the PaRIS gradient,
# demo/exchange_rate/exchange_rate_demo_gbp.py:66, 99-135).  rng='replay' (the default) with N <= 1024 consumes np.random
# in the reference's order -- per timestep N uniforms, N normals, then the data-dependent run of uniforms of
# accept_reject_based_backward_sampling (pf.py:260-341) -- so np.random.seed(s) reproduces the reference's numbers AND
# leaves the generator where the reference leaves it.  Fixtures: tests/golden/paris_seed.npz (reference outputs).
# ----------------------------------------------------------------------------------------------------------------------
def _seed_cases(kind):
    from conftest import Golden
    g = Golden("paris_seed.npz")
    return [(g, m) for m in g.meta if m["kind"] == kind]


def _helper_for(g, m):
    from sgmcmc_ssm_amd.models.svm import SVMHelper
    from sgmcmc_ssm_amd.models.garch import GARCHHelper
    from sgmcmc_ssm_amd.models.lgssm import LGSSMHelper
    HELPERS = {"svm": SVMHelper, "garch": GARCHHelper, "lgssm": LGSSMHelper}
    key = m["key"]
    fm = None
    if m["has_forward_message"]:
        fm = dict(log_constant=0.0, mean_precision=g.get(key, "fm_mean_precision").copy(),
                  precision=g.get(key, "fm_precision").reshape(1, 1).copy())
    return HELPERS[m["model"]](n=1, m=1, forward_message=fm)


def _params_for(model, theta):
    from sgmcmc_ssm_amd.models.svm import SVMParameters
    from sgmcmc_ssm_amd.models.garch import GARCHParameters
    from sgmcmc_ssm_amd.models.lgssm import LGSSMParameters
    if model == "svm":
        return SVMParameters(A=np.eye(1) * theta[0], LQinv=np.eye(1) * theta[1], LRinv=np.eye(1) * theta[2])
    if model == "lgssm":
        return LGSSMParameters(A=np.eye(1) * theta[0], C=np.eye(1) * theta[1], LQinv=np.eye(1) * theta[2], LRinv=np.eye(1) * theta[3])
    return GARCHParameters(log_mu=theta[0], logit_phi=theta[1], logit_lambduh=theta[2], LRinv=np.eye(1) * theta[3])


@pytest.mark.parametrize("idx", range(11))
def test_paris_helper_seed_for_seed(idx):
    from test_host_logic import vec
    g, m = _seed_cases("helper")[idx]
    key = m["key"]
    helper = _helper_for(g, m)
    p = _params_for(m["model"], g.get(key, "theta"))
    kw = dict(observations=g.get(key, "y").reshape(-1, 1), parameters=p, subsequence_start=m["t1"], subsequence_end=m["tL"],
              weights=g.get(key, "weights"), pf="paris", N=m["N"], kernel=m["kernel"], **m["kwargs"])
    np.random.seed(m["seed"])
    grad = helper.pf_gradient_estimate(**kw)
    nxt = np.random.random_sample()
    ref = g.get(key, "grad")
    np.testing.assert_allclose(vec(m["model"], grad), ref, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(ref).max()), err_msg=str(m))
    assert nxt == float(g.get(key, "next_draw")), m          # the generator stands where the reference's stands
    np.random.seed(m["seed"])
    ll = helper.pf_loglikelihood_estimate(**kw)
    nxt = np.random.random_sample()
    assert abs(ll - float(g.get(key, "loglik"))) <= 1e-9 * abs(float(g.get(key, "loglik"))), m
    assert nxt == float(g.get(key, "next_draw_loglik")), m


@pytest.mark.parametrize("idx", range(2))
def test_paris_sampler_seed_for_seed(idx):
    """Sampler.noisy_gradient / sample_sgld + project_parameters with pf='paris' (S = 16, B = 4 windows): the
    reference's three-step trajectory after np.random.seed."""
    from test_host_logic import SAMPLERS, vec
    g, m = _seed_cases("sampler")[idx]
    key = m["key"]
    model = m["model"]
    y = g.get(key, "y").reshape(-1, 1)
    sampler = SAMPLERS[model][0](n=1, m=1, observations=y, parameters=_params_for(model, g.get(key, "theta0")))
    kwargs = dict(kind="pf", pf="paris", N=m["N"], minibatch_size=1, **m["kwargs"])
    np.random.seed(5150)
    grad = sampler.noisy_gradient(**kwargs)
    ref = g.get(key, "noisy_gradient")
    np.testing.assert_allclose(vec(model, grad), ref, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(ref).max()))
    np.random.seed(5151)
    traj = [sampler.parameters.theta().copy()]
    for _ in range(3):
        sampler.sample_sgld(epsilon=m["eps"], **kwargs)
        sampler.project_parameters()
        traj.append(sampler.parameters.theta().copy())
    np.testing.assert_allclose(np.array(traj), g.get(key, "sgld_traj"), rtol=1e-9, atol=1e-11)
    assert np.random.random_sample() == float(g.get(key, "next_draw"))


def test_paris_stream_too_short_is_reported_and_retried(ctx):
    """The kernel reports a stream that ran out (paris_consumed = -1) instead of reading past it; the host loop retries
    the timestep with a longer block and ends at the same numbers."""
    from sgmcmc_ssm_amd import particle_filters as pfm
    g, m = _seed_cases("helper")[1]
    key = m["key"]
    N = m["N"]
    rs = np.random.RandomState(1)
    x = rs.normal(size=(N, 1))
    q = dict(model="svm", kernel="prior", smoother="paris", stat="score", dtype="f64", rng="replay", N=N, t1=0, tL=1, lambduh=1.0,
             theta=g.get(key, "theta"), prior_mean=0.0, prior_var=1.0, y=np.array([0.3]), init_x=x, init_logw=np.zeros(N),
             init_stats=np.zeros((N, 3)), u=rs.random_sample((1, N)), z=rs.normal(size=(1, N)), Ntilde=2, max_accept_reject=30,
             paris_manual_threshold=5)
    short = ctx.run_batch([dict(q, paris_stream=rs.random_sample(N))], want_final=True)[0]
    assert short["paris_consumed"] == -1
    block = rs.random_sample(64 * N)
    ok = ctx.run_batch([dict(q, paris_stream=block)], want_final=True)[0]
    assert 2 * N < ok["paris_consumed"] < 64 * N
    again = ctx.run_batch([dict(q, paris_stream=block[:ok["paris_consumed"]])], want_final=True)[0]      # exactly enough
    assert again["paris_consumed"] == ok["paris_consumed"] and np.array_equal(again["statistics"], ok["statistics"])
    # the hint-driven host loop: force a tiny first block
    pfm._paris_block_hint[(m["N"], 3)] = 1
    helper = _helper_for(g, m)
    np.random.seed(m["seed"])
    grad = helper.pf_gradient_estimate(observations=g.get(key, "y").reshape(-1, 1), parameters=_params_for("svm", g.get(key, "theta")),
                                       subsequence_start=m["t1"], subsequence_end=m["tL"], weights=g.get(key, "weights"),
                                       pf="paris", N=m["N"], **m["kwargs"])
    from test_host_logic import vec
    np.testing.assert_allclose(vec("svm", grad), g.get(key, "grad"), rtol=1e-9)
    assert np.random.random_sample() == float(g.get(key, "next_draw"))


@pytest.mark.parametrize("model,kernel,N,T,pre", [("svm", "prior", 1000, 40, 0), ("svm", "prior", 257, 30, 1), ("garch", "optimal", 300, 25, 1),
                                                 ("lgssm", "optimal", 64, 50, 0), ("lgssm", "prior", 999, 12, 3), ("svm", "prior", 1, 9, 1)])
@pytest.mark.parametrize("accept_reject", [True, False])
def test_paris_whole_window_in_one_launch(ctx, monkeypatch, model, kernel, N, T, pre, accept_reject):
    """PFG_FLAG_PARIS_RAW_STREAM (N <= 1024): the kernel takes everything from one stream of doubles in np.random's order,
    NumPy's legacy Gaussians included (polar method, cached second variate -- `pre` odd leaves one pending on entry, odd N
    leaves one pending on exit).  Against the CPU oracle consuming the same RandomState (bit-exact to the reference,
    tests/test_oracle_golden.py) at rtol 1e-9, the generator afterwards bit for bit where the oracle's is (key, position,
    cached Gaussian), and against the one-launch-per-timestep path on the same seed."""
    from sgmcmc_ssm_amd import particle_filters as pfm
    from test_host_logic import default_params, GEN
    np.random.seed(4)
    p = default_params(model)
    y = GEN[model](T=T, parameters=p)["observations"].reshape(-1)
    w = np.random.uniform(0.5, 2.0, size=T - 5)
    kw = dict(t1=2, tL=T - 3, weights=w, prior_mean=0.1, prior_var=1.3)
    ref_rs, rs = np.random.RandomState(99), np.random.RandomState(99)
    for _ in range(pre):
        ref_rs.normal(); rs.normal(); ref_rs.random_sample(2); rs.random_sample(2)
    ref = po.pf_window_paris_rng(model, p.theta(), y, N, rng=ref_rs, kernel=kernel, stat="score", Ntilde=2,
                                 accept_reject=accept_reject, **kw)
    calls = []
    real = pfm._paris_raw_window
    monkeypatch.setattr(pfm, "_paris_raw_window", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    out = pfm.buffered_pf_wrapper("paris", model, kernel, y, p.theta(), N, random_state=rs, accept_reject=accept_reject, **kw)
    assert calls == [1]
    np.testing.assert_allclose(out["mean_statistic"], ref["mean_statistic"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(out["loglikelihood_estimate"], ref["loglikelihood_estimate"], rtol=1e-9)
    np.testing.assert_allclose(out["x_t"][:, 0], ref["x_t"][:, 0], rtol=1e-9, atol=1e-9)
    a, b = rs.get_state(), ref_rs.get_state()
    assert np.array_equal(a[1], b[1]) and a[2] == b[2] and a[3] == b[3] and a[4] == b[4]
    assert rs.normal() == ref_rs.normal() and rs.random_sample() == ref_rs.random_sample()
    # the per-timestep path (host-drawn u / z, libm normals) on the same seed: same numbers to rounding, same generator
    rs2 = np.random.RandomState(99)
    for _ in range(pre):
        rs2.normal(); rs2.random_sample(2)
    monkeypatch.setattr(pfm, "_paris_raw_window", pfm._paris_replay_window)
    step = pfm.buffered_pf_wrapper("paris", model, kernel, y, p.theta(), N, random_state=rs2, accept_reject=accept_reject, **kw)
    np.testing.assert_allclose(out["mean_statistic"], step["mean_statistic"], rtol=1e-10, atol=1e-10)
    c = rs2.get_state()
    assert np.array_equal(a[1], c[1]) and a[2:] == c[2:]


def test_paris_large_n_one_launch_equals_one_launch_per_timestep(monkeypatch):
    """Round 4: 1024 < N <= 16384 runs the whole PaRIS window in ONE launch of the large-N kernel (in-kernel legacy
    Gaussians, stream cursor carried across timesteps); one launch per timestep (PFGRAD_PARIS_PER_TIMESTEP=1: round 3's
    path, pinned seed for seed by paris_seed.npz) must give the same numbers and leave np.random at the same place -- with a
    cached Gaussian pending on entry, accept_reject=False, odd N (a cached variate pending after every call)."""
    from sgmcmc_ssm_amd import particle_filters as pfm
    from test_host_logic import default_params, GEN
    for model, kernel, N, T, kw, pre in (("svm", "prior", 1500, 6, dict(), 0), ("garch", "optimal", 2501, 5, dict(Ntilde=3), 1),
                                         ("lgssm", "optimal", 1025, 4, dict(accept_reject=False), 0),
                                         ("svm", "prior", 5000, 3, dict(max_accept_reject=5, manual_sample_threshold=100), 3)):
        np.random.seed(11)
        p = default_params(model)
        y = GEN[model](T=T, parameters=p)["observations"].reshape(-1)
        outs = []
        for per_step in (False, True):
            if per_step:
                monkeypatch.setenv("PFGRAD_PARIS_PER_TIMESTEP", "1")
            else:
                monkeypatch.delenv("PFGRAD_PARIS_PER_TIMESTEP", raising=False)
            rs = np.random.RandomState(321)
            for _ in range(pre):
                rs.normal()                       # an odd number of normals leaves a cached Gaussian behind
            o = pfm.buffered_pf_wrapper("paris", model, kernel, y, p.theta(), N, random_state=rs, t1=1, tL=T, prior_var=2.0,
                                        weights=1.0 + np.arange(T - 1), **kw)
            outs.append((o, rs.get_state()))
        (a, sa), (b, sb) = outs
        np.testing.assert_allclose(a["mean_statistic"], b["mean_statistic"], rtol=1e-10, atol=1e-10, err_msg=str((model, N)))
        np.testing.assert_allclose(a["x_t"], b["x_t"], rtol=1e-10, atol=1e-10)
        assert abs(a["loglikelihood_estimate"] - b["loglikelihood_estimate"]) <= 1e-10 * abs(b["loglikelihood_estimate"])
        assert np.array_equal(sa[1], sb[1]) and sa[2:4] == sb[2:4] and abs(sa[4] - sb[4]) <= 1e-12 * max(1.0, abs(sb[4])), (model, N)


def test_paris_one_launch_stream_too_short_and_refusals(ctx, monkeypatch):
    """A raw stream that runs out is reported (-1) and the host draws a longer one; the flag's preconditions are checked."""
    from sgmcmc_ssm_amd import _capi, particle_filters as pfm
    from test_host_logic import default_params, GEN
    np.random.seed(4)
    p = default_params("svm")
    T, N = 30, 500
    y = GEN["svm"](T=T, parameters=p)["observations"].reshape(-1)
    ref_rs = np.random.RandomState(7)
    ref = po.pf_window_paris_rng("svm", p.theta(), y, N, rng=ref_rs, kernel="prior", stat="score", Ntilde=2, t1=0, tL=T,
                                 prior_mean=0.0, prior_var=1.0)
    monkeypatch.setitem(pfm._paris_raw_hint, (N, 2, True), 10.0)           # a hopeless first guess: two retries
    runs = []
    real = _capi.Context.run_batch
    monkeypatch.setattr(_capi.Context, "run_batch", lambda self, *a, **k: (runs.append(1), real(self, *a, **k))[1])
    rs = np.random.RandomState(7)
    out = pfm.buffered_pf_wrapper("paris", "svm", "prior", y, p.theta(), N, random_state=rs, prior_mean=0.0, prior_var=1.0)
    assert len(runs) >= 2
    np.testing.assert_allclose(out["mean_statistic"], ref["mean_statistic"], rtol=1e-9, atol=1e-9)
    assert rs.random_sample() == ref_rs.random_sample()
    monkeypatch.undo()
    base = dict(model="svm", kernel="prior", smoother="paris", stat="score", dtype="f64", rng="replay", N=N, t1=0, tL=T, lambduh=1.0,
                theta=p.theta(), prior_mean=0.0, prior_var=1.0, y=y, Ntilde=2, max_accept_reject=30, paris_manual_threshold=5,
                paris_stream=np.random.random_sample(200), flags=_capi.FLAG_PARIS_RAW_STREAM)
    assert ctx.run_batch([dict(base)], want_final=True)[0]["paris_consumed"] == -1
    with pytest.raises(ValueError, match="must be NULL"):
        ctx.run_batch([dict(base, z0=np.zeros(N), u=np.zeros((T, N)), z=np.zeros((T, N)))])
    # round 4: the large-N kernel takes the whole-window stream too (f64): a stream that is too short reports -1 there as
    # well, an f32 window is refused (the stream is np.random's doubles)
    assert ctx.run_batch([dict(base, N=2000)], want_final=True)[0]["paris_consumed"] == -1
    with pytest.raises(NotImplementedError, match="dtype f64"):
        ctx.run_batch([dict(base, N=2000, dtype="f32")])
    with pytest.raises(ValueError, match="RAW_CARRY|replay streams"):          # the binding refuses it before the library does
        ctx.run_batch([dict(base, flags=_capi.FLAG_PARIS_RAW_CARRY)])
