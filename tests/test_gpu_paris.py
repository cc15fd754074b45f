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
    probs = [make_problem(model, kernel, "paris", y, p.theta(), N, prior_mean=pm, prior_var=pv, seed=5, stream=b)
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
                              dtype=dtype, stat="suff") for b in range(256)]
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
