"""Smoothed latent marginals (`pf_latent_var_distr` / predict(kind='pf'))."""
import numpy as np
import pytest

from oracle import pf_oracle as po
from conftest import Golden
from test_host_logic import default_params, SAMPLERS, GEN


@pytest.fixture(scope="module")
def golden_latent():
    return Golden("latent.npz")


def _kw(meta):
    kernel = meta["kernel"] or po.DEFAULT_KERNEL[meta["model"]]
    return dict(kernel=kernel, pf=meta["pf"], t1=meta["t1"], tL=meta["tL"],
                prior_mean=meta["prior_mean"], prior_var=meta["prior_var"])


def test_oracle_elementwise_matches_reference(golden_latent):
    """The oracle's elementwise run reproduces the reference's pf_latent_var_distr bit for bit for every
    smoother the reference dispatches: poyiadjis_N, nemeth, paris (np.random order) and poyiadjis_N2."""
    g = golden_latent
    assert len(g.meta) == 16 and {m["pf"] for m in g.meta} == {"poyiadjis_N", "nemeth", "paris", "poyiadjis_N2"}
    for m in g.meta:
        rng = np.random.RandomState(m["seed"])
        xm, xc = po.latent_var_distr(m["model"], g.get(m["key"], "theta"), g.get(m["key"], "y"), m["N"], rng=rng, **_kw(m))
        assert np.array_equal(xm, g.get(m["key"], "x_mean")) and np.array_equal(xc, g.get(m["key"], "x_cov")), m
        if m["model"] == "garch":
            rng = np.random.RandomState(m["seed"])
            xm, xc = po.latent_var_distr("garch", g.get(m["key"], "theta"), g.get(m["key"], "y"), m["N"], rng=rng,
                                         squared=True, **_kw(m))
            assert np.array_equal(xm, g.get(m["key"], "x_mean_sq")) and np.array_equal(xc, g.get(m["key"], "x_cov_sq"))


@pytest.mark.gpu
def test_latent_var_distr_gpu_matches_reference(golden_latent):
    """Helper.pf_latent_var_distr through the device elementwise pass (pfg_problem.elementwise) against
    the reference's own outputs on identical seeds: poyiadjis_N, nemeth, poyiadjis_N2 and paris (np.random's
    order: the window runs step by step to fix the data-dependent consumption, then once more in one launch on the
    same numbers with the elementwise pass) replay NumPy's stream and leave the generator where the reference does."""
    from sgmcmc_ssm_amd.models.svm import SVMHelper
    from sgmcmc_ssm_amd.models.garch import GARCHHelper
    from sgmcmc_ssm_amd.models.lgssm import LGSSMHelper
    helpers = dict(svm=SVMHelper, garch=GARCHHelper, lgssm=LGSSMHelper)
    g = golden_latent
    n = 0
    for m in g.meta:
        model = m["model"]
        p = default_params(model)
        fm = None if model == "garch" else dict(log_constant=0.0, mean_precision=np.zeros(1),
                                                precision=np.eye(1) / m["prior_var"])
        helper = helpers[model](n=1, m=1, forward_message=fm)
        np.random.seed(m["seed"])
        xm, xc = helper.pf_latent_var_distr(observations=g.get(m["key"], "y").reshape(-1, 1), parameters=p,
                                            subsequence_start=m["t1"], subsequence_end=m["tL"], pf=m["pf"],
                                            N=m["N"], kernel=m["kernel"])
        np.testing.assert_allclose(xm, g.get(m["key"], "x_mean"), rtol=1e-9, atol=1e-9, err_msg=str(m))
        np.testing.assert_allclose(xc, g.get(m["key"], "x_cov"), rtol=1e-8, atol=1e-9)
        if m["pf"] == "paris":
            # the generator stands where the reference's stands: the oracle (bit-exact to the reference, see
            # test_oracle_elementwise_matches_reference) draws the same next number
            rng = np.random.RandomState(m["seed"])
            po.latent_var_distr(model, g.get(m["key"], "theta"), g.get(m["key"], "y"), m["N"], rng=rng, **_kw(m))
            assert np.random.random_sample() == rng.random_sample(), m
        if model == "garch":
            np.random.seed(m["seed"])
            xm, xc = helper.pf_latent_var_distr(observations=g.get(m["key"], "y").reshape(-1, 1), parameters=p,
                                                subsequence_start=m["t1"], subsequence_end=m["tL"], N=m["N"],
                                                pf=m["pf"], kernel=m["kernel"], squared=True)
            np.testing.assert_allclose(xm, g.get(m["key"], "x_mean_sq"), rtol=1e-9, atol=1e-9)
        n += 1
    assert n == 16
    with pytest.raises(ValueError):
        helper.pf_latent_var_distr(observations=np.zeros((4, 1)), parameters=p, lag=0)
    with pytest.raises(ValueError):
        helper.pf_latent_var_distr(observations=np.zeros((4, 1)), parameters=p, pf="filter")


@pytest.mark.gpu
@pytest.mark.parametrize("model,kernel", [("svm", "prior"), ("garch", "optimal"), ("lgssm", "optimal"), ("lgssm", "prior")])
@pytest.mark.parametrize("N,lam", [(100, 1.0), (300, 0.9), (1500, 0.95), (1500, 1.0)])
def test_elementwise_pass_vs_oracle_nemeth(model, kernel, N, lam):
    """The device elementwise pass (per-particle [N, 3L] statistics AND their average) against the
    oracle's elementwise run on the same replayed streams, with importance weights, lambda < 1 and
    N > 1024 (large-N kernel's trace)."""
    from sgmcmc_ssm_amd import _capi
    rs = np.random.RandomState(N)
    T, t1, tL = 12, 2, 10
    p = default_params(model)
    np.random.seed(3)
    y = GEN[model](T=T, parameters=p)["observations"].reshape(-1)
    w = rs.uniform(1.0, 3.0, size=tL - t1)
    z0, u, z = po.draw_streams(rs, N, T)
    ref = po.pf_window(model, p.theta(), y, N, z0, u, z, kernel=kernel, pf="nemeth", lambduh=lam, stat="suff", t1=t1,
                       tL=tL, weights=w, prior_mean=0.0, prior_var=1.3, elementwise_statistic=True)
    q = dict(model=model, kernel=kernel, smoother="nemeth", stat="none", dtype="f64", rng="replay", N=N, t1=t1, tL=tL,
             lambduh=lam, prior_mean=0.0, prior_var=1.3, y=y, weights=w, theta=p.theta(), z0=z0, u=u, z=z)
    o = _capi.default_context(0).run_batch([q], want_final=True, want_elementwise=True)[0]
    np.testing.assert_allclose(o["ew_stats"], ref["statistics"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(o["ew_mean"], ref["mean_statistic"], rtol=1e-9, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("model,kernel", [("svm", "prior"), ("garch", "optimal"), ("lgssm", "optimal")])
@pytest.mark.parametrize("N,Ntilde,R", [(64, 2, 4), (300, 3, 2), (1100, 2, 3)])
def test_elementwise_pass_vs_oracle_paris(model, kernel, N, Ntilde, R):
    """pf='paris' (what the exchange-rate demos call predict(target='latent', kind='pf') with): chain of
    trust as for the PaRIS gradient -- the oracle in np.random order is bit-exact vs the reference
    (test_oracle_elementwise_matches_reference); the same oracle code on uniforms addressed by
    (timestep, j, round, particle) is what the kernel + elementwise pass are compared with here."""
    from sgmcmc_ssm_amd import _capi
    rs = np.random.RandomState(N * 3 + Ntilde)
    T, t1, tL = 8, 1, 7
    p = default_params(model)
    np.random.seed(4)
    y = GEN[model](T=T, parameters=p)["observations"].reshape(-1)
    w = rs.uniform(1.0, 3.0, size=tL - t1)
    z0, u, z = po.draw_streams(rs, N, T)
    idx_u = rs.random_sample((T, Ntilde, R, N))
    acc_u = rs.random_sample((T, Ntilde, R, N))
    man_u = rs.random_sample((T, Ntilde, N))
    ref = po.pf_window(model, p.theta(), y, N, z0, u, z, kernel=kernel, pf="paris", stat="suff", t1=t1, tL=tL,
                       weights=w, prior_mean=0.0, prior_var=1.3, Ntilde=Ntilde, max_accept_reject=R,
                       manual_sample_threshold=0, paris_draws=po.PoolDraws(idx_u, acc_u, man_u),
                       elementwise_statistic=True)
    q = dict(model=model, kernel=kernel, smoother="paris", stat="none", dtype="f64", rng="replay", N=N, t1=t1, tL=tL,
             prior_mean=0.0, prior_var=1.3, y=y, weights=w, theta=p.theta(), z0=z0, u=u, z=z, Ntilde=Ntilde,
             max_accept_reject=R, paris_idx_u=idx_u, paris_acc_u=acc_u, paris_man_u=man_u)
    o = _capi.default_context(0).run_batch([q], want_final=True, want_elementwise=True)[0]
    np.testing.assert_allclose(o["ew_stats"], ref["statistics"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(o["ew_mean"], ref["mean_statistic"], rtol=1e-9, atol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("model,N,T", [("svm", 200, 25), ("garch", 1500, 12), ("lgssm", 64, 30)])
def test_latent_var_distr_paris_no_accept_reject_seed_for_seed(model, N, T):
    """paris_smoother(accept_reject=False) (pf.py:226-236) with elementwise statistics, against the oracle on the
    same seed: in one launch the kernel addresses the stream as (t * N + i) * Ntilde + j.  N = 1500: large-N kernel."""
    from sgmcmc_ssm_amd.models.svm import SVMHelper
    from sgmcmc_ssm_amd.models.garch import GARCHHelper
    from sgmcmc_ssm_amd.models.lgssm import LGSSMHelper
    np.random.seed(3)
    p = default_params(model)
    y = GEN[model](T=T, parameters=p)["observations"]
    fm = None if model == "garch" else dict(log_constant=0.0, mean_precision=np.zeros(1), precision=np.eye(1) / 1.7)
    helper = dict(svm=SVMHelper, garch=GARCHHelper, lgssm=LGSSMHelper)[model](n=1, m=1, forward_message=fm)
    np.random.seed(77)
    xm, xc = helper.pf_latent_var_distr(observations=y, parameters=p, pf="paris", N=N, accept_reject=False,
                                        subsequence_start=2, subsequence_end=T - 3)
    rng = np.random.RandomState(77)
    pm, pv, _ = helper._prior_x(fm, p)
    kw = dict(kernel=po.DEFAULT_KERNEL[model], pf="paris", t1=2, tL=T - 3, accept_reject=False,
              prior_mean=float(np.asarray(pm).reshape(-1)[0]), prior_var=float(np.asarray(pv).reshape(-1)[0]))
    rm, rc = po.latent_var_distr(model, p.theta(), y.reshape(-1), N, rng=rng, **kw)
    np.testing.assert_allclose(xm, rm, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(xc, rc, rtol=1e-8, atol=1e-9)
    assert np.random.random_sample() == rng.random_sample()


@pytest.mark.gpu
def test_latent_var_distr_paris_statistics(golden_latent):
    """pf='paris' on the device generator (rng='device'): its smoothed means agree with the reference's
    (different random numbers) within Monte-Carlo error, and with the O(N) smoother's."""
    from sgmcmc_ssm_amd.models.svm import SVMHelper
    g = golden_latent
    m = [q for q in g.meta if q["pf"] == "paris" and q["model"] == "svm"][0]
    p = default_params("svm")
    fm = dict(log_constant=0.0, mean_precision=np.zeros(1), precision=np.eye(1) / m["prior_var"])
    helper = SVMHelper(n=1, m=1, forward_message=fm)
    y = g.get(m["key"], "y").reshape(-1, 1)
    np.random.seed(1)
    xm, xc = helper.pf_latent_var_distr(observations=y, parameters=p, pf="paris", N=4000, rng="device")
    np.random.seed(2)
    xm2, xc2 = helper.pf_latent_var_distr(observations=y, parameters=p, pf="poyiadjis_N", N=4000)
    assert xm.shape == (y.shape[0], 1) and np.all(xc > 0)
    assert np.sqrt(np.mean((xm - xm2) ** 2)) < 0.15              # posterior sd ~1
    assert np.sqrt(np.mean((xm - g.get(m["key"], "x_mean")) ** 2)) < 0.35      # reference run: N = 120 particles


@pytest.mark.gpu
def test_predict_latent_gpu():
    """sampler.predict(target='latent', kind='pf') for a plain and a Seq sampler, incl. N > 1024
    (large-N kernel) and the device RNG; smoothed means track the simulated latent path."""
    np.random.seed(21)
    p = default_params("lgssm")
    data = GEN["lgssm"](T=80, parameters=p)
    y, x_true = data["observations"], data["latent_vars"]
    Sampler, SeqSampler = SAMPLERS["lgssm"]
    s = Sampler(n=1, m=1, observations=y, parameters=p)
    xm, xc = s.predict(target='latent', kind='pf', N=2000)
    assert xm.shape == (80, 1) and xc.shape == (80, 1, 1) and np.all(xc > 0)
    assert np.sqrt(np.mean((xm - x_true) ** 2)) < 1.0          # posterior sd ~0.6 for these parameters
    xm2, _ = s.predict(target='latent', kind='pf', N=2000, rng="device")
    assert np.sqrt(np.mean((xm2 - xm) ** 2)) < 0.25
    seq = SeqSampler(n=1, m=1, observations=[y[:30], y[30:]], parameters=p)
    res = seq.predict(target='latent', kind='pf', N=300)
    assert len(res) == 2 and res[0][0].shape == (30, 1) and res[1][0].shape == (50, 1)
    with pytest.raises(NotImplementedError):
        s.predict(target='y', kind='pf')
