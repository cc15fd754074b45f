"""Smoothed latent marginals (`pf_latent_var_distr` / predict(kind='pf'))."""
import numpy as np
import pytest

from oracle import pf_oracle as po
from conftest import Golden
from test_host_logic import default_params, SAMPLERS, GEN
from sgmcmc_ssm_amd import particle_filters


@pytest.fixture(scope="module")
def golden_latent():
    return Golden("latent.npz")


def _kw(meta):
    kernel = meta["kernel"] or po.DEFAULT_KERNEL[meta["model"]]
    return dict(kernel=kernel, pf=meta["pf"], t1=meta["t1"], tL=meta["tL"],
                prior_mean=meta["prior_mean"], prior_var=meta["prior_var"])


def test_oracle_elementwise_matches_reference(golden_latent):
    g = golden_latent
    assert len(g.meta) == 9
    for m in g.meta:
        rng = np.random.RandomState(m["seed"])
        xm, xc = po.latent_var_distr(m["model"], g.get(m["key"], "theta"), g.get(m["key"], "y"), m["N"], rng=rng, **_kw(m))
        assert np.array_equal(xm, g.get(m["key"], "x_mean")) and np.array_equal(xc, g.get(m["key"], "x_cov")), m
        if m["model"] == "garch":
            rng = np.random.RandomState(m["seed"])
            xm, xc = po.latent_var_distr("garch", g.get(m["key"], "theta"), g.get(m["key"], "y"), m["N"], rng=rng,
                                         squared=True, **_kw(m))
            assert np.array_equal(xm, g.get(m["key"], "x_mean_sq")) and np.array_equal(xc, g.get(m["key"], "x_cov_sq"))


def test_lineage_tracing_equals_elementwise_statistics(golden_latent):
    """Host-side genealogy tracing (product code) on the ORACLE's traces reproduces the oracle's
    elementwise statistics bit for bit (same values, same final average)."""
    g = golden_latent
    for m in g.meta:
        N = m["N"]
        y = g.get(m["key"], "y")
        T = y.shape[0]
        tL = T if m["tL"] is None else m["tL"]
        streams = po.draw_streams(np.random.RandomState(m["seed"]), N, T)
        kw = _kw(m)
        if m["pf"] == "nemeth":
            # Nemeth shrinkage: the forward recursion replayed on the host over the recorded
            # trajectory (product code) vs the oracle's elementwise run, two lambdas, +- weights
            for lam, weights in ((None, None), (0.7, np.linspace(1.0, 2.0, tL - m["t1"]))):
                out = po.pf_window(m["model"], g.get(m["key"], "theta"), y, N, *streams, stat="suff",
                                   elementwise_statistic=True, save_all=True, weights=weights, lambduh=lam, **kw)
                stats, avg = particle_filters.nemeth_elementwise_statistics(
                    m["model"], np.array(out["all_x_t"]), np.array(out["all_ancestors"]),
                    np.array(out["all_log_weights"]), m["t1"], tL, 0.95 if lam is None else lam, weights)
                assert np.array_equal(stats, out["statistics"]), m
                assert np.array_equal(avg, out["mean_statistic"]), m
            continue
        out = po.pf_window(m["model"], g.get(m["key"], "theta"), y, N, *streams, stat="suff",
                           elementwise_statistic=True, save_all=True, **kw)
        w = np.linspace(1.0, 2.0, tL - m["t1"])
        for weights in (None, w):
            if weights is not None:
                out = po.pf_window(m["model"], g.get(m["key"], "theta"), y, N, *streams, stat="suff",
                                   elementwise_statistic=True, save_all=True, weights=weights, **kw)
            stats, avg = particle_filters.smoothed_sufficient_statistics(
                m["model"], out["all_x_t"], out["all_ancestors"], out["log_weights"], m["t1"], tL, weights)
            assert np.array_equal(stats, out["statistics"]), m
            assert np.array_equal(avg, out["mean_statistic"]), m


@pytest.mark.gpu
def test_latent_var_distr_gpu_matches_reference(golden_latent):
    from sgmcmc_ssm_amd.models.svm import SVMHelper
    from sgmcmc_ssm_amd.models.garch import GARCHHelper
    from sgmcmc_ssm_amd.models.lgssm import LGSSMHelper
    helpers = dict(svm=SVMHelper, garch=GARCHHelper, lgssm=LGSSMHelper)
    g = golden_latent
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
        if model == "garch":
            np.random.seed(m["seed"])
            xm, xc = helper.pf_latent_var_distr(observations=g.get(m["key"], "y").reshape(-1, 1), parameters=p,
                                                subsequence_start=m["t1"], subsequence_end=m["tL"], N=m["N"],
                                                pf=m["pf"], kernel=m["kernel"], squared=True)
            np.testing.assert_allclose(xm, g.get(m["key"], "x_mean_sq"), rtol=1e-9, atol=1e-9)
    with pytest.raises(ValueError):
        helper.pf_latent_var_distr(observations=np.zeros((4, 1)), parameters=p, lag=0)
    with pytest.raises(NotImplementedError):
        helper.pf_latent_var_distr(observations=np.zeros((4, 1)), parameters=p, pf="paris")


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
