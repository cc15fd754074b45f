"""Host-side logic (CPU): Parameters / Prior / projection / window weights / data generators
and the sampler orchestration, against fixtures produced by the reference."""
import numpy as np
import pytest

from sgmcmc_ssm_amd.models.svm import (SVMParameters, SVMPrior, SVMSampler, SeqSVMSampler,
                                       generate_svm_data)
from sgmcmc_ssm_amd.models.garch import (GARCHParameters, GARCHPrior, GARCHSampler, SeqGARCHSampler,
                                         generate_garch_data)
from sgmcmc_ssm_amd.models.lgssm import (LGSSMParameters, LGSSMPrior, LGSSMSampler, SeqLGSSMSampler,
                                         generate_lgssm_data)
from sgmcmc_ssm_amd import sgmcmc_sampler, particle_filters
from oracle_backend import run_windows_oracle

PARAMS = {"svm": SVMParameters, "garch": GARCHParameters, "lgssm": LGSSMParameters}
PRIORS = {"svm": SVMPrior, "garch": GARCHPrior, "lgssm": LGSSMPrior}
SAMPLERS = {"svm": (SVMSampler, SeqSVMSampler), "garch": (GARCHSampler, SeqGARCHSampler),
            "lgssm": (LGSSMSampler, SeqLGSSMSampler)}
GEN = {"svm": generate_svm_data, "garch": generate_garch_data, "lgssm": generate_lgssm_data}
DATA_SEED = {"svm": 12345, "garch": 222, "lgssm": 333}


def from_theta(model, th):
    if model == "svm":
        return SVMParameters(A=np.eye(1) * th[0], LQinv=np.eye(1) * th[1], LRinv=np.eye(1) * th[2])
    if model == "lgssm":
        return LGSSMParameters(A=np.eye(1) * th[0], C=np.eye(1) * th[1], LQinv=np.eye(1) * th[2],
                               LRinv=np.eye(1) * th[3])
    return GARCHParameters(log_mu=th[0], logit_phi=th[1], logit_lambduh=th[2], LRinv=np.eye(1) * th[3])


def vec(model, d):
    names = {"svm": ("A", "LQinv_vec", "LRinv_vec"), "lgssm": ("A", "C", "LQinv_vec", "LRinv_vec"),
             "garch": ("log_mu", "logit_phi", "logit_lambduh", "LRinv_vec")}[model]
    return np.array([float(np.asarray(d[k]).reshape(-1)[0]) for k in names])


def default_params(model):
    if model == "svm":
        return SVMParameters(A=np.eye(1) * 0.95, Q=np.eye(1) * 0.5, R=np.eye(1) * 0.5)
    if model == "lgssm":
        return LGSSMParameters(A=np.eye(1) * 0.9, C=np.eye(1) * 1.0, Q=np.eye(1) * 0.7, R=np.eye(1) * 1.0)
    lm, lp, ll = GARCHParameters.convert_alpha_beta_gamma(0.1, 0.8, 0.05)
    return GARCHParameters(log_mu=lm, logit_phi=lp, logit_lambduh=ll, LRinv=np.eye(1) * 0.3 ** -0.5)


# ---------------------------------------------------------------------------------------
def test_parameters_interface():
    p = default_params("svm")
    assert list(p.var_dict) == ["A", "LQinv_vec", "LRinv_vec"]
    assert p.dim == {"n": 1, "m": 1} and p.n == 1 and p.m == 1
    assert p.A.shape == (1, 1) and p.LQinv_vec.shape == (1,) and p.LQinv.shape == (1, 1)
    np.testing.assert_allclose(p.Q, [[0.5]], rtol=1e-15)
    np.testing.assert_allclose(p.Qinv, p.LQinv ** 2 + 1e-16)
    assert p.phi is p.var_dict["A"]
    q = p.copy()
    q.var_dict["A"] += 1.0
    assert p.A[0, 0] == 0.95
    v = p.as_vector()
    assert v.shape == (3,)
    d = SVMParameters.from_vector_to_dict(v, **p.dim)
    assert d["A"].shape == (1, 1) and d["LRinv_vec"].shape == (1,)
    r = p + {k: np.ones_like(x) for k, x in p.var_dict.items()}
    assert r.A[0, 0] == 1.95
    assert list(default_params("lgssm").var_dict) == ["A", "C", "LQinv_vec", "LRinv_vec"]
    g = default_params("garch")
    assert list(g.var_dict) == ["log_mu", "logit_phi", "logit_lambduh", "LRinv_vec"]
    np.testing.assert_allclose([g.alpha[0], g.beta[0], g.gamma[0]], [0.1, 0.8, 0.05], rtol=1e-12)
    np.testing.assert_allclose(g.theta(), [g.log_mu[0], g.logit_phi[0], g.logit_lambduh[0], g.LRinv[0, 0]])
    with pytest.raises(ValueError):
        SVMParameters(A=np.eye(1), LQinv=np.eye(1))


def test_theta_matches_golden(golden_sampler):
    for model in PARAMS:
        np.testing.assert_array_equal(default_params(model).theta(), golden_sampler[model + "/theta0"])


def test_data_generators_reproduce_reference(golden_sampler):
    for model in PARAMS:
        np.random.seed(DATA_SEED[model])
        data = GEN[model](T=200, parameters=default_params(model))
        assert data["observations"].shape == (200, 1)
        np.testing.assert_array_equal(data["observations"].reshape(-1), golden_sampler[model + "/y"])


def test_random_subsequence_and_weights(golden_host):
    for m in golden_host.meta["subseq"]:
        np.random.seed(m["seed"])
        s, e, w = sgmcmc_sampler.random_subsequence_and_weights(S=m["S"], T=m["T"])
        assert (s, e) == (m["start"], m["end"])
        np.testing.assert_array_equal(w, golden_host.get(m["key"], "weights"))
    np.random.seed(0)
    assert sgmcmc_sampler.random_subsequence_and_weights(S=16, T=300)[:2] == (172, 188)
    with pytest.raises(ValueError):
        sgmcmc_sampler.random_subsequence_and_weights(S=7, T=20, partition_style="strict")


def test_prior_grad_and_logprior(golden_host):
    for m in golden_host.meta["prior"]:
        model = m["model"]
        prior = PRIORS[model].generate_default_prior(var=m["var"], n=1, m=1)
        p = from_theta(model, golden_host.get(m["key"], "theta"))
        g = prior.grad_logprior(p)
        np.testing.assert_array_equal(vec(model, g), golden_host.get(m["key"], "grad"))
        assert prior.logprior(p) == float(golden_host.get(m["key"], "logprior"))


def test_project_parameters(golden_host):
    for m in golden_host.meta["project"]:
        p = from_theta(m["model"], golden_host.get(m["key"], "before"))
        out = p.project_parameters()
        assert out is p
        np.testing.assert_array_equal(p.theta(), golden_host.get(m["key"], "after"))
    p = default_params("svm")
    p.project_parameters(A=dict(fixed=np.eye(1) * 0.3), Q=dict(thresh=False))
    assert p.A[0, 0] == 0.3


def test_prior_sampling_runs():
    np.random.seed(1)
    for model in PARAMS:
        prior = PRIORS[model].generate_default_prior(var=1.0, n=1, m=1)
        p = prior.sample_prior()
        assert isinstance(p, PARAMS[model]) and np.all(np.isfinite(p.theta()))
        q = PRIORS[model].generate_prior(p, from_mean=True, var=2.0)
        assert np.all(np.isfinite(vec(model, q.grad_logprior(p))))


# ---------------------------------------------------------------------------------------
# sampler orchestration: oracle stands in for the GPU (monkeypatched), trajectories must be
# IDENTICAL to the reference's (same RNG order, same arithmetic on the host side)
# ---------------------------------------------------------------------------------------
@pytest.fixture
def oracle_backend(monkeypatch):
    monkeypatch.setattr(particle_filters, "run_windows", run_windows_oracle)


def _check_sampler_case(g, meta, exact=True, rtol=0.0):
    model, key = meta["model"], meta["key"]
    Sampler = SAMPLERS[model][0]
    y = g[model + "/y"].reshape(-1, 1)
    kwargs = dict(kind="pf", pf=meta["pf"], N=meta["N"], subsequence_length=meta["S"],
                  buffer_length=meta["B"], minibatch_size=1)
    cmp = (np.testing.assert_array_equal if exact else
           (lambda a, b: np.testing.assert_allclose(a, b, rtol=rtol, atol=rtol)))
    sampler = Sampler(n=1, m=1, observations=y, parameters=default_params(model))
    np.random.seed(meta["seed"])
    cmp(vec(model, sampler.noisy_gradient(**kwargs)), g.get(key, "noisy_gradient"))
    np.random.seed(meta["seed"])
    cmp(vec(model, sampler.noisy_gradient(is_scaled=False, **kwargs)), g.get(key, "noisy_gradient_unscaled"))
    np.random.seed(meta["seed"])
    cmp(np.float64(sampler.noisy_loglikelihood(**kwargs)), g.get(key, "noisy_loglikelihood"))
    np.random.seed(meta["seed"] + 1)
    traj = [sampler.parameters.theta()]
    for _ in range(meta["nsteps"]):
        sampler.sample_sgld(epsilon=meta["eps"], **kwargs)
        traj.append(sampler.parameters.theta())
        sampler.project_parameters()
        traj.append(sampler.parameters.theta())
    cmp(np.array(traj), g.get(key, "sgld_traj"))
    for it in ("SGD", "ADAGRAD", "SGLD"):
        sampler = Sampler(n=1, m=1, observations=y, parameters=default_params(model))
        np.random.seed(meta["seed"] + 2)
        plist = sampler.fit(iter_type=it, num_iters=3, output_all=True, epsilon=meta["eps"] * 0.1,
                            subsequence_length=meta["S"], buffer_length=meta["B"], kind="pf",
                            pf_kwargs=dict(pf=meta["pf"], N=meta["N"]))
        assert len(plist) == 4
        cmp(np.array([q.theta() for q in plist]), g.get(key, "fit_" + it))


def test_sampler_trajectories_match_reference(oracle_backend, golden_sampler):
    assert len(golden_sampler.meta) == 12
    for meta in golden_sampler.meta:
        _check_sampler_case(golden_sampler, meta, exact=True)


def _check_seq_and_minibatch(g, model, exact=True, rtol=0.0):
    Sampler, SeqSampler = SAMPLERS[model]
    cmp = (np.testing.assert_array_equal if exact else
           (lambda a, b: np.testing.assert_allclose(a, b, rtol=rtol, atol=rtol)))
    y = g[model + "/y"].reshape(-1, 1)
    eps = {"svm": 0.1, "garch": 0.01, "lgssm": 0.1}[model]
    sampler = Sampler(n=1, m=1, observations=y, parameters=default_params(model))
    np.random.seed(77)
    cmp(vec(model, sampler.noisy_gradient(kind="pf", pf="poyiadjis_N", N=100, subsequence_length=10,
                                         buffer_length=3, minibatch_size=2)), g[model + "/minibatch2"])
    seqs = [y[0:60], y[60:95], y[95:160], y[160:200]]
    for nseq in (1, -1):
        sampler = SeqSampler(n=1, m=1, observations=seqs, parameters=default_params(model))
        np.random.seed(99 + nseq)
        cmp(vec(model, sampler.noisy_gradient(kind="pf", pf="poyiadjis_N", N=150, subsequence_length=16,
                                             buffer_length=4, num_sequences=nseq)),
            g["{0}/seq_grad_{1}".format(model, nseq)])
        np.random.seed(199 + nseq)
        traj = [sampler.parameters.theta()]
        for _ in range(3):
            sampler.sample_sgld(epsilon=eps * 0.1, kind="pf", pf="poyiadjis_N", N=150,
                                subsequence_length=16, buffer_length=4, num_sequences=nseq)
            sampler.project_parameters()
            traj.append(sampler.parameters.theta())
        cmp(np.array(traj), g["{0}/seq_traj_{1}".format(model, nseq)])
        k = "{0}/seq_loglike_{1}".format(model, nseq)
        if k in g:
            np.random.seed(299 + nseq)
            cmp(np.float64(sampler.noisy_loglikelihood(kind="pf", pf="poyiadjis_N", N=150,
                                                       subsequence_length=16, buffer_length=4,
                                                       num_sequences=nseq)), g[k])


@pytest.mark.parametrize("model", ["svm", "garch", "lgssm"])
def test_seq_sampler_and_minibatch_match_reference(oracle_backend, golden_sampler, model):
    _check_seq_and_minibatch(golden_sampler, model, exact=True)


def test_replay_stream_prefetch_is_adopted_and_exact(oracle_backend, golden_sampler, monkeypatch):
    """The next step's replay stream is prefetched on a worker thread while the current window runs
    (particle_filters.speculation) and adopted only when np.random is bit for bit where the clone was:
    trajectories equal the reference's (same fixtures as above) AND the prefetch really is used; any other
    use of np.random between steps voids it without changing a single number."""
    g = golden_sampler
    spec = particle_filters.speculation
    monkeypatch.setattr(particle_filters, "_SPECULATE_MIN", 1000)      # the fixtures' windows are small
    for meta in [m for m in g.meta if m["model"] == "svm" and m["pf"] == "poyiadjis_N"]:
        y = g["svm/y"].reshape(-1, 1)
        kwargs = dict(kind="pf", pf="poyiadjis_N", N=meta["N"], subsequence_length=meta["S"],
                      buffer_length=meta["B"], minibatch_size=1)
        sampler = SVMSampler(n=1, m=1, observations=y, parameters=default_params("svm"))
        spec.cancel()                                   # a prefetch left over from an earlier loop
        a0, d0 = spec.adopted, spec.discarded
        np.random.seed(meta["seed"] + 1)
        traj = [sampler.parameters.theta()]
        for _ in range(meta["nsteps"]):
            sampler.sample_sgld(epsilon=meta["eps"], **kwargs)
            traj.append(sampler.parameters.theta())
            sampler.project_parameters()
            traj.append(sampler.parameters.theta())
        np.testing.assert_array_equal(np.array(traj), g[meta["key"] + "/sgld_traj"])
        assert spec.adopted - a0 == meta["nsteps"] - 1 and spec.discarded == d0
        # something else draws from np.random between two steps: the prefetch is void, the numbers are not
        spec.cancel()
        sampler = SVMSampler(n=1, m=1, observations=y, parameters=default_params("svm"))
        np.random.seed(5)
        sampler.sample_sgld(epsilon=meta["eps"], **kwargs)
        extra = np.random.normal()
        d1 = spec.discarded
        sampler.sample_sgld(epsilon=meta["eps"], **kwargs)
        got = sampler.parameters.theta()
        assert spec.discarded == d1 + 1
        spec.cancel()
        sampler = SVMSampler(n=1, m=1, observations=y, parameters=default_params("svm"))
        np.random.seed(5)
        sampler.sample_sgld(epsilon=meta["eps"], **kwargs)
        spec.cancel()                                   # no prefetch at all this time
        assert np.random.normal() == extra
        sampler.sample_sgld(epsilon=meta["eps"], **kwargs)
        np.testing.assert_array_equal(got, sampler.parameters.theta())
    spec.cancel()


def eurus_segments():
    """BASELINE config 5's data: the 49 gap-split EUR/USD hourly segments of the reference's demo
    (tests/golden/eurus.npz, data arrays + reference outputs; tests/golden/make_golden.py)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "eurus.npz"))
    bounds = np.concatenate([[0], np.cumsum(g["segment_lengths"])])
    segs = [g["segments"][bounds[k]:bounds[k + 1]].reshape(-1, 1) for k in range(len(bounds) - 1)]
    return g, segs


def _check_eurus(exact=True, rtol=0.0):
    """SeqSVMSampler on the EURUS segments with the demo's settings (save_svm_params.py:60-66:
    N = 10000, S = 16, B = 4, num_sequences = 1, epsilon = 0.001) against the reference's outputs."""
    g, segs = eurus_segments()
    assert len(segs) == 49 and sum(len(s) for s in segs) == 5907
    cmp = (np.testing.assert_array_equal if exact else
           (lambda a, b: np.testing.assert_allclose(a, b, rtol=rtol, atol=rtol)))
    th = g["theta0"]
    mk = lambda t: SVMParameters(A=np.eye(1) * t[0], LQinv=np.eye(1) * t[1], LRinv=np.eye(1) * t[2])
    pfkw = dict(kind="pf", pf="poyiadjis_N", N=10000, subsequence_length=16, buffer_length=4, num_sequences=1)
    sampler = SeqSVMSampler(n=1, m=1, observations=segs, parameters=mk(th))
    np.random.seed(7)
    cmp(vec("svm", sampler.noisy_gradient(**pfkw)), g["noisy_gradient"])
    np.random.seed(8)
    traj = [sampler.parameters.theta()]
    for _ in range(4):
        sampler.sample_sgld(epsilon=0.001, **pfkw)
        sampler.project_parameters()
        traj.append(sampler.parameters.theta())
    cmp(np.array(traj), g["sgld_traj"])
    sampler = SeqSVMSampler(n=1, m=1, observations=segs, parameters=mk(th))
    np.random.seed(9)
    plist = sampler.fit(iter_type="SGLD", num_iters=3, output_all=True, epsilon=0.001, subsequence_length=16,
                        num_sequences=1, buffer_length=4, kind="pf", pf_kwargs=dict(pf="poyiadjis_N", N=10000))
    cmp(np.array([q.theta() for q in plist]), g["fit_SGLD"])
    sampler5 = SeqSVMSampler(n=1, m=1, observations=segs[:5], parameters=mk(g["fit_SGLD"][0]))
    np.random.seed(10)
    cmp(vec("svm", sampler5.noisy_gradient(kind="pf", pf="poyiadjis_N", N=1000, subsequence_length=-1, buffer_length=0,
                                           num_sequences=-1)), g["grad_all5"])
    np.random.seed(11)
    cmp(np.float64(sampler5.noisy_loglikelihood(kind="pf", pf="poyiadjis_N", N=1000, subsequence_length=-1,
                                                buffer_length=0, num_sequences=-1)), g["loglike_all5"])


def test_eurus_seq_sampler_matches_reference(oracle_backend):
    _check_eurus(exact=True)


def _check_predictive(model, exact=True, rtol=0.0):
    """Helper.pf_predictive_loglikelihood_estimate + Sampler/SeqSampler.predictive_loglikelihood
    (kind='pf') against the reference fixtures (tests/golden/predictive.npz)."""
    import json
    from conftest import Golden
    g = Golden("predictive.npz")
    cmp = (np.testing.assert_array_equal if exact else
           (lambda a, b: np.testing.assert_allclose(a, b, rtol=rtol, atol=rtol)))
    Sampler, SeqSampler = SAMPLERS[model]
    n = 0
    for m in g.meta:
        if m["model"] != model:
            continue
        p = default_params(model)
        sampler = Sampler(n=1, m=1, observations=g.get(m["key"], "y").reshape(-1, 1), parameters=p)
        helper = sampler.message_helper
        fm = None
        if model != "garch":
            fm = dict(log_constant=0.0, mean_precision=np.zeros(1), precision=np.eye(1) / m["prior_var"])
        np.random.seed(m["seed"])
        pred = helper.pf_predictive_loglikelihood_estimate(
            observations=g.get(m["key"], "y").reshape(-1, 1), parameters=p, num_steps_ahead=m["K"],
            subsequence_start=m["t1"], subsequence_end=m["tL"], N=m["N"], kernel=m["kernel"],
            forward_message=fm)
        cmp(pred, g.get(m["key"], "pred"))
        n += 1
    assert n == 2
    y = g["samp_" + model + "/y"].reshape(-1, 1)
    sampler = Sampler(n=1, m=1, observations=y, parameters=default_params(model))
    np.random.seed(555)
    cmp(sampler.predictive_loglikelihood(kind="pf", num_steps_ahead=3, subsequence_length=20, buffer_length=4,
                                         minibatch_size=2, N=60), g["samp_" + model + "/windowed"])
    np.random.seed(556)
    cmp(sampler.predictive_loglikelihood(kind="pf", num_steps_ahead=4, num_samples=50),
        g["samp_" + model + "/full"])
    seq = SeqSampler(n=1, m=1, observations=[y[0:50], y[50:85], y[85:120]], parameters=default_params(model))
    np.random.seed(557)
    pl = seq.predictive_loglikelihood(kind="pf", num_steps_ahead=2, N=40)
    rec = [r for r in json.loads(str(g["sampler_meta"])) if r["model"] == model][0]
    if rec["seq"] == "ok":
        cmp(pl, g["samp_" + model + "/seq"])
    else:
        # the reference's SeqLGSSMSampler raises IndexError here (it re-checks one sequence as a
        # list of sequences, lgssm/sampler.py:61); ours runs -- same fix as noisy_loglikelihood
        assert "IndexError" in rec["seq"] and np.all(np.isfinite(pl)) and pl.shape == (3,)
    with pytest.raises(ValueError, match="filter"):
        helper.pf_predictive_loglikelihood_estimate(observations=y, parameters=default_params(model),
                                                    pf="poyiadjis_N")


@pytest.mark.parametrize("model", ["svm", "garch", "lgssm"])
def test_predictive_loglikelihood_matches_reference(oracle_backend, model):
    _check_predictive(model, exact=True)


def _check_sgrld(exact=True, rtol=0.0):
    """SGRLD / SGRD (LGSSM preconditioner) on particle-filter gradients vs the reference
    trajectories (tests/golden/sgrld.npz)."""
    import json
    from conftest import Golden
    from sgmcmc_ssm_amd.models.lgssm import LGSSMPreconditioner
    g = Golden("sgrld.npz")
    cmp = (np.testing.assert_array_equal if exact else
           (lambda a, b: np.testing.assert_allclose(a, b, rtol=rtol, atol=rtol)))
    Sampler, SeqSampler = SAMPLERS["lgssm"]
    y = g["y"].reshape(-1, 1)
    for ci, m in enumerate(g.meta):
        kw = dict(kind="pf", pf="poyiadjis_N", N=m["N"], subsequence_length=m["S"], buffer_length=m["B"],
                  minibatch_size=1)
        key = m["key"]
        sampler = Sampler(n=1, m=1, observations=y, parameters=default_params("lgssm"))
        pre = LGSSMPreconditioner()
        np.random.seed(700 + ci)
        cmp(vec("lgssm", sampler.noisy_gradient(preconditioner=pre, is_scaled=m["is_scaled"], **kw)),
            g[key + "/precond_gradient"])
        np.random.seed(710 + ci)
        traj = [sampler.parameters.theta()]
        for _ in range(4):
            sampler.sample_sgrld(epsilon=0.05, preconditioner=pre, is_scaled=m["is_scaled"], **kw)
            traj.append(sampler.parameters.theta())
            sampler.project_parameters()
            traj.append(sampler.parameters.theta())
        cmp(np.array(traj), g[key + "/sgrld_traj"])
        sampler = Sampler(n=1, m=1, observations=y, parameters=default_params("lgssm"))
        np.random.seed(720 + ci)
        traj = [sampler.parameters.theta()]
        for _ in range(3):
            sampler.step_precondition_sgd(epsilon=0.05, preconditioner=pre, is_scaled=m["is_scaled"], **kw)
            sampler.project_parameters()
            traj.append(sampler.parameters.theta())
        cmp(np.array(traj), g[key + "/sgrd_traj"])
        for it in ("SGRLD", "SGRD"):
            sampler = Sampler(n=1, m=1, observations=y, parameters=default_params("lgssm"))
            np.random.seed(730 + ci)
            plist = sampler.fit(iter_type=it, num_iters=3, output_all=True, epsilon=0.02,
                                subsequence_length=m["S"], buffer_length=m["B"], kind="pf",
                                pf_kwargs=dict(pf="poyiadjis_N", N=m["N"]))
            cmp(np.array([q.theta() for q in plist]), g[key + "/fit_" + it])
    seq = SeqSampler(n=1, m=1, observations=[y[0:60], y[60:95], y[95:150]], parameters=default_params("lgssm"))
    np.random.seed(741)
    plist = seq.fit(iter_type="SGRLD", num_iters=3, output_all=True, epsilon=0.02, subsequence_length=16,
                    buffer_length=4, kind="pf", num_sequences=2, pf_kwargs=dict(pf="poyiadjis_N", N=90))
    cmp(np.array([q.theta() for q in plist]), g["seq/fit_SGRLD"])
    # sample_sgld_cv: control variates on the same windows (the PF ignores `parameters=`, as there)
    from sgmcmc_ssm_amd.models.svm import SVMParameters
    ysv = g["cv/y"].reshape(-1, 1)
    sm = SAMPLERS["svm"][0](n=1, m=1, observations=ysv, parameters=default_params("svm"))
    center = SVMParameters(A=np.eye(1) * 0.9, Q=np.eye(1) * 0.6, R=np.eye(1) * 0.4)
    cgrad = dict(A=np.array([[0.3]]), LQinv_vec=np.array([-0.2]), LRinv_vec=np.array([0.1]))
    # the reference's own sample_sgld_cv dies with a TypeError (noisy_gradient hands `parameters` to
    # grad_logprior twice, sgmcmc_sampler.py:447-450); ours is the documented algorithm: check it
    # against its definition, composed from noisy_gradient calls on the same windows and draws
    errs = json.loads(str(g["errors"]))
    assert errs["sample_sgld_cv"] == "TypeError"
    kw = dict(kind="pf", pf="poyiadjis_N", N=120, subsequence_length=16, buffer_length=4, minibatch_size=2)
    np.random.seed(808)
    sm.sample_sgld_cv(epsilon=0.05, centering_parameters=center, centering_gradient=cgrad, **kw)
    got = sm.parameters.theta()
    sm2 = SAMPLERS["svm"][0](n=1, m=1, observations=ysv, parameters=default_params("svm"))
    np.random.seed(808)
    bds = [sm2._random_subsequence_and_buffers(buffer_length=4, subsequence_length=16, T=150) for _ in range(2)]
    cur = sm2.noisy_gradient(buffer_dicts=bds, **kw)
    cen = sm2.noisy_gradient(parameters=center, buffer_dicts=bds, **kw)
    noise = sm2._get_sgmcmc_noise(**kw)
    for var in sm2.parameters.var_dict:
        sm2.parameters.var_dict[var] += 0.05 * (cgrad[var] + cur[var] - cen[var]) + np.sqrt(0.1) * noise[var]
    cmp(got, sm2.parameters.theta())
    assert not np.allclose(vec("svm", cur), vec("svm", cen))     # separate draws, different prior term
    # SVM / GARCH have no default preconditioner in the reference either
    assert errs["svm"] == errs["garch"] == "NotImplementedError"
    for model in ("svm", "garch"):
        sm = SAMPLERS[model][0](n=1, m=1, observations=np.zeros((30, 1)), parameters=default_params(model))
        with pytest.raises(NotImplementedError):
            sm.fit(iter_type="SGRLD", num_iters=1, epsilon=0.1, subsequence_length=-1, buffer_length=-1, kind="pf")


def test_sgrld_matches_reference(oracle_backend):
    _check_sgrld(exact=True)


def test_helper_known_answer(oracle_backend, golden_window):
    """Helper.pf_gradient_estimate on the SVM T=1000 N=1000 case."""
    from sgmcmc_ssm_amd.models.svm import SVMHelper
    m = golden_window.meta[0]
    p = default_params("svm")
    fm = dict(log_constant=0.0, mean_precision=np.zeros(1), precision=np.eye(1) / m["prior_var"])
    helper = SVMHelper(forward_message=fm, **p.dim)
    np.random.seed(99)
    g = helper.pf_gradient_estimate(observations=golden_window.get(m["key"], "y").reshape(-1, 1),
                                    parameters=p, N=1000)
    np.testing.assert_allclose([g["LRinv_vec"], g["LQinv_vec"], g["A"]],
                               [5.62738493, 9.88344914, -123.63120925], rtol=0, atol=1e-6)


def test_unsupported_paths_raise():
    y = np.zeros((10, 1))
    s = SVMSampler(n=1, m=1, observations=y, parameters=default_params("svm"))
    with pytest.raises(NotImplementedError):
        s.noisy_gradient(kind="marginal")
    with pytest.raises(ValueError):
        s.noisy_gradient(kind="pf", pf="bogus", N=10)
    with pytest.raises(NotImplementedError):
        s.noisy_gradient(kind="pf", kernel="optimal", N=10)
    with pytest.raises(ValueError):
        s.sample_sgld(epsilon=0.1, preconditioner=object())
    with pytest.raises(NotImplementedError):
        SVMSampler(n=2, m=1)


def test_run_batch_restores_the_garbage_collector_on_a_refused_problem():
    """run_batch switches the collector off while it marshals a large batch; a problem it refuses (here: window weights
    shorter than the window) must not leave it off for the rest of the process."""
    import gc
    from sgmcmc_ssm_amd import _capi

    class NoLibrary(_capi.Context):              # the marshalling needs no GPU: no handle is created
        def __init__(self):
            self.lib, self.handle = None, None
    ctx = NoLibrary()
    good = dict(model="svm", kernel="prior", N=8, y=np.zeros(4), theta=[0.9, 1.0, 1.0], rng="device", seed=1, stream=0)
    bad = dict(good, t1=0, tL=4, weights=np.ones(2))
    assert gc.isenabled()
    with pytest.raises(ValueError, match="weights shorter"):
        ctx.run_batch([good] * 300 + [bad], want_final=True)
    assert gc.isenabled()


def test_stream_pool_is_shared_safely_between_threads():
    """The replay-stream buffer pool is used by the caller and by the prefetch worker thread: hammer it from several
    threads (take, return) -- no IndexError from a racing pop, no buffer handed to two takers at once."""
    import threading
    from sgmcmc_ssm_amd import particle_filters as pfm
    errors, seen = [], []

    def work():
        try:
            for _ in range(300):
                bufs = pfm._stream_buffers(16, 8)
                seen.append(id(bufs[0]))
                bufs[0][0, 0] = threading.get_ident()
                assert bufs[0][0, 0] == threading.get_ident()
                pfm._recycle_bufs(bufs)
        except Exception as e:        # noqa: BLE001
            errors.append(e)
    threads = [threading.Thread(target=work) for _ in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert len(pfm._stream_pool.get((16, 8), [])) <= 6
