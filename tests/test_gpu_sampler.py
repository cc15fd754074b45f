"""GPU: the Python drop-in API (Helper / Sampler / SeqSampler) on the real HIP backend against
the reference's golden trajectories.  Same checks as tests/test_host_logic.py runs against the
oracle, now through libpfgrad.so; tolerance 1e-8 relative (fp64 REPLAY: exp/log rounding and
parallel-sum order only; the bar in BASELINE.json is gradient L2 error < 1e-4)."""
import numpy as np
import pytest

from test_host_logic import (_check_predictive, _check_sgrld, _check_sampler_case, _check_seq_and_minibatch, default_params,
                             vec)

pytestmark = pytest.mark.gpu
RTOL = 1e-8


def test_sampler_trajectories_match_reference_gpu(golden_sampler):
    for meta in golden_sampler.meta:
        _check_sampler_case(golden_sampler, meta, exact=False, rtol=RTOL)


@pytest.mark.parametrize("model", ["svm", "garch", "lgssm"])
def test_seq_sampler_and_minibatch_gpu(golden_sampler, model):
    _check_seq_and_minibatch(golden_sampler, model, exact=False, rtol=RTOL)


@pytest.mark.parametrize("model", ["svm", "garch", "lgssm"])
def test_predictive_loglikelihood_gpu(model):
    """pf_predictive_loglikelihood_estimate / predictive_loglikelihood(kind='pf') on the device
    (large-N kernel's predictive mode) vs the reference fixtures, fp64 REPLAY."""
    _check_predictive(model, exact=False, rtol=RTOL)


@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_predictive_device_rng_gpu(dtype):
    """Device-RNG predictive run: same estimator, different random numbers -- agrees with the
    replayed run within Monte-Carlo error (N = 4000 particles; also covers N > 1024)."""
    from sgmcmc_ssm_amd.models.svm import SVMSampler, generate_svm_data
    np.random.seed(8)
    p = default_params("svm")
    y = generate_svm_data(T=60, parameters=p)["observations"]
    sampler = SVMSampler(n=1, m=1, observations=y, parameters=p)
    np.random.seed(1)
    ref = sampler.predictive_loglikelihood(kind="pf", num_steps_ahead=3, N=4000)
    np.random.seed(2)
    got = sampler.predictive_loglikelihood(kind="pf", num_steps_ahead=3, N=4000, rng="device", dtype=dtype)
    assert got.shape == (4,) and np.all(np.isfinite(got))
    np.testing.assert_allclose(got, ref, atol=1.5, rtol=0.02)


def test_sgrld_gpu():
    """SGRLD / SGRD trajectories (LGSSM preconditioner) with the gradients from the HIP kernels."""
    _check_sgrld(exact=False, rtol=RTOL)


def test_helper_known_answer_gpu(golden_window):
    from sgmcmc_ssm_amd.models.svm import SVMHelper
    m = golden_window.meta[0]
    p = default_params("svm")
    fm = dict(log_constant=0.0, mean_precision=np.zeros(1), precision=np.eye(1) / m["prior_var"])
    helper = SVMHelper(forward_message=fm, **p.dim)
    y = golden_window.get(m["key"], "y").reshape(-1, 1)
    np.random.seed(99)
    g = helper.pf_gradient_estimate(observations=y, parameters=p, N=1000)
    ref = golden_window.get(m["key"], "mean_statistic")
    got = np.array([g["LRinv_vec"], g["LQinv_vec"], g["A"]])
    assert np.linalg.norm(got - ref) < 1e-4           # BASELINE.json bar
    np.testing.assert_allclose(got, ref, rtol=1e-9)
    np.random.seed(99)
    ll = helper.pf_loglikelihood_estimate(observations=y, parameters=p, N=1000)
    assert abs(ll - (-1561.7975224303605)) < 1e-8


def test_fit_timed_and_philox_mode():
    """fit_timed returns (parameters list, times); rng='philox' runs without host streams."""
    from sgmcmc_ssm_amd.models.svm import SVMSampler, generate_svm_data
    np.random.seed(3)
    p = default_params("svm")
    data = generate_svm_data(T=300, parameters=p)
    sampler = SVMSampler(n=1, m=1, observations=data["observations"], parameters=p.copy())
    plist, times = sampler.fit_timed(iter_type="SGLD", epsilon=0.05, subsequence_length=16, buffer_length=4,
                                     kind="pf", pf_kwargs=dict(pf="poyiadjis_N", N=500, rng="philox"),
                                     max_time=0.5, min_save_time=0.1)
    assert len(plist) == len(times) >= 2 and times[0] == 0.0
    assert all(np.all(np.isfinite(q.theta())) for q in plist)
    g1 = sampler.noisy_gradient(kind="pf", N=2000, rng="philox", dtype="f32", subsequence_length=-1)
    g2 = sampler.noisy_gradient(kind="pf", N=2000, rng="philox", dtype="f64", subsequence_length=-1)
    assert all(np.isfinite(vec("svm", g)).all() for g in (g1, g2))
