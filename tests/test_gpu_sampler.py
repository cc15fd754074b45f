"""GPU: the Python drop-in API (Helper / Sampler / SeqSampler) on the real HIP backend against
the reference's golden trajectories.  Same checks as tests/test_host_logic.py runs against the
oracle, now through libpfgrad.so; tolerance 1e-8 relative (fp64 REPLAY: exp/log rounding and
parallel-sum order only; the bar in BASELINE.json is gradient L2 error < 1e-4)."""
import numpy as np
import pytest

from test_host_logic import (_check_eurus, _check_predictive, _check_sgrld, _check_sampler_case, _check_seq_and_minibatch,
                             default_params, eurus_segments, vec)

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


def test_eurus_seq_sampler_gpu():
    """BASELINE config 5 on its own data: SeqSVMSampler over the 49 EURUS segments, N = 10000, S = 16,
    B = 4 (REPLAY -> large-N kernel) against the reference's gradient / SGLD / fit trajectories."""
    _check_eurus(exact=False, rtol=RTOL)


def test_eurus_chain_ensemble_matches_seq_sampler():
    """The resident path of config 5: ChainEnsemble over the LIST of EURUS segments (device generator,
    pf_big_kernel<.., 16384>) estimates the same gradient as the reference-pinned SeqSVMSampler: mean
    over chains vs mean over replayed seeds, both at theta0, within Monte-Carlo error."""
    from sgmcmc_ssm_amd.ensemble import ChainEnsemble
    from sgmcmc_ssm_amd.models.svm import SVMParameters, SeqSVMSampler
    g, segs = eurus_segments()
    th = g["theta0"]
    p = SVMParameters(A=np.eye(1) * th[0], LQinv=np.eye(1) * th[1], LRinv=np.eye(1) * th[2])
    C = 2048
    ens = ChainEnsemble("svm", segs, p, num_chains=C, N=10000, epsilon=1e-9, subsequence_length=16, buffer_length=4,
                        seed=5)
    ens.launch_pf()
    ens.synchronize()
    assert ens.ctx.last_variant() == "big16384"
    s, _ = ens.last_gradient_statistics()               # columns [LRinv, LQinv, A], already x T_total / T_seq
    dev_mean, dev_se = s.mean(0), s.std(0) / np.sqrt(C)
    sampler = SeqSVMSampler(n=1, m=1, observations=segs, parameters=p)
    runs = 96
    ref = []
    for r in range(runs):
        np.random.seed(4000 + r)
        gr = sampler.noisy_gradient(kind="pf", pf="poyiadjis_N", N=10000, subsequence_length=16, buffer_length=4,
                                    num_sequences=1, is_scaled=False)
        pr = sampler.prior.grad_logprior(sampler.parameters)
        ref.append([float(np.asarray(gr[k]).reshape(-1)[0]) - float(np.asarray(pr[k]).reshape(-1)[0])
                    for k in ("LRinv_vec", "LQinv_vec", "A")])
    ref = np.array(ref, dtype=float).reshape(runs, 3)
    ref_mean, ref_se = ref.mean(0), ref.std(0) / np.sqrt(runs)
    assert np.all(np.abs(dev_mean - ref_mean) <= 5 * np.sqrt(dev_se ** 2 + ref_se ** 2)), (dev_mean, ref_mean, dev_se, ref_se)


def test_eurus_sghmc_extension():
    """BASELINE config 5 as it is worded (SGHMC with the Poyiadjis O(N) filter on the EURUS segments,
    N = 10000): SGHMC is an extension (the reference has no momentum sampler), so it is pinned only by
    construction -- friction 1 IS the SGLD trajectory, bit for bit -- and by staying finite and distinct."""
    from sgmcmc_ssm_amd.ensemble import ChainEnsemble
    from sgmcmc_ssm_amd.models.svm import SVMParameters
    g, segs = eurus_segments()
    th = g["theta0"]
    p = SVMParameters(A=np.eye(1) * th[0], LQinv=np.eye(1) * th[1], LRinv=np.eye(1) * th[2])

    def run(sampler, friction):
        e = ChainEnsemble("svm", segs, p, num_chains=64, N=10000, epsilon=0.001, subsequence_length=16, buffer_length=4,
                          seed=9, sampler=sampler, friction=friction)
        e.step(3)
        e.synchronize()
        return e.theta()
    sgld, hmc1, hmc = run("sgld", 1.0), run("sghmc", 1.0), run("sghmc", 0.1)
    np.testing.assert_array_equal(sgld, hmc1)
    assert np.all(np.isfinite(hmc)) and not np.array_equal(hmc, sgld)
    assert np.all(np.abs(hmc[:, 0]) <= 0.9999 + 1e-12) and np.all(hmc[:, 1:] > 0)      # projection applied


def test_device_generator_runs_are_a_function_of_np_random_state():
    """rng='device' through the Sampler API: seed AND stream of the device generator come from
    np.random, so np.random.seed(s) reproduces a run within one process (and across processes)."""
    from sgmcmc_ssm_amd.models.svm import SVMSampler, generate_svm_data
    np.random.seed(3)
    p = default_params("svm")
    y = generate_svm_data(T=100, parameters=p)["observations"]
    sampler = SVMSampler(n=1, m=1, observations=y, parameters=p)
    outs = []
    for _ in range(2):
        np.random.seed(21)
        outs.append(vec("svm", sampler.noisy_gradient(kind="pf", N=500, rng="device", subsequence_length=-1)))
    np.testing.assert_array_equal(outs[0], outs[1])
    np.random.seed(22)
    other = vec("svm", sampler.noisy_gradient(kind="pf", N=500, rng="device", subsequence_length=-1))
    assert not np.array_equal(outs[0], other)


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


# ---------------------------------------------------------------------------------------------------------------------
# resident fit (round 4): fit / fit_timed / fit_evaluate with iter_type='SGLD' and pf_kwargs rng='device' run a one-chain
# ChainEnsemble on the GPU (hipGraph replay of the step), the parameters copied back at the save points
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("model,S,B", [("svm", 16, 4), ("svm", -1, -1), ("garch", 16, 4), ("lgssm", 8, 2)])
def test_resident_fit_is_the_chain_ensemble_bitwise(model, S, B):
    """np.random.seed(s); sampler.fit('SGLD', pf_kwargs=dict(rng='device')) is (seed from np.random, chain 0) of a
    ChainEnsemble built by hand with the same arguments -- every iterate, bit for bit; the host loop
    (pf_kwargs resident=False) is still there and is a different (equally valid) trajectory."""
    from sgmcmc_ssm_amd.ensemble import ChainEnsemble
    from sgmcmc_ssm_amd.models.svm import SVMSampler, generate_svm_data
    from sgmcmc_ssm_amd.models.garch import GARCHSampler, generate_garch_data
    from sgmcmc_ssm_amd.models.lgssm import LGSSMSampler, generate_lgssm_data
    Sampler, gen = {"svm": (SVMSampler, generate_svm_data), "garch": (GARCHSampler, generate_garch_data),
                    "lgssm": (LGSSMSampler, generate_lgssm_data)}[model]
    np.random.seed(3)
    p = default_params(model)
    y = gen(T=200, parameters=p)["observations"]
    eps = 0.01 if model == "garch" else 0.05
    kw = dict(iter_type="SGLD", epsilon=eps, subsequence_length=S, buffer_length=B, kind="pf",
              pf_kwargs=dict(pf="poyiadjis_N", N=300, rng="device"))
    sampler = Sampler(n=1, m=1, observations=y, parameters=p.copy())
    assert sampler._resident_plan("SGLD", kw) is not None
    np.random.seed(77)
    hist = sampler.fit(num_iters=9, output_all=True, **kw)
    assert len(hist) == 10 and np.array_equal(hist[0].theta(), p.theta())
    assert np.array_equal(sampler.parameters.theta(), hist[-1].theta())
    # the same ensemble by hand: the seed is the two randint draws fit() takes from np.random
    np.random.seed(77)
    seed = int(np.random.randint(0, 2 ** 31 - 1)) | (int(np.random.randint(0, 2 ** 31 - 1)) << 31)
    ens = ChainEnsemble(model, y, p.copy(), num_chains=1, N=300, pf="poyiadjis_N", epsilon=eps, prior=sampler.prior,
                        subsequence_length=S, buffer_length=B, seed=seed, chain_offset=0,
                        forward_message=getattr(sampler, "forward_message", None), window_sampling="device")
    ref = ens.run(9, thin=1)
    got = np.array([h.theta() for h in hist[1:]])
    assert np.array_equal(got, ref[:, 0, :got.shape[1]]), (got, ref)
    assert np.all(np.isfinite(got))
    # final-only fit: one graph replay of nine steps -> the ninth iterate
    sampler2 = Sampler(n=1, m=1, observations=y, parameters=p.copy())
    np.random.seed(77)
    last = sampler2.fit(num_iters=9, **kw)
    assert np.array_equal(last.theta(), hist[-1].theta())
    # the host loop is untouched by rng='replay' and can still be asked for
    kw_host = dict(kw, pf_kwargs=dict(kw["pf_kwargs"], resident=False))
    assert sampler._resident_plan("SGLD", kw_host) is None
    assert sampler._resident_plan("SGLD", dict(kw, pf_kwargs=dict(pf="poyiadjis_N", N=300))) is None          # rng='replay'
    assert sampler._resident_plan("SGD", kw) is None and sampler._resident_plan("SGLD", dict(kw, minibatch_size=2)) is None


def test_resident_fit_timed_and_seq_sampler():
    """fit_timed on the resident path: (parameters, times) at the reference's save points; a sequence list with
    num_sequences = 1 (the exchange-rate demos' call, exchange_rate_full_demo.py:96-103) runs resident too (host-side
    window sampling, no graph); num_sequences = -1 keeps the host loop."""
    from sgmcmc_ssm_amd.models.svm import SVMSampler, SeqSVMSampler, generate_svm_data
    np.random.seed(5)
    p = default_params("svm")
    y = generate_svm_data(T=400, parameters=p)["observations"]
    sampler = SVMSampler(n=1, m=1, observations=y, parameters=p.copy())
    plist, times = sampler.fit_timed(iter_type="SGLD", epsilon=0.05, subsequence_length=16, buffer_length=4, kind="pf",
                                     pf_kwargs=dict(pf="poyiadjis_N", N=500, rng="device"), max_time=0.6, min_save_time=0.1)
    assert len(plist) == len(times) >= 3 and times[0] == 0.0 and all(b > a for a, b in zip(times, times[1:]))
    assert all(np.all(np.isfinite(q.theta())) for q in plist)
    assert not np.array_equal(plist[-1].theta(), plist[0].theta())
    pl, tm, metrics = sampler.fit_evaluate(iter_type="SGLD", epsilon=0.05, subsequence_length=16, buffer_length=4, kind="pf",
                                           pf_kwargs=dict(pf="poyiadjis_N", N=500, rng="device"), max_time=0.3, min_save_time=0.1,
                                           metric_functions=lambda s: dict(metric="A", variable="A", value=float(s.parameters.A[0, 0])))
    assert list(pl["iteration"]) == list(tm["iteration"]) and pl["iteration"].iloc[-1] >= 64          # >= one burst per save
    assert len(metrics) == len(pl)
    segs = [y[k * 80:(k + 1) * 80] for k in range(5)]
    seq = SeqSVMSampler(n=1, m=1, observations=segs, parameters=p.copy())
    kw = dict(iter_type="SGLD", epsilon=0.01, subsequence_length=16, buffer_length=4, kind="pf",
              pf_kwargs=dict(pf="poyiadjis_N", N=500, rng="device"))
    assert seq._resident_plan("SGLD", dict(kw, num_sequences=1)) is not None
    assert seq._resident_plan("SGLD", dict(kw, num_sequences=-1)) is None
    np.random.seed(8)
    h = seq.fit(num_iters=5, output_all=True, num_sequences=1, **kw)
    assert len(h) == 6 and all(np.all(np.isfinite(q.theta())) for q in h)
