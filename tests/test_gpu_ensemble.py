"""GPU: resident multi-chain SGLD engine (ChainEnsemble) -- Philox particle filter is
statistically equivalent to the oracle, the update kernel implements sample_sgld +
project_parameters, chains are independent and reproducible."""
import numpy as np
import pytest

from oracle import pf_oracle as po
from test_host_logic import default_params, GEN, PRIORS, vec

pytestmark = pytest.mark.gpu


def _series(model, T, seed=5):
    from test_host_logic import GEN
    np.random.seed(seed)
    return GEN[model](T=T, parameters=default_params(model))["observations"]


@pytest.mark.parametrize("model,dtype,N", [("svm", "f64", 200), ("svm", "f32", 200), ("garch", "f64", 200),
                                           ("lgssm", "f64", 200), ("lgssm", "f64", 100), ("svm", "f64", 100)])
def test_philox_gradient_statistically_matches_oracle(model, dtype, N):
    """Mean score / log-lik over 512 Philox chains vs mean over 96 oracle (MT19937) runs:
    |difference| < 5 standard errors for every component.  N = 200 runs wg256x1, N = 100 the
    one-wave variant wg64x2 (BASELINE config 1's particle count)."""
    from sgmcmc_ssm_amd.ensemble import ChainEnsemble
    T, C, R = 60, 512, 96
    y = _series(model, T)
    p = default_params(model)
    ens = ChainEnsemble(model, y, p, num_chains=C, N=N, epsilon=1e-3, dtype=dtype, seed=9)
    ens.launch_pf()
    ens.synchronize()
    g, ll = ens.last_gradient_statistics()
    rs = np.random.RandomState(4)
    kernel = ens.kernel
    if model == "garch":
        pm, pv = po.garch_prior_x(p.theta())
        pv = float(pv[0])
    else:
        pm, pv = 0.0, 10.0
    ref = []
    for _ in range(R):
        o = po.pf_window_rng(model, p.theta(), y, N, rng=rs, kernel=kernel, pf="poyiadjis_N",
                             prior_mean=pm, prior_var=pv)
        ref.append(np.append(o["mean_statistic"], o["loglikelihood_estimate"]))
    ref = np.array(ref)
    got = np.column_stack([g, ll])
    se = np.sqrt(got.var(axis=0) / C + ref.var(axis=0) / R)
    zscore = np.abs(got.mean(axis=0) - ref.mean(axis=0)) / se
    assert np.all(zscore < 5.0), (zscore, got.mean(axis=0), ref.mean(axis=0))
    # spread of the estimator itself is also the same (ratio of std within 35 %)
    ratio = got.std(axis=0) / ref.std(axis=0)
    assert np.all((ratio > 0.65) & (ratio < 1.5)), ratio


def test_chains_independent_and_reproducible():
    from sgmcmc_ssm_amd.ensemble import ChainEnsemble
    y = _series("svm", 80)
    p = default_params("svm")

    def run(offset, C=16):
        e = ChainEnsemble("svm", y, p, num_chains=C, N=128, epsilon=0.05, seed=3, chain_offset=offset)
        e.step(3)
        e.synchronize()
        return e.theta()
    a, b = run(0), run(0)
    np.testing.assert_array_equal(a, b)                       # same seeds -> same trajectories
    assert len({tuple(r) for r in a}) == 16                   # distinct streams -> distinct chains
    c = run(8)                                                # rank 1 of a 2-rank job with C=8
    np.testing.assert_array_equal(a[8:], c[:8])               # global chain id, not rank, fixes a chain


@pytest.mark.parametrize("windows", ["host", "device"])
def test_buffered_chains_do_not_depend_on_the_rank_partition(windows):
    """Buffered windows (S > 0): a chain's trajectory is fixed by its GLOBAL chain id -- window
    starts included -- whether 16 chains run as one rank or as two ranks of 8 (chain_offset 0 / 8)."""
    from sgmcmc_ssm_amd.ensemble import ChainEnsemble
    y = _series("svm", 120)
    p = default_params("svm")

    def run(offset, C):
        e = ChainEnsemble("svm", y, p, num_chains=C, N=128, epsilon=0.05, seed=11, chain_offset=offset,
                          subsequence_length=16, buffer_length=4, window_sampling=windows)
        e.step(4)
        e.synchronize()
        return e.theta()
    whole = run(0, 16)
    np.testing.assert_array_equal(whole[:8], run(0, 8))
    np.testing.assert_array_equal(whole[8:], run(8, 8))
    assert len({tuple(r) for r in whole}) == 16


def test_strict_partition_needs_divisible_length():
    """partition_style='strict' with T % S != 0 raises as random_subsequence_and_weights does
    (sgmcmc_sampler.py:1991-1993)."""
    from sgmcmc_ssm_amd.ensemble import ChainEnsemble
    y = _series("svm", 100)
    with pytest.raises(ValueError, match="does not evenly divide"):
        ChainEnsemble("svm", y, default_params("svm"), num_chains=4, N=64, subsequence_length=16, buffer_length=2,
                      partition_style="strict")
    e = ChainEnsemble("svm", y[:96], default_params("svm"), num_chains=4, N=64, subsequence_length=16, buffer_length=2,
                      partition_style="strict")
    e.step(2)
    e.synchronize()
    assert np.all(np.isfinite(e.theta()))


@pytest.mark.parametrize("model", ["svm", "lgssm", "garch"])
def test_sgld_update_kernel_matches_host_formula(model):
    """theta' - theta - eps*(grad_prior + ghat)/T must be N(0, 2 eps / T) noise:
    check mean (5 sigma) and variance (10 %) over 4096 chains, then the projection."""
    from sgmcmc_ssm_amd.ensemble import ChainEnsemble
    from sgmcmc_ssm_amd import _capi
    T, C, eps = 50, 4096, 0.02
    y = _series(model, T)
    rs = np.random.RandomState(1)
    p = default_params(model)
    P = _capi.THETA_DIM[model]
    theta0 = np.tile(p.theta(), (C, 1)) * rs.uniform(0.9, 1.05, size=(C, P))
    if model == "lgssm":
        theta0[:, 1] = 1.0
    prior = PRIORS[model].generate_default_prior(var=1.0, n=1, m=1)
    ens = ChainEnsemble(model, y, theta0, N=64, epsilon=eps, prior=prior, seed=11)
    ens.launch_pf()
    ens.synchronize()
    ghat, _ = ens.last_gradient_statistics()
    before = ens.theta()
    ens.launch_update()
    ens.synchronize()
    after = ens.theta()
    names = po.SCORE_NAMES[model]
    order = list(p.var_dict)
    resid = np.zeros((C, P))
    for c in range(0, C, 8):                                  # host formula on a subset (speed)
        q = ens._params_from_theta(before[c])
        gp = prior.grad_logprior(q)
        for j, var in enumerate(order):
            g = float(np.asarray(gp[var]).reshape(-1)[0]) + ghat[c, names.index(var)]
            resid[c, j] = after[c, j] - before[c, j] - eps * g / T
    sub = resid[::8]
    free = [j for j, var in enumerate(order) if not (model == "lgssm" and var == "C")]
    sd = np.sqrt(2 * eps / T)
    n = sub.shape[0]
    assert np.all(np.abs(sub[:, free].mean(axis=0)) < 5 * sd / np.sqrt(n)), sub.mean(axis=0)
    assert np.all(np.abs(sub[:, free].std(axis=0) / sd - 1) < 0.15), sub.std(axis=0) / sd
    if model == "lgssm":
        assert np.all(after[:, 1] == 1.0)                     # C pinned to identity
    # projection: |A| <= 0.9999, Cholesky factors reflected positive
    bad = theta0.copy()
    if model != "garch":
        bad[:, 0] = 0.99999
        bad[: C // 2, P - 1] = -0.8
        bad[: C // 2, P - 2] = -1.2
    else:
        bad[: C // 2, 3] = -0.8
    ens2 = ChainEnsemble(model, y, bad, N=64, epsilon=1e-6, prior=prior, seed=12)
    ens2.theta_dev[:, :P] = __import__("torch").from_numpy(bad).to(ens2.device)
    ens2.out_dev.zero_()
    ens2.launch_update()
    ens2.synchronize()
    th = ens2.theta()
    if model != "garch":
        assert np.all(np.abs(th[:, 0]) <= 0.9999 + 1e-12)
        assert np.all(th[:, P - 1] > 0) and np.all(th[:, P - 2] > 0)
    else:
        assert np.all(th[:, 3] > 0)


def test_buffered_windows_and_gather():
    from sgmcmc_ssm_amd.ensemble import ChainEnsemble
    y = _series("garch", 300)
    p = default_params("garch")
    ens = ChainEnsemble("garch", y, p, num_chains=64, N=256, epsilon=0.01, subsequence_length=16,
                        buffer_length=4, seed=2)
    ens.step(4)
    ens.synchronize()
    th = ens.theta()
    assert np.all(np.isfinite(th))
    d = ens._desc
    assert np.all(d["tL"] - d["t1"] == 16) and np.all(d["T"] <= 24) and np.all(d["T"] >= 20)
    # weights row used on the device equals random_subsequence_and_weights for that start
    from sgmcmc_ssm_amd.sgmcmc_sampler import random_subsequence_and_weights
    np.random.seed(0)
    s, e, w = random_subsequence_and_weights(16, 300)
    np.testing.assert_array_equal(ens._weights_table[s], w)
    g = ens.gather_samples().cpu().numpy()
    np.testing.assert_array_equal(g, th)


def test_systematic_resampling_extension():
    """EXTENSION (parity-unpinned: the reference has multinomial resampling only): systematic
    resampling estimates the same score / log-likelihood (5 standard errors) with no larger spread."""
    from sgmcmc_ssm_amd.ensemble import ChainEnsemble
    y = _series("svm", 80)
    p = default_params("svm")
    res = {}
    for mode in ("multinomial", "systematic"):
        ens = ChainEnsemble("svm", y, p, num_chains=768, N=300, epsilon=1e-3, seed=21, resampling=mode)
        ens.launch_pf()
        ens.synchronize()
        g, ll = ens.last_gradient_statistics()
        res[mode] = np.column_stack([g, ll])
    a, b = res["multinomial"], res["systematic"]
    se = np.sqrt(a.var(axis=0) / 768 + b.var(axis=0) / 768)
    assert np.all(np.abs(a.mean(axis=0) - b.mean(axis=0)) / se < 5.0)
    assert np.all(b.std(axis=0) < 1.15 * a.std(axis=0))
    with pytest.raises(ValueError):
        from sgmcmc_ssm_amd.particle_filters import make_problem
        make_problem("svm", "prior", "poyiadjis_N", y, p.theta(), 64, resampling="systematic")


def test_run_checkpoint_resume_and_sghmc():
    """run() collects thinned samples; a checkpoint taken mid-run resumes bit-identically (device
    RNG is keyed by (seed, chain, step)); SGHMC with friction 1 is exactly SGLD (extension)."""
    from sgmcmc_ssm_amd.ensemble import ChainEnsemble
    y = _series("garch", 200)
    p = default_params("garch")
    kw = dict(num_chains=32, N=128, epsilon=0.01, subsequence_length=16, buffer_length=4, seed=5)
    a = ChainEnsemble("garch", y, p, **kw)
    full = a.run(6, thin=2)
    assert full.shape == (3, 32, 4)
    b = ChainEnsemble("garch", y, p, **kw)
    first = b.run(2, thin=2)
    state = b.state_dict()
    c = ChainEnsemble("garch", y, p, **kw)
    c.load_state_dict(state)
    rest = c.run(4, thin=2)
    np.testing.assert_array_equal(np.concatenate([first, rest]), full)
    with pytest.raises(ValueError):
        ChainEnsemble("garch", y, p, **dict(kw, seed=6)).load_state_dict(state)
    # SGHMC: friction 1 == SGLD; friction < 1 keeps a momentum and still samples finite values
    s1 = ChainEnsemble("garch", y, p, sampler="sghmc", friction=1.0, **kw)
    np.testing.assert_array_equal(s1.run(6, thin=2), full)
    s2 = ChainEnsemble("garch", y, p, sampler="sghmc", friction=0.2, **kw)
    out = s2.run(6, thin=1)
    assert np.all(np.isfinite(out)) and not np.array_equal(out[-1], full[-1])
    assert float(s2.momentum_dev.abs().sum()) > 0.0


def test_device_window_sampling_and_graph_replay():
    """window_sampling='device': a Philox-keyed kernel rewrites the window descriptors in HBM
    (no host work per step); K steps captured in a hipGraph replay bitwise like K eager steps."""
    from sgmcmc_ssm_amd import _capi
    from sgmcmc_ssm_amd.ensemble import ChainEnsemble
    from sgmcmc_ssm_amd.models.garch import generate_garch_data
    p = default_params("garch")
    np.random.seed(4)
    T, S, B, C = 400, 16, 4, 512
    y = generate_garch_data(T=T, parameters=p)["observations"]
    kw = dict(num_chains=C, N=256, epsilon=0.01, subsequence_length=S, buffer_length=B, seed=11,
              window_sampling="device")
    ens = ChainEnsemble("garch", y, p, **kw)
    # descriptors after one device-side draw
    ens.launch_windows()
    ens.synchronize()
    d = np.frombuffer(ens.desc_dev.cpu().numpy().tobytes(), dtype=_capi.DEV_PROBLEM_DTYPE)
    left = (d["y"].astype(np.int64) - ens.y_dev.data_ptr()) // 8
    start = left + d["t1"]
    assert np.all((start >= 0) & (start <= T - S)) and np.all(d["tL"] - d["t1"] == S)
    assert np.all(left == np.maximum(0, start - B)) and np.all(left + d["T"] == np.minimum(T, start + S + B))
    assert np.all(d["weights"].astype(np.int64) == ens.weights_dev.data_ptr() + start * 8 * S)
    # starts ~ U{0..T-S}: mean and spread of 512 draws
    assert abs(start.mean() - (T - S) / 2) < 5 * (T - S) / np.sqrt(12 * C) and len(np.unique(start)) > 200
    # eager vs graph: same seeds, same trajectory, bit for bit
    a = ChainEnsemble("garch", y, p, **kw)
    b = ChainEnsemble("garch", y, p, **kw)
    sa = a.run(7, thin=1)
    sb = b.run(7, thin=3, graph_steps=3)                 # 2 replays of 3 steps + 1 eager step
    assert sb.shape == (2, C, 4) and np.all(np.isfinite(sa))
    np.testing.assert_array_equal(sb[0], sa[2])
    np.testing.assert_array_equal(sb[1], sa[5])
    np.testing.assert_array_equal(b.theta(), sa[6])
    assert a.steps_done == b.steps_done == 7
    # a different step draws different windows; the host-sampled ensemble cannot be captured
    with pytest.raises(ValueError):
        ChainEnsemble("garch", y, p, **dict(kw, window_sampling="host")).run(4, graph_steps=2)
    # full-sequence chains need no windows: graph works with the default settings
    pf_ = default_params("svm")
    ys = GEN["svm"](T=120, parameters=pf_)["observations"]
    c = ChainEnsemble("svm", ys, pf_, num_chains=64, N=128, epsilon=0.05, seed=2)
    e = ChainEnsemble("svm", ys, pf_, num_chains=64, N=128, epsilon=0.05, seed=2)
    np.testing.assert_array_equal(c.run(4, thin=4, graph_steps=4)[0], e.run(4, thin=4)[0])


@pytest.mark.parametrize("N", [1024, 1000, 4096])
def test_device_generator_normals_and_uniform_streams(N):
    """The device generator's Gaussian draws (f32 transcendental units, widened to fp64): a T = 0
    window returns x0 = prior_mean + sd * z, so z is observable.  1M draws: moments, tails and a
    Kolmogorov-Smirnov test against N(0,1); distinct streams / seeds give distinct draws."""
    from scipy import stats
    from sgmcmc_ssm_amd import _capi
    ctx = _capi.default_context(0)
    B = 1 << 20
    B = B // N
    probs = [dict(model="svm", kernel="prior", smoother="nemeth", stat="score", dtype="f64", rng="device", N=N,
                  t1=0, tL=0, lambduh=1.0, prior_mean=0.5, prior_var=4.0, y=np.zeros(0), theta=[0.9, 1.0, 1.0],
                  seed=99, stream=b) for b in range(B)]
    outs = ctx.run_batch(probs, want_final=True)
    z = (np.concatenate([o["x_t"][:, 0] for o in outs]) - 0.5) / 2.0
    n = z.size
    assert abs(z.mean()) < 5 / np.sqrt(n) and abs(z.var() - 1) < 5 * np.sqrt(2.0 / n)
    assert abs(stats.kurtosis(z)) < 5 * np.sqrt(24.0 / n) and abs(stats.skew(z)) < 5 * np.sqrt(6.0 / n)
    assert stats.kstest(z, "norm").pvalue > 1e-4
    tail = np.mean(np.abs(z) > 4.0)
    assert 2e-5 < tail < 1.5e-4 and np.abs(z).max() < 6.8          # P(|z|>4) = 6.3e-5
    assert len(np.unique(z)) > 0.99 * n                              # no stream reuse across windows / lanes
    # reproducible for a given (seed, stream, kernel variant): the same launch again gives the same
    # draws (a launch of <= 64 windows runs the latency variant, whose lanes own other particles)
    again = ctx.run_batch(probs[:80], want_final=True)
    assert np.array_equal(again[0]["x_t"], outs[0]["x_t"]) and np.array_equal(again[79]["x_t"], outs[79]["x_t"])
    assert not np.array_equal(again[1]["x_t"], outs[0]["x_t"])


def test_sequence_list_ensemble_matches_seq_sampler():
    """ChainEnsemble over a LIST of sequences (the Seq*Sampler / EURUS-segments setting): every
    chain and step draws one sequence and a buffered window inside it, the gradient rescaled by
    T_total / T_sequence through the resident weights.  Descriptor invariants, and the mean
    gradient over 4096 chains against SeqSVMSampler._noisy_grad_loglikelihood(num_sequences=1)."""
    from sgmcmc_ssm_amd import _capi
    from sgmcmc_ssm_amd.ensemble import ChainEnsemble
    from sgmcmc_ssm_amd.models.svm import SeqSVMSampler
    p = default_params("svm")
    np.random.seed(6)
    y = GEN["svm"](T=137, parameters=p)["observations"]
    cuts = [0, 40, 52, 77, 137]                      # lengths 40, 12 (<= S: whole sequence), 25, 60
    seqs = [y[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
    S, B, C, N = 16, 4, 4096, 128
    ens = ChainEnsemble("svm", seqs, p, num_chains=C, N=N, epsilon=1e-3, subsequence_length=S, buffer_length=B,
                        seed=21)
    ens.step(1)                                      # second draw of windows happens inside step()
    ens.synchronize()
    d = ens._desc
    left = (d["y"].astype(np.int64) - ens.y_dev.data_ptr()) // 8
    seg = np.searchsorted(np.array(cuts), left, side="right") - 1
    lo, hi = np.array(cuts)[seg], np.array(cuts)[seg + 1]
    assert np.all(left + d["T"] <= hi) and np.all(left >= lo)             # windows never cross a sequence
    assert set(np.unique(seg)) == {0, 1, 2, 3}
    short = (hi - lo) <= S
    assert np.all(d["tL"][short] - d["t1"][short] == (hi - lo)[short]) and np.all(d["tL"][~short] - d["t1"][~short] == S)
    g, _ = ens.last_gradient_statistics()
    # reference semantics through the drop-in Seq sampler (device generator), 600 draws
    sampler = SeqSVMSampler(n=1, m=1, observations=seqs, parameters=p.copy())
    np.random.seed(3)
    ref = np.array([vec("svm", sampler._noisy_grad_loglikelihood(
        num_sequences=1, kind="pf", pf="poyiadjis_N", N=N, subsequence_length=S, buffer_length=B, rng="device"))
        for _ in range(600)])
    got = g[:, [2, 1, 0]]                            # score columns [LRinv, LQinv, A] -> var_dict order
    se = np.sqrt(got.var(axis=0) / C + ref.var(axis=0) / len(ref))
    z = np.abs(got.mean(axis=0) - ref.mean(axis=0)) / se
    assert np.all(z < 5.0), (z, got.mean(axis=0), ref.mean(axis=0))
    assert np.all(np.isfinite(ens.theta()))
    with pytest.raises(NotImplementedError):
        ChainEnsemble("svm", seqs, p, num_chains=4, N=N, subsequence_length=S, buffer_length=B, window_sampling="device")


def test_full_size_workload_device_vs_replay():
    """BASELINE's full size (SVM T=1000, N=1000), the bench workload itself: the device-generator
    kernel the bench times (wg256x4s, 32-bit CDF, thread-major order, f32-unit normals, fused
    math) against the REPLAY kernel that is pinned bit-level to the reference, as estimators of the
    same score / log-likelihood: means over 3072 device chains vs 48 replayed seeds within 5
    standard errors, comparable spread, and the log-likelihood identity E[exp(ll)] (both unbiased)."""
    from sgmcmc_ssm_amd import _capi
    from sgmcmc_ssm_amd.ensemble import ChainEnsemble
    from sgmcmc_ssm_amd.particle_filters import draw_replay_streams
    p = default_params("svm")
    np.random.seed(12345)
    y = GEN["svm"](T=1000, parameters=p)["observations"]
    C, N, R = 3072, 1000, 48
    ens = ChainEnsemble("svm", y, p, num_chains=C, N=N, epsilon=1e-4, seed=77)
    assert ens.ctx.variant_name("svm", "prior", "f64", "device", N) == "wg256x4s"
    ens.launch_pf()
    ens.synchronize()
    g, ll = ens.last_gradient_statistics()
    got = np.column_stack([g, ll])
    ctx = _capi.default_context(0)
    rs = np.random.RandomState(5)
    ref = []
    for _ in range(R):
        z0, u, z = draw_replay_streams(N, 1000, rs)
        o = ctx.run_batch([dict(model="svm", kernel="prior", smoother="nemeth", stat="score", dtype="f64",
                                rng="replay", N=N, t1=0, tL=1000, lambduh=1.0, prior_mean=0.0, prior_var=10.0,
                                y=y.reshape(-1), theta=p.theta(), z0=z0, u=u, z=z)])[0]
        ref.append(np.append(o["mean_stat"], o["loglik"]))
    ref = np.array(ref)
    se = np.sqrt(got.var(axis=0) / C + ref.var(axis=0) / R)
    zscore = np.abs(got.mean(axis=0) - ref.mean(axis=0)) / se
    assert np.all(zscore < 5.0), (zscore, got.mean(axis=0), ref.mean(axis=0))
    ratio = got.std(axis=0) / ref.std(axis=0)
    assert np.all((ratio > 0.6) & (ratio < 1.6)), ratio


def test_device_generator_large_sample():
    """The inputs of the filter as the bench instantiation draws them, 2.6e8 of each kind reduced on the device
    (tools/generator_tails.py; profiles/r03_generator_tails.txt holds the 2e9-draw run): moments, two-sided tails out to
    5.5 sigma against N(0,1) within 5 Poisson standard errors, chi-square of 512 equiprobable bins, of the words' top
    12 and low 8 bits, no correlation between neighbouring chains or between a child's word and its normal."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import generator_tails
    o = generator_tails.collect(chains=128, launches=2)
    assert o["variant"] == "wg256x4s" and o["draws"] == 256 * 10 ** 6
    assert all(abs(o["moments"][k]) < 5 for k in ("z_mean", "z_var", "z_third", "z_fourth")), o["moments"]
    for t in o["tails"]:
        if t["expected"] >= 5:
            assert abs(t["z"]) < 5, t
    assert o["max_abs_z"] < 6.8
    for k in ("chi2_normal_equiprobable", "chi2_words_top12", "chi2_words_low8"):
        assert o[k]["p"] > 1e-5, (k, o[k])
    assert abs(o["corr_neighbouring_chains"]["z"]) < 5 and abs(o["corr_word_normal"]["z"]) < 5


def test_poyiadjis_n_twin_of_the_1024_thread_unit():
    """PFG_SMOOTHER_POYIADJIS_N launches (what ChainEnsemble issues for lambduh = 1) run the 1024 x 4 fp64 unit's twin with
    the filter / lambda != 1 / other statistics compiled out: the numbers of the general kernel (same key, same draws; to
    the last place or two -- the compiler fuses the score's multiply-adds differently where the general branch is gone), and a descriptor that is not (NEMETH, lambduh = 1, score) gets NaNs from it, not another estimator's result."""
    import torch
    from sgmcmc_ssm_amd.ensemble import ChainEnsemble
    from sgmcmc_ssm_amd import _capi
    y = _series("svm", 40)
    p = default_params("svm")
    ens = ChainEnsemble("svm", y, p, num_chains=6, N=4000, epsilon=1e-3, seed=21)
    st = torch.cuda.current_stream().cuda_stream
    ens.launch_pf()                                       # lambduh = 1: the POYIADJIS_N launch
    ens.synchronize()
    assert ens.ctx.last_variant() == "wg1024x4s_score1"
    twin = ens.out_dev.cpu().numpy().copy()
    ens.out_dev.zero_()
    ens.ctx.launch_device(ens.model, ens.kernel, ens.dtype, "device", ens.N, ens.C, ens.desc_dev.data_ptr(), st)
    ens.synchronize()
    assert ens.ctx.last_variant() == "wg1024x4s"
    general = ens.out_dev.cpu().numpy().copy()
    assert np.isfinite(general[:, :5]).all() and np.allclose(twin, general, rtol=1e-13, atol=0.0)      # another specialisation: last-place differences allowed
    # chain 2 asks for lambduh = 0.9, chain 4 for the filter: the twin refuses them, the others are untouched
    d = ens._desc.copy()
    d["lambduh"][2] = 0.9
    d["smoother"][4] = _capi.SMOOTHER["filter"]
    bad = torch.from_numpy(d.view(np.uint8).reshape(ens.C, -1).copy()).to(ens.device)
    ens.out_dev.zero_()
    ens.ctx.launch_device_smoother(ens.model, ens.kernel, ens.dtype, "device", "poyiadjis_n", ens.N, ens.C, bad.data_ptr(), st)
    ens.synchronize()
    o = ens.out_dev.cpu().numpy()
    assert np.isnan(o[[2, 4]]).all() and np.allclose(o[[0, 1, 3, 5]], general[[0, 1, 3, 5]], rtol=1e-13, atol=0.0)
    # the general kernel serves them
    ens.ctx.launch_device(ens.model, ens.kernel, ens.dtype, "device", ens.N, ens.C, bad.data_ptr(), st)
    ens.synchronize()
    assert np.isfinite(ens.out_dev.cpu().numpy()[:, :5]).all()
