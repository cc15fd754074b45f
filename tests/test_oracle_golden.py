"""Pin the CPU oracle: it must reproduce the REFERENCE's outputs (tests/golden, produced
by tests/golden/make_golden.py from /root/reference) with absolute error 0.0."""
import numpy as np
import pytest

from oracle import pf_oracle as po


def _run(meta, g, save_all):
    key = meta["key"]
    y = g.get(key, "y")
    theta = g.get(key, "theta")
    weights = g.get(key, "weights")
    rng = np.random.RandomState(meta["seed"])
    z0, u, z = po.draw_streams(rng, meta["N"], meta["T"])
    return po.pf_window(meta["model"], theta, y, meta["N"], z0, u, z,
                        kernel=meta["kernel"], pf=meta["pf"], lambduh=meta["lambduh"],
                        stat=meta["stat"], t1=meta["t1"], tL=meta["tL"], weights=weights,
                        prior_mean=meta["prior_mean"], prior_var=meta["prior_var"],
                        save_all=save_all)


def test_trace_cases_bit_exact(golden_trace):
    g = golden_trace
    assert len(g.meta) == 25
    for meta in g.meta:
        out = _run(meta, g, save_all=True)
        for name in ("all_x_t", "all_log_weights", "all_statistics",
                     "all_loglikelihood_estimate"):
            ref = g.get(meta["key"], name)
            got = np.asarray(out[name], dtype=float)
            assert got.shape == ref.shape, (meta, name)
            assert np.array_equal(got, ref), (meta, name, np.max(np.abs(got - ref)))
        if meta["pf"] != "filter":
            assert np.array_equal(out["mean_statistic"], g.get(meta["key"], "mean_statistic"))


def test_window_cases_bit_exact(golden_window):
    g = golden_window
    for meta in g.meta:
        if meta["N"] * meta["T"] > 300000:
            continue   # the big ones run in test_window_big (kept separate for timing)
        out = _run(meta, g, save_all=False)
        key = meta["key"]
        assert out["loglikelihood_estimate"] == float(g.get(key, "loglikelihood_estimate")), meta
        if meta["pf"] != "filter":
            assert np.array_equal(out["mean_statistic"], g.get(key, "mean_statistic")), meta
        else:
            assert np.array_equal(out["statistics"], g.get(key, "statistics")), meta
        if g.get(key, "x_t") is not None:
            assert np.array_equal(out["x_t"], g.get(key, "x_t"))
            assert np.array_equal(out["log_weights"], g.get(key, "log_weights"))


def test_window_big(golden_window):
    g = golden_window
    n = 0
    for meta in g.meta:
        if meta["N"] * meta["T"] <= 300000 or meta["stat"] != "score":
            continue
        out = _run(meta, g, save_all=False)
        key = meta["key"]
        assert out["loglikelihood_estimate"] == float(g.get(key, "loglikelihood_estimate")), meta
        assert np.array_equal(out["mean_statistic"], g.get(key, "mean_statistic")), meta
        n += 1
    assert n >= 2


def test_known_answer_svm(golden_window):
    """SVM A=.95,Q=.5,R=.5, data seed 12345, np.random.seed(99), N=1000, T=1000."""
    meta = golden_window.meta[0]
    assert (meta["model"], meta["N"], meta["T"], meta["seed"]) == ("svm", 1000, 1000, 99)
    ms = golden_window.get(meta["key"], "mean_statistic")
    np.testing.assert_allclose(ms, [5.62738493, 9.88344914, -123.63120925], rtol=0, atol=1e-7)
    assert abs(float(golden_window.get(meta["key"], "loglikelihood_estimate")) + 1561.7975224303605) < 1e-9


def test_stream_order_matches_global_state():
    """draw_streams on the global legacy state == on a RandomState with the same seed."""
    np.random.seed(5)
    a = po.draw_streams(np.random, 7, 3)
    b = po.draw_streams(np.random.RandomState(5), 7, 3)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_multinomial_matches_numpy_choice():
    rng = np.random.RandomState(3)
    p = rng.dirichlet(np.ones(50))
    s1 = np.random.RandomState(11)
    idx = s1.choice(range(50), size=50, replace=True, p=p)
    s2 = np.random.RandomState(11)
    u = s2.random_sample(50)
    assert np.array_equal(idx, po.multinomial_ancestors(p, u))
    # both consumed exactly 50 doubles
    assert s1.random_sample() == s2.random_sample()


def test_device_ancestors_are_the_references_resampler_in_slot_order():
    """Ties `po.device_ancestors` -- the restatement of the device-generator kernels' resampling the recorded-draw tests
    replay -- to `po.multinomial_ancestors`, the reference's np.random.choice semantics (pf.py:26-30).  The kernels keep
    the CDF in THREAD-major slot order (slot q = tid PPT + k holds particle k NT + tid; slots past N weigh 0): on the
    slot-permuted weights, with the uniform the kernel's word stands for, u = (word + 0.5) / 2^32, the reference's
    cumsum / searchsorted must pick the same slot
      * 'f64_uniform' (pf_big_kernel / whole-GPU window: an fp64 CDF against the uniform itself): always;
      * 'f64': always;
      * 'fixed32' (LDS-resident kernels: floor(cdf 2^32) against the raw word): except where a CDF edge falls inside the
        upper half of the word's 2^-32 cell, (word + 0.5, word + 1) / 2^32 -- expected N^2 / 2^33 children per step, each
        then one slot further."""
    rs = np.random.RandomState(2024)
    n_fixed_diff, n_fixed_total = 0, 0
    for NT, PPT, N, spread in ((256, 4, 1000, 2.0), (256, 4, 1024, 30.0), (64, 2, 100, 1.0), (1024, 4, 4000, 5.0), (512, 2, 777, 0.0),
                               (256, 1, 200, 80.0), (1024, 1, 1000, 3.0)):
        NP = NT * PPT
        q = np.arange(NP)
        particle = (q % PPT) * NT + q // PPT
        for rep in range(6):
            logw = rs.randn(N) * spread
            if rep == 5:
                logw[:] = -1e300
                logw[rs.randint(N)] = 0.0           # one particle carries everything
            p = np.exp(logw - np.max(logw))
            w_slots = np.where(particle < N, p[np.minimum(particle, N - 1)], 0.0)
            words = rs.randint(0, 2 ** 32, size=N, dtype=np.uint64).astype(np.uint32)
            u = (words.astype(np.float64) + 0.5) / 4294967296.0
            pos_ref = po.multinomial_ancestors(w_slots / np.sum(w_slots), u)                 # the reference, on the slots
            anc_ref = np.minimum((np.minimum(pos_ref, NP - 1) % PPT) * NT + np.minimum(pos_ref, NP - 1) // PPT, N - 1)
            assert np.array_equal(po.device_ancestors(logw, u, NT, PPT, cdf="f64_uniform"), anc_ref), (NT, PPT, N, rep)
            assert np.array_equal(po.device_ancestors(logw, words, NT, PPT, cdf="f64"), anc_ref), (NT, PPT, N, rep)
            a32 = po.device_ancestors(logw, words, NT, PPT, cdf="fixed32")
            diff = np.flatnonzero(a32 != anc_ref)
            n_fixed_diff += diff.size
            n_fixed_total += N
            # a child that differs took the next slot with positive weight
            slot_of = np.empty(N, dtype=np.int64)
            slot_of[particle[particle < N]] = q[particle < N]
            for i in diff:
                assert slot_of[a32[i]] > slot_of[anc_ref[i]], (NT, PPT, N, rep, i)
                between = w_slots[slot_of[anc_ref[i]] + 1:slot_of[a32[i]]]
                assert np.all(between == 0.0)
    # expectation over these cases: sum N^2 / 2^33 ~ 2e-3
    assert n_fixed_diff <= 2, (n_fixed_diff, n_fixed_total)


def test_paris_oracle_bit_exact():
    """PaRIS smoother restatement vs the reference's traces (default settings and forced
    manual-sampling fallback), consuming the legacy stream in the reference's order."""
    from conftest import Golden
    g = Golden("paris.npz")
    assert len(g.meta) == 15
    for meta in g.meta:
        key = meta["key"]
        rng = np.random.RandomState(meta["seed"])
        out = po.pf_window_paris_rng(meta["model"], g.get(key, "theta"), g.get(key, "y"), meta["N"], rng=rng,
                                     kernel=meta["kernel"], stat="score", t1=meta["t1"], tL=meta["tL"],
                                     weights=g.get(key, "weights"), prior_mean=meta["prior_mean"],
                                     prior_var=meta["prior_var"], save_all=True, **meta["kwargs"])
        for name in ("all_x_t", "all_log_weights", "all_statistics", "all_loglikelihood_estimate"):
            assert np.array_equal(np.asarray(out[name], dtype=float), g.get(key, name)), (meta, name)
        assert np.array_equal(out["mean_statistic"], g.get(key, "mean_statistic"))


def test_predictive_oracle_bit_exact():
    """Predictive log-likelihood restatement (filter + logsumexp, model-specific k-step statistics)
    vs the reference, consuming the legacy stream in the reference's order."""
    from conftest import Golden
    g = Golden("predictive.npz")
    assert len(g.meta) == 6
    for m in g.meta:
        kernel = m["kernel"] or po.DEFAULT_KERNEL[m["model"]]
        rng = np.random.RandomState(m["seed"])
        pred = po.pf_predictive_loglikelihood_estimate(
            m["model"], g.get(m["key"], "theta"), g.get(m["key"], "y"), m["N"], rng=rng, num_steps_ahead=m["K"],
            kernel=kernel, t1=m["t1"], tL=m["tL"], prior_mean=m["prior_mean"], prior_var=m["prior_var"])
        assert np.array_equal(pred, g.get(m["key"], "pred")), (m, pred, g.get(m["key"], "pred"))


def test_poyiadjis_n2_oracle_bit_exact():
    """The O(N^2) Poyiadjis smoother (pf.py:84-136) restatement vs the reference: traced tiny
    cases for every (model, kernel) and statistic, window-level cases up to N = 300."""
    from conftest import Golden
    g = Golden("n2.npz")
    assert len(g.meta) == 13
    for meta in g.meta:
        out = _run(meta, g, save_all=meta["traced"])
        key = meta["key"]
        assert out["loglikelihood_estimate"] == float(g.get(key, "loglikelihood_estimate")), meta
        for name in ("x_t", "log_weights", "statistics", "mean_statistic"):
            assert np.array_equal(np.asarray(out[name], dtype=float), g.get(key, name)), (meta, name)
        if meta["traced"]:
            for name in ("all_x_t", "all_log_weights", "all_statistics", "all_loglikelihood_estimate"):
                assert np.array_equal(np.asarray(out[name], dtype=float), g.get(key, name)), (meta, name)


def test_theta_grid_oracle_bit_exact():
    """Round 3: the reference run over a GRID of parameters per (model, kernel) -- LGSSM C = 0.3 / 1.7 (the
    optimal kernel's weight "assumes C = 1", models/lgssm/kernels.py:117-120: pinned as the reference computes it),
    |A| = 0.9999, Cholesky factors 0.1 and 10, GARCH phi = 0.999 and lambduh = 0.01 / 0.99
    (models/garch/kernels.py:136-180): traced tiny cases (every step) and N = 1000 windows, abs-err 0.0."""
    from conftest import Golden
    g = Golden("theta_grid.npz")
    assert len(g.meta) == 76
    tags = {(m["model"], m["tag"]) for m in g.meta}
    assert len(tags) == 11
    for meta in g.meta:
        out = _run(meta, g, save_all=meta["traced"])
        key = meta["key"]
        if meta["traced"]:
            for name in ("all_x_t", "all_log_weights", "all_statistics", "all_loglikelihood_estimate"):
                ref = g.get(key, name)
                got = np.asarray(out[name], dtype=float)
                assert got.shape == ref.shape and np.array_equal(got, ref), (meta, name, np.max(np.abs(got - ref)))
            if meta["pf"] != "filter":
                assert np.array_equal(out["mean_statistic"], g.get(key, "mean_statistic")), meta
        else:
            assert out["loglikelihood_estimate"] == float(g.get(key, "loglikelihood_estimate")), meta
            assert np.array_equal(out["mean_statistic"], g.get(key, "mean_statistic")), meta
            assert np.array_equal(out["log_weights"], g.get(key, "log_weights")), meta


def test_paris_seed_fixtures_oracle_bit_exact():
    """Round 3: PaRIS through the reference's public entry points at the demos' particle count (paris_seed.npz): the
    oracle in np.random order reproduces gradient and log-likelihood with abs-err 0.0 and leaves the generator where
    the reference leaves it (the next draw), incl. accept_reject=False (pf.py:226-236) and non-default thresholds."""
    from conftest import Golden
    g = Golden("paris_seed.npz")
    n = 0
    for m in g.meta:
        if m["kind"] != "helper":
            continue
        key = m["key"]
        theta, y = g.get(key, "theta"), g.get(key, "y")
        if m["model"] == "garch":
            pm, pv = po.garch_prior_x(theta)
            pv = float(np.asarray(pv).reshape(-1)[0])
        else:
            prec = float(g.get(key, "fm_precision")[0])
            pv = 1.0 / prec
            pm = float(g.get(key, "fm_mean_precision")[0]) / pv           # the reference's solve(prior_var, mean_precision)
        kernel = m["kernel"] or po.DEFAULT_KERNEL[m["model"]]
        for stat, ref_name, next_name in (("score", "grad", "next_draw"), ("suff", "loglik", "next_draw_loglik")):
            rng = np.random.RandomState(m["seed"])
            out = po.pf_window_paris_rng(m["model"], theta, y, m["N"], rng=rng, kernel=kernel, stat=stat, t1=m["t1"], tL=m["tL"],
                                         weights=g.get(key, "weights"), prior_mean=pm, prior_var=pv, **m["kwargs"])
            if stat == "score":
                got = dict(zip(po.SCORE_NAMES[m["model"]], out["mean_statistic"]))
                names = {"svm": ("A", "LQinv_vec", "LRinv_vec"), "lgssm": ("A", "C", "LQinv_vec", "LRinv_vec"),
                         "garch": ("log_mu", "logit_phi", "logit_lambduh", "LRinv_vec")}[m["model"]]
                assert np.array_equal(np.array([got[k] for k in names]), g.get(key, "grad")), m
            else:
                assert out["loglikelihood_estimate"] == float(g.get(key, "loglik")), m
            assert rng.random_sample() == float(g.get(key, next_name)), (m, stat)
        n += 1
    assert n == 11


def test_giant_fixtures_oracle_bit_exact():
    """Round 4: the reference's own giant-N calls (giant.npz: the 48-step window of svm_grad_compare.py:58-82 with
    N = 10^5 and 10^6, garch_grad_compare.py:66-93, an LGSSM Nemeth window, a filter log-likelihood) -- the oracle
    reproduces gradient and log-likelihood with abs-err 0.0 and leaves the generator where the reference leaves it.
    (This pins the oracle at the sizes where the kernels' resampling CDF has to be NumPy's cumsum bit for bit.)"""
    from conftest import Golden
    g = Golden("giant.npz")
    names = {"svm": ("A", "LQinv_vec", "LRinv_vec"), "lgssm": ("A", "C", "LQinv_vec", "LRinv_vec"),
             "garch": ("log_mu", "logit_phi", "logit_lambduh", "LRinv_vec")}
    n = 0
    for m in g.meta:
        key = m["key"]
        theta, y = g.get(key, "theta"), g.get(key, "y")
        if m["model"] == "garch":
            pm, pv = po.garch_prior_x(theta)
            pv = float(np.asarray(pv).reshape(-1)[0])
        else:
            pv = 1.0 / float(g.get(key, "fm_precision")[0])
            pm = float(g.get(key, "fm_mean_precision")[0]) / pv
        kernel = m["kernel"] or po.DEFAULT_KERNEL[m["model"]]
        kw = dict(kernel=kernel, pf=m["pf"], t1=m["t1"], tL=m["tL"], weights=g.get(key, "weights"), prior_mean=pm, prior_var=pv,
                  **m["kwargs"])
        if m["has_grad"]:
            rng = np.random.RandomState(m["seed"])
            got = po.pf_gradient_estimate(m["model"], theta, y, m["N"], rng=rng, **kw)
            assert np.array_equal(np.array([got[k] for k in names[m["model"]]]), g.get(key, "grad")), m
            assert rng.random_sample() == float(g.get(key, "next_draw")), m
        if m["has_loglik"]:
            rng = np.random.RandomState(m["seed"])
            ll = po.pf_loglikelihood_estimate(m["model"], theta, y, m["N"], rng=rng, **kw)
            assert ll == float(g.get(key, "loglik")), m
            assert rng.random_sample() == float(g.get(key, "next_draw_loglik")), m
        n += 1
    assert n == 7
