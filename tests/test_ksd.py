"""KSD evaluation: batched trace gradients + IMQ kernel Stein discrepancy."""
import numpy as np
import pytest

from oracle import ksd_oracle
from conftest import Golden
from test_host_logic import default_params, vec, SAMPLERS, run_windows_oracle
from sgmcmc_ssm_amd import particle_filters


@pytest.fixture(scope="module")
def golden_ksd():
    return Golden("ksd.npz")


def test_ksd_oracle_matches_reference(golden_ksd):
    for m in golden_ksd.meta:
        v = ksd_oracle.imq_ksd(golden_ksd.get(m["key"], "x"), golden_ksd.get(m["key"], "g"), c=m["c"], beta=m["beta"])
        ref = float(golden_ksd.get(m["key"], "value"))
        assert abs(v - ref) <= 1e-12 * ref, (m, v, ref)       # same terms, different summation order


def _trace(model, K=5):
    rs = np.random.RandomState(8)
    plist = []
    for _ in range(K):
        p = default_params(model)
        for k in p.var_dict:
            p.var_dict[k] = p.var_dict[k] * rs.uniform(0.9, 1.05)
        plist.append(p.project_parameters())
    return plist


def _check_trace_equals_loop(model, seq, cmp):
    from test_host_logic import GEN
    np.random.seed(5)
    y = GEN[model](T=120, parameters=default_params(model))["observations"]
    Sampler, SeqSampler = SAMPLERS[model]
    if seq:
        sampler = SeqSampler(n=1, m=1, observations=[y[:50], y[50:90], y[90:]], parameters=default_params(model))
        kw = dict(kind="pf", pf="poyiadjis_N", N=64, subsequence_length=8, buffer_length=2, num_sequences=2,
                  is_scaled=False)
    else:
        sampler = Sampler(n=1, m=1, observations=y, parameters=default_params(model))
        kw = dict(kind="pf", pf="nemeth", N=64, subsequence_length=16, buffer_length=3, minibatch_size=2)
    plist = _trace(model)
    before = sampler.parameters
    np.random.seed(42)
    batched = sampler.noisy_gradient_trace(plist, **kw)
    assert sampler.parameters is before
    np.random.seed(42)
    looped = []
    for p in plist:
        sampler.parameters = p
        looped.append(sampler.noisy_gradient(**kw))
    for a, b in zip(batched, looped):
        cmp(vec(model, a), vec(model, b))


@pytest.mark.parametrize("model,seq", [("svm", False), ("garch", True), ("lgssm", False), ("svm", True)])
def test_gradient_trace_equals_loop_cpu(monkeypatch, model, seq):
    monkeypatch.setattr(particle_filters, "run_windows", run_windows_oracle)
    _check_trace_equals_loop(model, seq, np.testing.assert_array_equal)


@pytest.mark.gpu
def test_imq_ksd_kernel_matches_reference(golden_ksd):
    from sgmcmc_ssm_amd.trace_metric_functions import IMQ_KSD, compute_KSD
    for m in golden_ksd.meta:
        x, g = golden_ksd.get(m["key"], "x"), golden_ksd.get(m["key"], "g")
        v = IMQ_KSD(x, g, c=m["c"], beta=m["beta"])
        ref = float(golden_ksd.get(m["key"], "value"))
        assert abs(v - ref) <= 1e-10 * ref, (m, v, ref)
    # compute_KSD on a Parameters trace
    plist = _trace("svm", 12)
    rs = np.random.RandomState(2)
    grads = [[rs.normal(size=(1, 1)), rs.normal(size=1), rs.normal(size=1)] for _ in plist]
    res = compute_KSD(plist, grads, variables=["A", "LQinv_vec", "LRinv_vec"])
    for ii, var in enumerate(["A", "LQinv_vec", "LRinv_vec"]):
        x = np.array([np.asarray(getattr(p, var)).flatten() for p in plist])
        g = np.array([np.asarray(gr[ii]).flatten() for gr in grads])
        assert abs(res[var] - ksd_oracle.imq_ksd(x, g)) <= 1e-10 * res[var]
    with pytest.raises(ValueError):
        IMQ_KSD(np.zeros((3, 2)), np.zeros((3, 1)))


@pytest.mark.gpu
@pytest.mark.parametrize("model,seq", [("svm", False), ("garch", True)])
def test_gradient_trace_equals_loop_gpu(model, seq):
    _check_trace_equals_loop(model, seq, lambda a, b: np.testing.assert_allclose(a, b, rtol=1e-9, atol=1e-9))
