"""INTEGRATION.md section B -- the reference-side ctypes stub a maintainer would paste as
sgmcmc_ssm/particle_filters/hip_backend.py -- is EXECUTED here: the code block is extracted from the document as it
stands, run against the in-tree libpfgrad.so, and `buffered_pf_wrapper_hip` is called the way the reference's helpers
would call it (a Kernel INSTANCE whose class name selects the model, a Parameters-like object with `var_dict`, the
global np.random stream) on reference fixtures.  A drift between the documented binding and include/pfgrad.h (struct
sizes, field order, enum values) fails here, not at a user's desk."""
import collections
import os
import re

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_source():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## B."):]
    m = re.search(r"```python\n(.*?)```", sec, re.S)
    assert m, "INTEGRATION.md section B lost its code block"
    return m.group(1)


class SVMPriorKernel(object):          # stand-ins with the REFERENCE's class names: the stub dispatches on type(kernel).__name__
    pass


class GARCHOptimalKernel(object):
    pass


class LGSSMOptimalKernel(object):
    pass


class LGSSMPriorKernel(object):
    pass


class GARCHPriorKernel(object):
    pass


KERNELS = {("svm", "prior"): SVMPriorKernel, ("garch", "optimal"): GARCHOptimalKernel, ("garch", "prior"): GARCHPriorKernel,
           ("lgssm", "optimal"): LGSSMOptimalKernel, ("lgssm", "prior"): LGSSMPriorKernel}
VAR_NAMES = {"svm": ("A", "LQinv_vec", "LRinv_vec"), "lgssm": ("A", "C", "LQinv_vec", "LRinv_vec"),
             "garch": ("log_mu", "logit_phi", "logit_lambduh", "LRinv_vec")}


class _Params(object):
    def __init__(self, model, theta):
        # insertion order = the reference's Parameters.var_dict order (pfgrad.h: theta layout)
        self.var_dict = collections.OrderedDict((k, np.atleast_1d(float(v))) for k, v in zip(VAR_NAMES[model], theta))


@pytest.fixture(scope="module")
def stub():
    import torch  # noqa: F401   (one HIP runtime per process: torch's first, as sgmcmc_ssm_amd._capi does)
    from sgmcmc_ssm_amd import _build
    src = _stub_source()
    assert 'C.CDLL("libpfgrad.so")' in src
    src = src.replace('C.CDLL("libpfgrad.so")', "C.CDLL({0!r})".format(_build.LIB_PATH))
    ns = {}
    exec(compile(src, "INTEGRATION.md#B", "exec"), ns)
    return ns


def test_documented_stub_reproduces_reference_fixtures(stub, golden_trace):
    g = golden_trace
    n = 0
    for m in g.meta:
        if m["pf"] not in ("poyiadjis_N", "nemeth", "filter") or m["stat"] != "score":
            continue
        key = m["key"]
        kw = {}
        if m["pf"] == "nemeth":
            kw["lambduh"] = 0.95 if m["lambduh"] is None else m["lambduh"]
        np.random.seed(m["seed"])
        out = stub["buffered_pf_wrapper_hip"](m["pf"], g.get(key, "y").reshape(-1, 1), _Params(m["model"], g.get(key, "theta")), m["N"],
                                              KERNELS[(m["model"], m["kernel"])](), t1=m["t1"], tL=m["tL"],
                                              weights=g.get(key, "weights"), prior_mean=np.array([m["prior_mean"]]),
                                              prior_var=np.array([[m["prior_var"]]]), **kw)
        ll = float(g.get(key, "all_loglikelihood_estimate")[-1])
        assert abs(out["loglikelihood_estimate"] - ll) <= 1e-9 * max(1.0, abs(ll)), (m, out["loglikelihood_estimate"], ll)
        ref = g.get(key, "mean_statistic") if m["pf"] != "filter" else g.get(key, "all_statistics")[-1]
        np.testing.assert_allclose(out["mean_statistic"], ref, rtol=1e-9, atol=1e-8, err_msg=str(m))
        # the stub consumed np.random exactly as the reference's loop does: N + T (N + N) doubles / normals
        rs = np.random.RandomState(m["seed"])
        rs.normal(size=m["N"])
        for _ in range(m["T"]):
            rs.random_sample(m["N"]); rs.normal(size=m["N"])
        assert np.random.random_sample() == rs.random_sample()
        n += 1
    assert n >= 10


def test_documented_structs_match_the_header(stub):
    import ctypes as C
    from sgmcmc_ssm_amd import _capi
    assert C.sizeof(stub["pfg_problem"]) == C.sizeof(_capi.Problem) == stub["lib"].pfg_struct_size(0)
    assert C.sizeof(stub["pfg_result"]) == C.sizeof(_capi.Result) == stub["lib"].pfg_struct_size(1)
    for (n1, t1), (n2, t2) in zip(stub["pfg_problem"]._fields_, _capi.Problem._fields_):
        assert n1 == n2 and getattr(stub["pfg_problem"], n1).offset == getattr(_capi.Problem, n2).offset, (n1, n2)
    for (n1, t1), (n2, t2) in zip(stub["pfg_result"]._fields_, _capi.Result._fields_):
        assert n1 == n2 and getattr(stub["pfg_result"], n1).offset == getattr(_capi.Result, n2).offset, (n1, n2)
