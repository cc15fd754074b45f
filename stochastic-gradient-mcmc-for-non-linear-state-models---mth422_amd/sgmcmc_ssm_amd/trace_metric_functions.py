"""Metrics on whole parameter traces: the kernel Stein discrepancy evaluation.

Counterpart of the reference's `sgmcmc_ssm/trace_metric_functions.py` (`IMQ_KSD`,
`compute_KSD`) and of the gradient loop of `do_eval_ksd`
(nonlinear_ssm_pf_experiment_scripts/svm/driver.py:1006-1027), which calls
`sampler.noisy_gradient(kind='pf', pf='poyiadjis_N', N=10000, ...)` once per stored parameter.
Here every stored parameter is one workgroup of a single launch
(`SGMCMCSampler.noisy_gradient_trace`), and the O(K^2) IMQ pass is a HIP kernel."""
import numpy as np

from . import _capi


def IMQ_KSD(x, gradlogp, c=1, beta=0.5, max_block_size=1000, tqdm_out=None, ctx=None):
    """Inverse-multiquadric kernel Stein discrepancy.
    x, gradlogp: (num_points, d) arrays; returns a float.  (`max_block_size`, `tqdm_out` are the
    reference's host-memory blocking knobs and are ignored.)"""
    x, gradlogp = np.asarray(x, dtype=float), np.asarray(gradlogp, dtype=float)
    if x.shape != gradlogp.shape:
        raise ValueError("x and gradlogp dimensions do not match")
    ctx = ctx or _capi.default_context()
    return ctx.imq_ksd(x, gradlogp, c=c, beta=beta)


def compute_KSD(param_list, grad_list, variables=None, **kwargs):
    """{variable: IMQ_KSD} over a trace.  param_list: list of Parameters; grad_list: per
    parameter, a list of gradient arrays ordered like `variables`."""
    res = {}
    if variables is None:
        return res
    for ii, var in enumerate(variables):
        if not hasattr(param_list[0], var):
            continue
        x = np.array([np.asarray(getattr(p, var)).flatten() for p in param_list])
        g = np.array([np.asarray(grad[ii]).flatten() for grad in grad_list])
        if x.ndim == 1:
            x = x.reshape(-1, 1)
        if g.ndim == 1:
            g = g.reshape(-1, 1)
        res[var] = IMQ_KSD(x, g, **kwargs)
    return res
