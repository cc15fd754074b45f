"""CPU oracle for the kernel Stein discrepancy (TEST INFRASTRUCTURE ONLY).

NumPy restatement of `IMQ_KSD` / `compute_KSD` of the reference
(`sgmcmc_ssm/trace_metric_functions.py:20-112`), written from its behaviour.  Pinned against
values produced by the reference itself (tests/golden/ksd.npz, tests/golden/make_golden.py)."""
import numpy as np


def imq_ksd(x, gradlogp, c=1.0, beta=0.5):
    """sqrt( sum_{i,j} k0(x_i, x_j) ) / K for the Stein kernel of the inverse multiquadric
    k(x,y) = (c^2 + |x-y|^2)^(-beta):
      k0 = g_i.g_j k  + (g_i - g_j).(x_j - x_i)... expanded as in trace_metric_functions.py:66-75."""
    x = np.asarray(x, dtype=float)
    g = np.asarray(gradlogp, dtype=float)
    if x.shape != g.shape:
        raise ValueError("x and gradlogp dimensions do not match")
    K, d = x.shape
    total = 0.0
    for i in range(K):                      # row-wise to bound memory
        diff = x[i] - x                     # x0 - x1 with x0 = x_i
        diff2 = np.sum(diff ** 2, axis=1)
        base = diff2 + c ** 2
        base_beta = base ** -beta
        base_beta1 = base_beta / base
        coeffgrad = -2.0 * beta * base_beta1
        kterm = np.sum(g[i] * g, axis=1) * base_beta
        g0 = np.sum(g[i] * -diff, axis=1) * coeffgrad
        g1 = np.sum(g * diff, axis=1) * coeffgrad
        g01 = (-d + 2 * (beta + 1) * diff2 / base) * coeffgrad
        total += np.sum(kterm + g0 + g1 + g01)
    return np.sqrt(total) / K
