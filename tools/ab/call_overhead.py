import os, sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd")
import numpy as np
from sgmcmc_ssm_amd import _capi, particle_filters as pfm
ctx = _capi.default_context(0)
y = np.random.RandomState(0).normal(size=1000)
th = np.array([0.95, 0.5 ** -0.5, 0.5 ** -0.5])
for T in (1, 24, 100):
    q = pfm.make_problem("svm", "prior", "poyiadjis_N", y[:T], th, 1000, prior_var=10.0, rng="device", seed=1, stream=2)
    for _ in range(20): ctx.run_batch([q])
    ts, cs = [], []
    for _ in range(300):
        t0 = time.perf_counter(); ctx.run_batch([q]); ts.append(time.perf_counter() - t0); cs.append(ctx.last_call_seconds)
    print("T=%4d  python call %.1f us   C call %.1f us   variant %s" % (T, np.median(ts) * 1e6, np.median(cs) * 1e6, ctx.last_variant()))
