"""PCIe-inclusive rate of the C-ABI boundary that takes HOST buffers (pfg_run_batch): one call = pack + H2D of
every window's inputs, the launch, D2H of the results.  bench.py's `value` has its inputs resident in HBM; this
is the figure beside it.
usage: python tools/pcie_rate.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd"))
import numpy as np
import bench
from sgmcmc_ssm_amd import _capi, particle_filters as pfm

p0, y, prior, cfg = bench.make_workload("svm")
ctx = _capi.default_context(0)
N = 1000
theta = p0.theta()

def timed(probs, reps=5):
    """median seconds of (the Python call, the C call inside it)"""
    ctx.run_batch(probs)
    t, c = [], []
    for _ in range(reps):
        t0 = time.perf_counter(); ctx.run_batch(probs); t.append(time.perf_counter() - t0); c.append(ctx.last_call_seconds)
    return float(np.median(t)), float(np.median(c))

for B in (3072, 12288):
    probs = [pfm.make_problem("svm", "prior", "poyiadjis_N", y, theta, N, prior_mean=0.0, prior_var=10.0, rng="device",
                              seed=7, stream=b) for b in range(B)]
    s, c = timed(probs)
    h2d = B * (len(y) + 4) * 8 / 1e6
    print("device generator, %5d windows per pfg_run_batch: C call %.2f ms = %.1f k PF windows/s (inputs %.1f MB on the host, the shared series staged once; D2H %.1f KB); with the ctypes marshalling of %d problem dicts %.2f ms = %.1f k/s"
          % (B, c * 1e3, B / c / 1e3, h2d, B * 8 * 8 / 1e3, B, s * 1e3, B / s / 1e3))
rs = np.random.RandomState(3)
for B in (1, 8, 32):
    probs = [pfm.make_problem("svm", "prior", "poyiadjis_N", y, theta, N, prior_mean=0.0, prior_var=10.0, random_state=rs)
             for b in range(B)]
    s, c = timed(probs)
    print("REPLAY (host streams), %3d windows per pfg_run_batch: C call %.2f ms = %.0f PF windows/s   (H2D %.0f MB: %.1f GB/s if it were all copy)"
          % (B, c * 1e3, B / c, B * 16.0, B * 16e-3 / c))
