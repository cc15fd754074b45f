"""Timing of the PaRIS smoother (device RNG, default 16 accept-reject rounds): LDS-resident
(N <= 1024) and large-N (HBM-scratch) instantiations."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                                "stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd"))
from sgmcmc_ssm_amd import _capi
from sgmcmc_ssm_amd.particle_filters import make_problem

ctx = _capi.default_context(0)
rs = np.random.RandomState(0)
theta = np.array([0.95, 0.5 ** -0.5, 0.5 ** -0.5])
RS = [int(r) for r in os.environ.get("PARIS_R", "16").split(",")]
for dtype, R in [(d, r) for d in ("f64", "f32") for r in RS]:
    for N, T, B in ((1000, 100, 1), (1000, 100, 256), (4000, 24, 1), (10000, 24, 1), (10000, 24, 64), (10000, 24, 256)):
        x = np.zeros(T)
        for t in range(1, T):
            x[t] = 0.95 * x[t - 1] + rs.normal() * 0.5 ** 0.5
        y = np.exp(x / 2) * rs.normal(size=T) * 0.5 ** 0.5
        probs = [make_problem("svm", "prior", "paris", y, theta, N, prior_var=10.0, dtype=dtype, rng="device",
                              seed=3, stream=b, max_accept_reject=R) for b in range(B)]
        ctx.run_batch(probs)
        t0 = time.perf_counter()
        ctx.run_batch(probs)
        dt = time.perf_counter() - t0
        print("R={0:3d} ".format(R), end="")
        print("paris svm {0} N={1:6d} T={2:4d} B={3:4d}: {4:9.2f} ms/launch  {5:8.1f} us/timestep/window (amortised)".format(
            dtype, N, T, B, dt * 1e3, dt * 1e6 / (T * B)), flush=True)
