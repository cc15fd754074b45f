"""Ten N = 10^6 windows as ONE batch on one stream against 2 x 5 / 5 x 2 sub-batches on their own streams (probe)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd"))
import numpy as np, torch
from sgmcmc_ssm_amd.grid import ResidentWindows
N, T, B = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000, 48, 10
y = np.random.RandomState(0).randn(T) * 1.5
th = np.tile([0.95, 1.414, 1.414], (B, 1))
for S in (1, 2, 5, 10):
    per = B // S
    rws = [ResidentWindows("svm", y, th[:per], N, t1=16, tL=32, prior_var=5.0, seed=3, stream0=s * per) for s in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    def run(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            for rw, st in zip(rws, streams):
                rw.launch(st)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps
    run(2)
    dt = min(run(5), run(5))
    print("N = {0}: {1} stream(s) x {2} windows: {3:.3f} ms per {4} windows = {5:.1f} windows/s  frac {6:.3f}".format(
        N, S, per, dt * 1e3, B, B / dt, B * T * N * 80 / dt / 8e12), flush=True)
    del rws
