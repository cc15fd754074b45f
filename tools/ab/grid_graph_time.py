"""ResidentWindows: eager launches (T + 3 host launches per repetition) against one hipGraph launch per repetition(s)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd"))
import numpy as np, torch
from sgmcmc_ssm_amd.grid import ResidentWindows
T = 48
y = np.random.RandomState(0).randn(T) * 1.5
for N, B in ((20000, 1), (100000, 1), (100000, 4), (1000000, 1), (1000000, 10)):
    rw = ResidentWindows("svm", y, np.tile([0.95, 1.414, 1.414], (B, 1)), N, t1=16, tL=32, prior_var=5.0, seed=3)
    def wall(f, reps):
        f(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): f()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3
    e = min(wall(rw.launch, 20), wall(rw.launch, 20))
    g1 = min(wall(lambda: rw.launch_graph(1), 20), wall(lambda: rw.launch_graph(1), 20))
    g4 = min(wall(lambda: rw.launch_graph(4), 5), wall(lambda: rw.launch_graph(4), 5)) / 4
    print("N = {0:8d} B = {1:2d}: eager {2:.3f} ms per repetition | graph(1) {3:.3f} | graph(4) {4:.3f}".format(N, B, e, g1, g4), flush=True)
