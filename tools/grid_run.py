"""Minimal runner for profiling the whole-GPU window: python3 tools/grid_run.py MODEL N B REPS [T]  (device generator, resident)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd"))
import numpy as np
import torch
from sgmcmc_ssm_amd.grid import ResidentWindows
model, N, B, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
T = int(sys.argv[5]) if len(sys.argv) > 5 else 48
theta = {"svm": [0.95, 1.414, 1.414], "garch": [0.0, 2.0, 2.0, 1.8], "lgssm": [0.9, 1.0, 1.2, 1.0]}[model]
y = np.random.RandomState(0).randn(T) * 1.5
rw = ResidentWindows(model, y, np.tile(theta, (B, 1)), N, t1=T // 3, tL=2 * T // 3, prior_var=5.0, seed=3)
for _ in range(reps):
    rw.launch()
torch.cuda.synchronize()
print(rw.results()[1][:2])
