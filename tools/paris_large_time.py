"""Seed-compatible PaRIS above N = 1024: one launch per window (round 4) against one launch per timestep
(PFGRAD_PARIS_PER_TIMESTEP=1).  python tools/paris_large_time.py"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd"))
import numpy as np
from sgmcmc_ssm_amd.models.svm import SVMHelper, SVMParameters, generate_svm_data
from sgmcmc_ssm_amd.models.garch import GARCHHelper, GARCHParameters, generate_garch_data

np.random.seed(1)
p = SVMParameters(A=np.eye(1) * .95, Q=np.eye(1) * .5, R=np.eye(1) * .5)
y = generate_svm_data(T=300, parameters=p)["observations"]
helper = SVMHelper(n=1, m=1)
for N, T in ((2500, 24), (10000, 24), (10000, 300), (16384, 24)):
    for mode in ("per timestep", "one launch"):
        if mode == "per timestep":
            os.environ["PFGRAD_PARIS_PER_TIMESTEP"] = "1"
        else:
            os.environ.pop("PFGRAD_PARIS_PER_TIMESTEP", None)
        res = []
        for rep in range(3):
            np.random.seed(7)
            t0 = time.perf_counter()
            g = helper.pf_gradient_estimate(observations=y[:T], parameters=p, pf="paris", N=N)
            dt = time.perf_counter() - t0
            res.append(dt)
        nxt = np.random.random_sample()
        print("paris replay N={0:6d} T={1:4d} {2:13s}: {3:9.2f} ms per call = {4:7.3f} ms per timestep   A-gradient {5:.10f} next draw {6:.12f}".format(
            N, T, mode, min(res) * 1e3, min(res) * 1e3 / T, float(np.asarray(g["A"]).reshape(-1)[0]), nxt), flush=True)
