"""Timing of the whole-GPU window (tools; GPU box): device generator resident, and the drop-in REPLAY call.
usage: python tools/grid_time.py [N ...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd"))
import numpy as np
import torch
from sgmcmc_ssm_amd.grid import ResidentWindows
from sgmcmc_ssm_amd import _capi, particle_filters as pfm

Ns = [int(a) for a in sys.argv[1:]] or [100000, 1000000, 4000000]
T = 48
rs = np.random.RandomState(0)
y = rs.randn(T) * 1.5
out = []
for model, theta, bpp in ((("svm", [0.95, 1.414, 1.414], 80), ("garch", [0.0, 2.0, 2.0, 1.8], 112), ("lgssm", [0.9, 1.0, 1.2, 1.0], 96)) if os.environ.get("GRID_TIME_DEVICE", "1") == "1" else ()):
    for N in Ns:
        for B in (1, 4):
            if B > 1 and (model != "svm" or N > 1000000):
                continue
            rw = ResidentWindows(model, y, np.tile(theta, (B, 1)), N, t1=16, tL=32, prior_var=5.0, seed=3)
            st = torch.cuda.current_stream()
            rw.launch(); torch.cuda.synchronize()
            ms = []
            for _ in range(5):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(st); rw.launch(); b.record(st); torch.cuda.synchronize()
                ms.append(a.elapsed_time(b))
            k = float(np.median(ms))
            g, ll = rw.results()
            rec = dict(model=model, N=N, B=B, T=T, window_ms=k, us_per_step=k * 1e3 / T, alg_GBps=B * T * N * bpp / (k * 1e-3) / 1e9,
                       frac_of_8TBps=B * T * N * bpp / (k * 1e-3) / 8e12, grad=g[0].tolist(), loglik=float(ll[0]))
            print(json.dumps(rec), flush=True)
            out.append(rec)
            del rw
# drop-in REPLAY call at N = 10^6 (host stream generation + staging + kernels)
if os.environ.get("GRID_TIME_REPLAY", "1") == "1":
    ctx = _capi.default_context(0)
    for N in [n for n in Ns if n <= 1000000]:
        for rep in range(3):
            np.random.seed(5)
            t0 = time.perf_counter()
            q = pfm.make_problem("svm", "prior", "poyiadjis_N", y, [0.95, 1.414, 1.414], N, t1=16, tL=32, prior_var=5.0, rng="replay")
            t1 = time.perf_counter()
            o = ctx.run_batch([q])[0]
            t2 = time.perf_counter()
            pfm._recycle_streams([q])
            print(json.dumps(dict(kind="replay_dropin", N=N, rep=rep, make_problem_s=t1 - t0, run_batch_s=t2 - t1, c_call_s=ctx.last_call_seconds,
                                  grad=o["mean_stat"].tolist(), loglik=o["loglik"])), flush=True)
