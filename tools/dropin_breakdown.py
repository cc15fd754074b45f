"""Where a drop-in SVMSampler.sample_sgld step (full T = 1000 sequence, N = 1000, rng='replay') spends its time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd"))
import numpy as np
from sgmcmc_ssm_amd import _capi, particle_filters as pf
from sgmcmc_ssm_amd.models.svm import SVMSampler, SVMParameters, generate_svm_data
np.random.seed(1)
p = SVMParameters(A=np.eye(1) * 0.95, Q=np.eye(1) * 0.5, R=np.eye(1) * 0.5)
y = generate_svm_data(T=1000, parameters=p)["observations"]
s = SVMSampler(n=1, m=1, observations=y, parameters=p.copy())
kw = dict(kind="pf", pf="poyiadjis_N", N=1000)
acc = {}
def timed(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
    setattr(obj, name, g)
timed(pf.speculation, "take", "speculation.take (join)")
timed(pf.speculation, "start", "speculation.start")
timed(_capi.Context, "run_batch", "run_batch")
timed(_capi, "legacy_streams", "legacy_streams (either thread)")
for _ in range(10):
    s.sample_sgld(epsilon=0.01, **kw); s.project_parameters()
acc.clear()
n = 100
t0 = time.perf_counter()
for _ in range(n):
    s.sample_sgld(epsilon=0.01, **kw); s.project_parameters()
tot = (time.perf_counter() - t0) / n * 1e3
print("%.3f ms per step; adopted %d discarded %d" % (tot, pf.speculation.adopted, pf.speculation.discarded))
for k, v in acc.items():
    print("  %-34s %.3f ms" % (k, v / n * 1e3))
