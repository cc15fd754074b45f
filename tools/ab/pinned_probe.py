"""Does page-locking the replay stream buffers (pfg_host_register) pay?  One T = N = 1000 REPLAY window through
pfg_run_batch with plain and with registered u / z arrays; stages timed apart."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd"))
import numpy as np
import bench
from sgmcmc_ssm_amd import _capi, particle_filters as pfm

p0, y, prior, cfg = bench.make_workload("svm")
ctx = _capi.default_context(0)
N = T = 1000
def one(pin):
    pfm._stream_pool.clear()
    if not pin:
        pfm._PIN_MIN_BYTES = 1 << 62
    else:
        pfm._PIN_MIN_BYTES = 1 << 20
    rs = np.random.RandomState(5)
    ts = []
    for it in range(8):
        t0 = time.perf_counter()
        q = pfm.make_problem("svm", "prior", "poyiadjis_N", y, p0.theta(), N, prior_mean=0.0, prior_var=10.0, random_state=rs)
        t1 = time.perf_counter()
        out = ctx.run_batch([q])
        t2 = time.perf_counter()
        pfm._recycle_streams([q])
        ts.append((t1 - t0, t2 - t1))
    a = np.array(ts[2:]) * 1e3
    return a.mean(0), out[0]["mean_stat"]
for pin in (False, True, False, True):
    (gen, run), g = one(pin)
    print("pinned" if pin else "plain ", "generate %.2f ms   run_batch %.2f ms" % (gen, run), "pinned pairs:", len(pfm._pinned), g[:2])
