import sys, time, os
sys.path.insert(0, "stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd")
import numpy as np
from sgmcmc_ssm_amd import _capi
ctx = _capi.default_context(0)
rs = np.random.RandomState(0)
T, N = 1000, 1000
y = rs.normal(size=T)
def run(B, dtype, model="svm", kernel="prior", N=N, T=T):
    th = {"svm":[0.95,1.4,1.4],"garch":[0.0,2.0,2.0,1.8],"lgssm":[0.9,1.0,1.2,1.0]}[model]
    probs = [dict(model=model, kernel=kernel, dtype=dtype, rng="device", N=N, y=y[:T], theta=th, seed=1, stream=b,
                  prior_var=1.0) for b in range(B)]
    ctx.run_batch(probs[:1])
    t = time.time(); o = ctx.run_batch(probs); dt = time.time()-t
    print(f"{model}/{kernel} {dtype} N={N} T={T} B={B}: {dt*1e3:.2f} ms -> {B/dt:.1f} grads/s, {dt/T*1e6:.2f} us/timestep(batch) mean_stat0={o[0]['mean_stat']}", flush=True)
for dtype in ("f64","f32"):
    for B in (1, 64, 256, 512, 768, 1024, 2048):
        run(B, dtype)
run(256,"f64","garch","optimal"); run(256,"f64","lgssm","optimal")
run(1,"f64",N=4000); run(256,"f64",N=4000)
run(1,"f64",N=100,T=200,model="lgssm",kernel="optimal")
