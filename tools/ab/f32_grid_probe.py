"""f32 state on the whole-GPU window against the one-workgroup kernels and f64 on the same REPLAY streams (probe)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd"))
import numpy as np
from sgmcmc_ssm_amd import _capi
from oracle import pf_oracle as po
ctx = _capi.default_context(0)
for model, kernel, theta, pf, lam in (("garch", "optimal", [0.0, 2.0, 2.0, 1.8], "nemeth", 0.9), ("garch", "prior", [0.0, 2.0, 2.0, 1.8], "nemeth", 0.9),
                                      ("lgssm", "optimal", [0.9, 1.0, 1.2, 1.0], "nemeth", 0.9), ("svm", "prior", [0.95, 1.4, 1.4], "nemeth", 0.9)):
    for N in (3000, 12000):
        T = 6
        rs = np.random.RandomState(5)
        y = rs.normal(size=T)
        z0, u, z = po.draw_streams(rs, N, T)
        q = dict(model=model, kernel=kernel, smoother="nemeth", stat="score", rng="replay", N=N, t1=1, tL=T - 1, lambduh=lam,
                 prior_mean=0.0, prior_var=1.5, y=y, weights=rs.uniform(1.0, 5.0, size=T - 2), theta=theta, z0=z0, u=u, z=z)
        res = {}
        for dtype in ("f64", "f32"):
            for var in ("", "grid"):
                if var: os.environ["PFGRAD_VARIANT"] = var
                else: os.environ.pop("PFGRAD_VARIANT", None)
                o = ctx.run_batch([dict(q, dtype=dtype)], want_trace=True)[0]
                res[(dtype, var)] = o
                print(model, kernel, N, dtype, var or "default", ctx.last_variant(), o["mean_stat"], o["loglik"], flush=True)
        os.environ.pop("PFGRAD_VARIANT", None)
        for var in ("", "grid"):
            a, b = res[("f64", var)], res[("f32", var)]
            fl = [int(np.sum(a["all_ancestors"][t] != b["all_ancestors"][t])) for t in range(T)]
            print("   flips f32 vs f64 per step", var or "default", fl)
        a, b = res[("f32", "")], res[("f32", "grid")]
        print("   flips f32 grid vs f32 default", [int(np.sum(a["all_ancestors"][t] != b["all_ancestors"][t])) for t in range(T)],
              "max |x| diff", float(np.max(np.abs(a["all_x_t"] - b["all_x_t"]))), "max |lw| diff", float(np.nanmax(np.abs(a["all_log_weights"] - b["all_log_weights"]))))
