"""One chain alone (SVM T=1000 N=1000, device generator): ms per step per forced variant."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd"))
import numpy as np, torch
from sgmcmc_ssm_amd.ensemble import ChainEnsemble
from sgmcmc_ssm_amd.models.svm import SVMParameters, generate_svm_data
p = SVMParameters(A=np.eye(1) * .95, Q=np.eye(1) * .5, R=np.eye(1) * .5)
np.random.seed(1)
y = generate_svm_data(T=1000, parameters=p)["observations"]
for v in sys.argv[1:]:
    os.environ["PFGRAD_VARIANT"] = v
    ens = ChainEnsemble("svm", y, p, num_chains=1, N=1000, epsilon=1e-3, seed=3)
    ens.step(2); ens.synchronize()
    t0 = time.perf_counter(); ens.step(10); ens.synchronize()
    print(v, "%.3f ms/step" % ((time.perf_counter() - t0) / 10 * 1e3), flush=True)
