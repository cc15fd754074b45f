"""The README quick-start snippet, kept runnable (tools/readme_snippet.py is generated from README.md)."""
import sys, numpy as np
sys.path.insert(0, "stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd")
from sgmcmc_ssm_amd.models.svm import SVMSampler, SVMParameters, generate_svm_data
from sgmcmc_ssm_amd.ensemble import ChainEnsemble

p = SVMParameters(A=np.eye(1) * .95, Q=np.eye(1) * .5, R=np.eye(1) * .5)
y = generate_svm_data(T=1000, parameters=p)["observations"]

# drop-in, one chain, the reference's API; np.random.seed(s) reproduces the reference seed for seed
sampler = SVMSampler(n=1, m=1, observations=y, parameters=p.copy())
np.random.seed(0)
grad = sampler.noisy_gradient(kind="pf", pf="poyiadjis_N", N=1000)            # dict like parameters.var_dict
params = sampler.fit(iter_type="SGLD", num_iters=100, epsilon=0.1, subsequence_length=16, buffer_length=4,
                     kind="pf", pf_kwargs=dict(pf="poyiadjis_N", N=1000, rng="device"))   # device generator: fast

# 3072 independent chains resident on one MI355X (bench.py times 12288: ~290 k SGLD steps/s)
ens = ChainEnsemble("svm", y, p, num_chains=3072, N=1000, epsilon=0.1, seed=1)
samples = ens.run(20, thin=5)                                                 # [4, 3072, 3]

# the reference's ground-truth call of its bias experiments (svm_grad_compare.py:68-82): N = 10^6 particles, seed for seed
np.random.seed(4101)
g = sampler.noisy_gradient(kind="pf", pf="poyiadjis_N", N=1000000, subsequence_length=16, buffer_length=16)   # 0.25 s (reference: 38 s)

print(sorted(grad), samples.shape, params.theta(), {k: np.ravel(v) for k, v in g.items()})
