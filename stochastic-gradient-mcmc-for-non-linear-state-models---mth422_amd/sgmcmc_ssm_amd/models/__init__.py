"""Model families on the particle-filter gradient path: stochastic volatility (svm), GARCH
with observation noise (garch) and the 1-D linear-Gaussian SSM (lgssm).  Each sub-module
exports the names the reference's `sgmcmc_ssm.models.<model>` package exports for this path."""
