"""sgmcmc_ssm_amd: MI355X-native particle-filter gradient path of sgmcmc_ssm.

Mirrors the reference's Python API for this path (Parameters / Prior / Helper / Sampler for
the SVM, GARCH and 1-D LGSSM models) on top of libpfgrad.so (hand-written HIP, gfx950)."""
__version__ = "0.1.0"
