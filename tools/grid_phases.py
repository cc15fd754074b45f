"""Where one launch of the whole-GPU window's timestep spends its time: s_memtime stamps of the last tile's thread 0
(diagnostic build: python -m sgmcmc_ssm_amd._build gstamps -DPFG_GRID_STAMPS; run with PFGRAD_LIB=.../libpfgrad_gstamps.so)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd"))
import numpy as np
import torch
from sgmcmc_ssm_amd.grid import ResidentWindows
from sgmcmc_ssm_amd import _capi

names = ["P1 loads+spacing scans", "P2 scale+scan", "P3 pw", "P4 tile search+normals", "first scan load", "parent tiles (search)", "gather+propose+write", "epilogue"]
order = [0, 1, 2, 3, 4, 5, 8, 6, 7]
for N in [int(a) for a in sys.argv[1:]] or [100000, 1000000]:
    T = 12
    rw = ResidentWindows("svm", np.random.RandomState(0).randn(T), np.array([[0.95, 1.414, 1.414]]), N, prior_var=5.0, seed=3)
    stamps = torch.zeros(_capi.STAMP_WORDS, dtype=torch.int64, device=rw.device)
    rw._desc["stamps"] = stamps.data_ptr()
    rw.desc_dev.copy_(torch.from_numpy(rw._desc.view(np.uint8).reshape(rw.B, -1)))
    for _ in range(3):
        rw.launch()
    torch.cuda.synchronize()
    s = stamps.cpu().numpy()[4:16].astype(np.float64)[order]
    d = np.diff(s)
    print("N = {0}: last tile, last timestep, s_memtime ticks (100 MHz? see DESIGN 6: a tick is a shader cycle)".format(N))
    for n, x in zip(names, d):
        print("   {0:28s} {1:9.0f}".format(n, x))
    print("   {0:28s} {1:9.0f}".format("total", s[-1] - s[0]))
