# A/B of a tagged library on the 1024-particle tile class of the whole-GPU window, all three models, 1 and 8 windows
cd /root/repo
L=/root/repo/stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd/csrc
for lib in libpfgrad.so libpfgrad_$1.so; do
  echo "== $lib"
  PFGRAD_LIB=$L/$lib timeout -k 10 200 python - <<'PY' 2>/dev/null
import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd")
import numpy as np, torch
from sgmcmc_ssm_amd.grid import ResidentWindows
T = 48
y = np.random.RandomState(0).randn(T) * 1.5
for model, theta, bpp in (("svm", [0.95, 1.414, 1.414], 80), ("garch", [0.0, 2.0, 2.0, 1.8], 112), ("lgssm", [0.9, 1.0, 1.2, 1.0], 96)):
    for N, B in ((100000, 1), (100000, 8), (400000, 1), (400000, 8)):
        rw = ResidentWindows(model, y, np.tile(theta, (B, 1)), N, t1=16, tL=32, prior_var=5.0, seed=3)
        st = torch.cuda.current_stream()
        rw.launch(); torch.cuda.synchronize()
        ms = []
        for _ in range(7):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(st); rw.launch(); b.record(st); torch.cuda.synchronize()
            ms.append(a.elapsed_time(b))
        k = float(np.median(ms))
        print("%-5s N=%7d B=%d  %.2f us/step  frac %.3f  %s" % (model, N, B, k * 1e3 / T, B * T * N * bpp / (k * 1e-3) / 8e12, rw.ctx.last_variant()), flush=True)
PY
done
