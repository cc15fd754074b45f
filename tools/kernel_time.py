"""Kernel time of the bench workload (or another config) for the library PFGRAD_LIB selects.
usage: python tools/kernel_time.py [svm|garch] [chains] [reps]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd"))
import numpy as np, torch
import bench
from sgmcmc_ssm_amd.ensemble import ChainEnsemble
model = sys.argv[1] if len(sys.argv) > 1 else "svm"
C = int(sys.argv[2]) if len(sys.argv) > 2 else 3072
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
p0, y, prior, cfg = bench.make_workload(model)
ens = ChainEnsemble(model, y, p0, num_chains=C, N=1000, kernel=cfg["kernel"], epsilon=cfg["epsilon"], prior=prior,
                    subsequence_length=cfg["S"], buffer_length=cfg["B"], seed=2024)
ens.step(2); ens.synchronize()
st = torch.cuda.current_stream()
ms = []
for _ in range(reps):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st); ens.launch_pf(st); b.record(st); ens.launch_update(st); ens.synchronize()
    ms.append(a.elapsed_time(b))
g, ll = ens.last_gradient_statistics()
print(json.dumps({"lib": os.path.basename(os.environ.get("PFGRAD_LIB", "libpfgrad.so")), "model": model, "chains": C,
                  "variant": ens.ctx.last_variant(), "kernel_ms_median": float(np.median(ms)), "kernel_ms_min": float(np.min(ms)),
                  "mean_grad": np.mean(g, axis=0).round(4).tolist(), "mean_ll": float(np.mean(ll))}))
