"""PF-kernel time of one BASELINE config (bench.py's workloads) for the library / variant the environment selects
(PFGRAD_LIB, PFGRAD_VARIANT).  usage: python tools/cfg_time.py <c1..c5> [chains] [reps]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd"))
import numpy as np, torch
import bench
from sgmcmc_ssm_amd.ensemble import ChainEnsemble
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
w = bench.config_workload(cfg)
C = int(sys.argv[2]) if len(sys.argv) > 2 and int(sys.argv[2]) > 0 else w["chains"]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else int(os.environ.get("CFG_TIME_REPS", "6"))
ens = ChainEnsemble(w["model"], w["y"], w["p0"], num_chains=C, N=w["N"], kernel=w["kernel"], epsilon=w["epsilon"], prior=w["prior"],
                    subsequence_length=w["S"], buffer_length=w["B"], seed=2024,
                    window_sampling=("device" if w["S"] != -1 and not isinstance(w["y"], list) else "host"))
ens.step(2); ens.synchronize()
st = torch.cuda.current_stream()
ms = []
for _ in range(reps):
    if ens.window_sampling == "device":
        ens.launch_windows(st)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st); ens.launch_pf(st); b.record(st); ens.launch_update(st); ens.synchronize()
    ms.append(a.elapsed_time(b))
g, ll = ens.last_gradient_statistics()
k = float(np.median(ms))
print(json.dumps({"lib": os.path.basename(os.environ.get("PFGRAD_LIB", "libpfgrad.so")), "config": cfg, "chains": C,
                  "variant": ens.ctx.last_variant(), "kernel_ms_median": round(k, 4), "kernel_ms_min": round(float(np.min(ms)), 4),
                  "steps_per_s": round(C / (k * 1e-3), 1),
                  "mean_grad": np.mean(g, axis=0).round(4).tolist(), "mean_ll": round(float(np.mean(ll)), 4)}), flush=True)
