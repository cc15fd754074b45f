"""Kernel time of many small windows (N <= 256) per variant: python tools/small_n_time.py <N> <chains> [variants...]"""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 3 and sys.argv[3] != "--one":
    for v in sys.argv[3:]:
        env = dict(os.environ, PFGRAD_VARIANT=v)
        print(subprocess.run([sys.executable, __file__, sys.argv[1], sys.argv[2], "--one"], env=env, capture_output=True, text=True).stdout.strip())
    sys.exit(0)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd"))
import numpy as np, torch
import bench
from sgmcmc_ssm_amd.ensemble import ChainEnsemble
N, C = int(sys.argv[1]), int(sys.argv[2])
w = bench.config_workload("c1")
ens = ChainEnsemble(w["model"], w["y"], w["p0"], num_chains=C, N=N, kernel=w["kernel"], epsilon=w["epsilon"], prior=w["prior"],
                    subsequence_length=-1, buffer_length=-1, seed=3)
ens.step(2); ens.synchronize()
st = torch.cuda.current_stream(); ms = []
for _ in range(6):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st); ens.launch_pf(st); b.record(st); ens.launch_update(st); ens.synchronize(); ms.append(a.elapsed_time(b))
g, ll = ens.last_gradient_statistics()
print(json.dumps({"N": N, "chains": C, "variant": ens.ctx.last_variant(), "kernel_ms": float(np.median(ms)), "mean_ll": float(np.mean(ll))}))
