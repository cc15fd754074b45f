"""Per-phase cycle profile of the hot PF kernel from in-kernel s_memtime stamps.

Needs the diagnostic build (stamps between the phases of every timestep cost ~10 % wave cycles, so
it is a separate library):   python -m sgmcmc_ssm_amd._build stamps -DPFG_PHASE_STAMPS
Run:  PFGRAD_LIB=<csrc>/libpfgrad_stamps.so python tools/phase_profile.py [model] [chains]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd"))
import numpy as np, torch
import bench
from sgmcmc_ssm_amd.ensemble import ChainEnsemble

model = sys.argv[1] if len(sys.argv) > 1 else "svm"
C = int(sys.argv[2]) if len(sys.argv) > 2 else 3072
p0, y, prior, cfg = bench.make_workload(model)
ens = ChainEnsemble(model, y, p0, num_chains=C, N=1000, kernel=cfg["kernel"], epsilon=cfg["epsilon"], prior=prior,
                    subsequence_length=cfg["S"], buffer_length=cfg["B"], seed=2024)
ens.step(2); ens.synchronize()
ens.enable_stamps()
st = torch.cuda.current_stream()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(st); ens.launch_pf(st); b.record(st); ens.synchronize()
ms = a.elapsed_time(b)
ghz, cyc, phases = ens.kernel_clock()
names = ["A: local max + DPP max", "wait barrier 1", "B,C: exp, scans, partials", "wait barrier 2",
         "offsets, W, loglik, CDF write", "wait barrier 3", "E: draw words + search", "F: gather parents",
         "wait barrier 4", "G,H: normals, propose, weight, score, publish"]
out = {"model": model, "chains": C, "kernel_ms": ms, "in_kernel_clock_ghz": ghz, "workgroup_cycles_median": cyc,
       "variant": ens.ctx.last_variant()}
print(json.dumps(out))
if phases is None:
    print("no phase sums: not a -DPFG_PHASE_STAMPS build (PFGRAD_LIB?)")
else:
    tot = phases.sum()
    for n, v in zip(names, phases):
        print("{0:48s} {1:6.2f} %   {2:9.1f} cycles per wave-timestep".format(n, 100 * v / tot, v / (C * 4 * ens.T)))
    print("{0:48s} {1:9.1f} cycles per wave-timestep".format("total (4 waves per workgroup)", tot / (C * 4 * ens.T)))
