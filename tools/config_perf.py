"""Kernel time / throughput of the BASELINE.json configs through the resident ensemble."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd"))
import numpy as np, torch
from sgmcmc_ssm_amd.ensemble import ChainEnsemble
from sgmcmc_ssm_amd.models.svm import SVMParameters, generate_svm_data
from sgmcmc_ssm_amd.models.garch import GARCHParameters, generate_garch_data
from sgmcmc_ssm_amd.models.lgssm import LGSSMParameters, generate_lgssm_data

def params(model):
    if model == "svm": return SVMParameters(A=np.eye(1)*.95, Q=np.eye(1)*.5, R=np.eye(1)*.5), generate_svm_data
    if model == "lgssm": return LGSSMParameters(A=np.eye(1)*.9, C=np.eye(1), Q=np.eye(1)*.7, R=np.eye(1)), generate_lgssm_data
    lm, lp, ll = GARCHParameters.convert_alpha_beta_gamma(.1, .8, .05)
    return GARCHParameters(log_mu=lm, logit_phi=lp, logit_lambduh=ll, LRinv=np.eye(1)*.3**-.5), generate_garch_data

def run(name, model, T, N, C, S=-1, B=-1, dtype="f64", steps=4, variant=None):
    if variant: os.environ["PFGRAD_VARIANT"] = variant
    else: os.environ.pop("PFGRAD_VARIANT", None)
    p, gen = params(model)
    np.random.seed(1)
    y = gen(T=T, parameters=p)["observations"]
    ens = ChainEnsemble(model, y, p, num_chains=C, N=N, epsilon=1e-3, subsequence_length=S, buffer_length=B, dtype=dtype, seed=3)
    ens.step(1); ens.synchronize()
    st = torch.cuda.current_stream()
    evs = []
    t0 = time.perf_counter()
    for _ in range(steps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if ens._set_windows():
            ens.desc_dev.copy_(torch.from_numpy(ens._desc.view(np.uint8).reshape(ens.C, -1)), non_blocking=True)
        a.record(st); ens.launch_pf(st); b.record(st); ens.launch_update(st); ens.steps_done += 1
        evs.append((a, b))
    ens.synchronize()
    wall = (time.perf_counter() - t0) / steps
    ms = np.mean([a.elapsed_time(b) for a, b in evs])
    Tw = T if S == -1 else S + 2 * B
    var = ens.ctx.variant_name(model, ens.kernel, dtype, "device", N)
    print(f"{name:34s} {var:10s} C={C:5d} kernel {ms:8.3f} ms  wall/step {wall*1e3:8.3f} ms  {C/wall:10.0f} steps/s  {ms*1e3/Tw:7.2f} us/PF-timestep (batch)", flush=True)

which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "c1"): run("C1 lgssm T=200 N=100 full", "lgssm", 200, 100, 1); run("C1 lgssm T=200 N=100 full", "lgssm", 200, 100, 2048)
if which in ("all", "c2"): run("C2 svm T=1000 N=1000 full", "svm", 1000, 1000, 1); run("C2 svm T=1000 N=1000 full", "svm", 1000, 1000, 3072)
if which in ("all", "c3"): run("C3 garch T=1000 N=1000 S16 B4", "garch", 1000, 1000, 1, 16, 4); run("C3 garch T=1000 N=1000 S16 B4", "garch", 1000, 1000, 4096, 16, 4)
if which in ("all", "c4"):
    run("C4 svm T=1000 N=4000 full", "svm", 1000, 4000, 1, steps=2); run("C4 svm T=1000 N=4000 full", "svm", 1000, 4000, 256, steps=2)
    run("C4 svm T=1000 N=4000 full", "svm", 1000, 4000, 1, steps=2, variant="mem1024"); run("C4 svm T=1000 N=4000 full", "svm", 1000, 4000, 256, steps=2, variant="mem1024")
if which in ("all", "c5"): run("C5 svm T=126 N=10000 S16 B4", "svm", 126, 10000, 1, 16, 4); run("C5 svm T=126 N=10000 S16 B4", "svm", 126, 10000, 512, 16, 4)
if which in ("all", "f32"):
    run("C2 svm f32", "svm", 1000, 1000, 3072, dtype="f32"); run("C4 svm N=4000 f32", "svm", 1000, 4000, 256, dtype="f32", steps=2)
if which == "mid":
    for N in (1500, 2000, 3000):
        for v in ("wg1024x4", "wg1024x4s", "mem1024"):
            try:
                run(f"svm T=300 N={N} {v}", "svm", 300, N, 256, steps=2, variant=v)
                run(f"svm T=300 N={N} {v} f32", "svm", 300, N, 256, steps=2, variant=v, dtype="f32")
            except Exception as e:
                print(N, v, "failed", e)
