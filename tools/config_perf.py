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
if which == "graph":
    # launch-bound regime: short buffered windows.  host windows (eager) vs device windows (eager) vs hipGraph replay
    p, gen = params("garch")
    np.random.seed(1)
    y = gen(T=1000, parameters=p)["observations"]
    for C in (1, 64, 4096):
        for mode, K in (("host", 0), ("device", 0), ("device", 16), ("device", 64)):
            ens = ChainEnsemble("garch", y, p, num_chains=C, N=1000, epsilon=1e-3, subsequence_length=16, buffer_length=4,
                                seed=3, window_sampling=mode)
            n = 256
            ens.run(64, thin=64, graph_steps=K); ens.synchronize()
            t0 = time.perf_counter()
            ens.run(n, thin=n, graph_steps=K); ens.synchronize()
            dt = (time.perf_counter() - t0) / n
            print(f"C3 garch S16 B4 C={C:5d} windows={mode:6s} graph_steps={K:3d}: {dt*1e6:9.1f} us/step  {C/dt:12.0f} steps/s", flush=True)
if which == "c4mem":
    run("C4 svm T=1000 N=4000 full", "svm", 1000, 4000, 256, steps=2, variant="mem1024")
if which == "mid":
    for N in (1500, 2000, 3000):
        for v in ("wg1024x4", "wg1024x4s", "mem1024"):
            try:
                run(f"svm T=300 N={N} {v}", "svm", 300, N, 256, steps=2, variant=v)
                run(f"svm T=300 N={N} {v} f32", "svm", 300, N, 256, steps=2, variant=v, dtype="f32")
            except Exception as e:
                print(N, v, "failed", e)
if which == "dropin":
    # the reference-compatible Sampler API (REPLAY: host MT19937 streams + H2D), one chain
    from sgmcmc_ssm_amd.models.svm import SVMSampler
    np.random.seed(1)
    p, gen = params("svm")
    y = gen(T=1000, parameters=p)["observations"]
    for kw, name in [(dict(subsequence_length=-1, buffer_length=-1), "full T=1000"),
                     (dict(subsequence_length=16, buffer_length=4), "S=16 B=4"),
                     (dict(subsequence_length=16, buffer_length=4, rng="device"), "S=16 B=4 device rng"),
                     (dict(subsequence_length=-1, buffer_length=-1, rng="device"), "full T=1000 device rng"),
                     (dict(subsequence_length=16, buffer_length=4, pf="paris"), "S=16 B=4 paris")]:
        s = SVMSampler(n=1, m=1, observations=y, parameters=p.copy())
        kw = dict(dict(kind="pf", pf="poyiadjis_N", N=1000), **kw)
        s.sample_sgld(epsilon=0.01, **kw); s.project_parameters()
        n = 20 if kw["subsequence_length"] == -1 else 200
        t0 = time.perf_counter()
        for _ in range(n):
            s.sample_sgld(epsilon=0.01, **kw); s.project_parameters()
        dt = (time.perf_counter() - t0) / n
        print(f"drop-in SVMSampler.sample_sgld N=1000 {name:28s}: {dt*1e3:8.3f} ms/step  {1/dt:8.1f} steps/s", flush=True)
if which == "dropin_fit":
    # fit() on the resident path (rng='device': one-chain ChainEnsemble, hipGraph replay) against the host loop
    from sgmcmc_ssm_amd.models.svm import SVMSampler
    np.random.seed(1)
    p, gen = params("svm")
    y = gen(T=1000, parameters=p)["observations"]
    for S, B, name in ((16, 4, "S=16 B=4"), (-1, -1, "full T=1000")):
        for resident in (False, True):
            s = SVMSampler(n=1, m=1, observations=y, parameters=p.copy())
            kw = dict(iter_type="SGLD", epsilon=0.01, subsequence_length=S, buffer_length=B, kind="pf",
                      pf_kwargs=dict(pf="poyiadjis_N", N=1000, rng="device", resident=resident))
            s.fit(num_iters=64, **kw)
            n = 2048 if S != -1 else 128
            t0 = time.perf_counter()
            s.fit(num_iters=n, **kw)
            dt = (time.perf_counter() - t0) / n
            print(f"drop-in SVMSampler.fit SGLD N=1000 {name:12s} rng=device {'resident' if resident else 'host loop':9s}: {dt*1e3:8.4f} ms/step  {1/dt:9.1f} steps/s", flush=True)
if which == "dropin_large":
    # BASELINE configs 4 and 5 through the drop-in Sampler API, seed-compatible (rng='replay': pf_mem_kernel) and device rng
    from sgmcmc_ssm_amd.models.svm import SVMSampler, SeqSVMSampler
    import bench
    np.random.seed(1)
    p, gen = params("svm")
    y = gen(T=1000, parameters=p)["observations"]
    for rng in ("replay", "device"):
        s = SVMSampler(n=1, m=1, observations=y, parameters=p.copy())
        kw = dict(kind="pf", pf="poyiadjis_N", N=4000, subsequence_length=-1, buffer_length=-1, rng=rng)
        s.sample_sgld(epsilon=0.01, **kw); s.project_parameters()
        n = 10
        t0 = time.perf_counter()
        for _ in range(n):
            s.sample_sgld(epsilon=0.01, **kw); s.project_parameters()
        dt = (time.perf_counter() - t0) / n
        print(f"drop-in SVMSampler.sample_sgld N=4000 full T=1000 rng={rng:7s}: {dt*1e3:8.3f} ms/step  {1/dt:8.1f} steps/s", flush=True)
    w = bench.config_workload("c5")
    for rng in ("replay", "device"):
        s = SeqSVMSampler(n=1, m=1, observations=w["y"], parameters=w["p0"].copy())
        kw = dict(kind="pf", pf="poyiadjis_N", N=10000, subsequence_length=16, buffer_length=4, num_sequences=1, rng=rng)
        s.sample_sgld(epsilon=0.001, **kw); s.project_parameters()
        n = 100
        t0 = time.perf_counter()
        for _ in range(n):
            s.sample_sgld(epsilon=0.001, **kw); s.project_parameters()
        dt = (time.perf_counter() - t0) / n
        print(f"drop-in SeqSVMSampler.sample_sgld EURUS N=10000 S=16 B=4 rng={rng:7s}: {dt*1e3:8.3f} ms/step  {1/dt:8.1f} steps/s", flush=True)
if which == "paris":
    from sgmcmc_ssm_amd import _capi
    from sgmcmc_ssm_amd.particle_filters import make_problem
    ctx = _capi.default_context(0)
    p, gen = params("svm"); np.random.seed(1); y = gen(T=1000, parameters=p)["observations"].reshape(-1)
    for B, T in [(1, 24), (256, 24), (1, 1000), (256, 1000)]:
        probs = [make_problem("svm", "prior", "paris", y[:T], p.theta(), 1000, prior_var=10.0, seed=1, stream=b) for b in range(B)]
        ctx.run_batch(probs[:1]); t0 = time.perf_counter(); ctx.run_batch(probs); dt = time.perf_counter() - t0
        print(f"paris svm N=1000 T={T} B={B}: {dt*1e3:.2f} ms  ({dt/T*1e6:.1f} us per timestep-batch)", flush=True)
if which == "parisR":
    from sgmcmc_ssm_amd import _capi
    from sgmcmc_ssm_amd.particle_filters import make_problem
    ctx = _capi.default_context(0)
    p, gen = params("svm"); np.random.seed(1); y = gen(T=1000, parameters=p)["observations"].reshape(-1)
    T = 200
    for R in (0, 1, 2, 4, 8, 16, 32, 64):
        for Nt in (1, 2):
            probs = [make_problem("svm", "prior", "paris", y[:T], p.theta(), 1000, prior_var=10.0, seed=1, stream=b, max_accept_reject=R, Ntilde=Nt) for b in range(4)]
            ctx.run_batch(probs[:1]); t0 = time.perf_counter(); ctx.run_batch(probs); dt = time.perf_counter() - t0
            print(f"paris R={R} Ntilde={Nt}: {dt/T*1e6:.1f} us per timestep", flush=True)
