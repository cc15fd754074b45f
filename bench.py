#!/usr/bin/env python
"""bench.py -- SGLD steps/sec of the particle-filter gradient path on MI355X.

Workload (BASELINE.json configs[1]): SVM synthetic series, T=1000, N=1000 particles,
SGLD on the full sequence (S=-1), pf='poyiadjis_N', bootstrap kernel, epsilon=0.1,
prior variance 100 (nonlinear_ssm_pf_experiment_scripts/svm/{demo_setup.py:65-79,driver.py:54}).

One bench "step" = one SGLD step (sample_sgld + project_parameters, what evaluator.py:343-347
times) of EVERY chain on the GPU: one particle-filter launch (one workgroup per chain, the
whole T-loop inside) + one update launch.  `value` = chain-steps per second summed over all
GPUs (chains are independent; weak scaling: --chains-per-gpu is fixed as N grows).
Everything is resident in HBM when the timed region starts.

The JSON line also carries
  roofline      algorithmic bytes/launch (2*(n+1+h)*8 B per particle-timestep,
                fp64) / mean PF-kernel duration (HIP events on the launch stream), vs 8 TB/s
  cpu_baseline  the CPU oracle (NumPy restatement of the reference, bit-identical to it) timed
                on this box's host, 1 core, on a bounded sample of the same workload
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd")
for p in (ROOT, PKG_DIR):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

T_SERIES, N_PART = 1000, 1000
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def make_workload(model):
    if model == "svm":
        from sgmcmc_ssm_amd.models.svm import SVMParameters, SVMPrior, generate_svm_data
        p = SVMParameters(A=np.eye(1) * 0.95, Q=np.eye(1) * 0.5, R=np.eye(1) * 0.5)
        np.random.seed(12345)
        data = generate_svm_data(T=T_SERIES, parameters=p)
        prior = SVMPrior.generate_default_prior(var=100.0, n=1, m=1)
        return p, data["observations"], prior, dict(epsilon=0.1, S=-1, B=-1, kernel="prior", n=1, h=3)
    if model == "garch":
        from sgmcmc_ssm_amd.models.garch import GARCHParameters, GARCHPrior, generate_garch_data
        lm, lp, ll = GARCHParameters.convert_alpha_beta_gamma(0.1, 0.8, 0.05)
        p = GARCHParameters(log_mu=lm, logit_phi=lp, logit_lambduh=ll, LRinv=np.eye(1) * 0.3 ** -0.5)
        np.random.seed(222)
        data = generate_garch_data(T=T_SERIES, parameters=p)
        prior = GARCHPrior.generate_default_prior(var=1.0, n=1, m=1)
        return p, data["observations"], prior, dict(epsilon=0.01, S=16, B=4, kernel="optimal", n=2, h=4)
    raise ValueError(model)


def grad_error_vs_reference():
    """The second half of BASELINE.json's metric: gradient L2 error vs the reference on identical
    seeds.  Runs the HIP path (REPLAY generator, fp64) on the reference's own known-answer case
    committed under tests/golden (SVM T=1000 N=1000, np.random.seed(99): the SURVEY 8c vector,
    produced by the reference itself) and returns the errors.  No oracle involved."""
    from sgmcmc_ssm_amd import _capi
    path = os.path.join(ROOT, "tests", "golden", "pf_window.npz")
    if not os.path.exists(path):
        return None
    g = np.load(path)
    meta = [m for m in json.loads(str(g["meta"])) if m["key"] == "w0"][0]
    N, T = meta["N"], meta["T"]
    rs = np.random.RandomState(meta["seed"])      # the reference's consumption order of the legacy stream
    z0 = rs.normal(size=N)
    u, z = np.empty((T, N)), np.empty((T, N))
    for t in range(T):
        u[t] = rs.random_sample(N)
        z[t] = rs.normal(size=N)
    q = dict(model="svm", kernel="prior", smoother="nemeth", stat="score", dtype="f64", rng="replay", N=N, t1=0,
             tL=T, lambduh=1.0, prior_mean=meta["prior_mean"], prior_var=meta["prior_var"], y=g["w0/y"],
             theta=g["w0/theta"], z0=z0, u=u, z=z)
    o = _capi.default_context(torch_device_index()).run_batch([q])[0]
    ref = g["w0/mean_statistic"]
    return {"grad_l2_err_vs_ref": float(np.linalg.norm(o["mean_stat"] - ref)),
            "grad_l2_ref_norm": float(np.linalg.norm(ref)),
            "loglik_abs_err_vs_ref": float(abs(o["loglik"] - float(g["w0/loglikelihood_estimate"]))),
            "case": "SVM T=1000 N=1000, np.random.seed(99), reference fixture tests/golden/pf_window.npz:w0, REPLAY fp64"}


def torch_device_index():
    import torch
    return torch.cuda.current_device()


def cpu_baseline(model, p0, y, prior, cfg, budget_s=12.0):
    """Reference CPU path: SGLD steps/s of ONE chain with the NumPy oracle (bit-identical to the
    reference's arithmetic), single thread.  Bounded: >= 3 steps, about `budget_s` seconds."""
    from oracle import pf_oracle as po
    from sgmcmc_ssm_amd.sgmcmc_sampler import random_subsequence_and_weights
    params = p0.copy()
    T = y.shape[0]
    eps = cfg["epsilon"]
    names = po.SCORE_NAMES[model]
    rng = np.random.RandomState(0)

    def one_step():
        if cfg["S"] == -1:
            lo, hi, t1, tL, w = 0, T, 0, T, None
        else:
            np.random.seed(rng.randint(2 ** 31))
            s, e, w = random_subsequence_and_weights(cfg["S"], T)
            lo, hi = max(0, s - cfg["B"]), min(T, e + cfg["B"])
            t1, tL = s - lo, e - lo
        if model == "garch":
            pm, pv = po.garch_prior_x(params.theta())
        else:
            pm, pv = 0.0, 10.0
        g = po.pf_gradient_estimate(model, params.theta(), y[lo:hi], N_PART, rng=rng, kernel=cfg["kernel"],
                                    pf="poyiadjis_N", t1=t1, tL=tL, weights=w, prior_mean=pm,
                                    prior_var=float(np.asarray(pv).reshape(-1)[0]))
        gp = prior.grad_logprior(params)
        for var in params.var_dict:
            delta = (gp[var] + g[var]) / T
            noise = rng.normal(loc=0, scale=np.sqrt(1.0 / T), size=params.var_dict[var].shape)
            params.var_dict[var] += eps * delta + np.sqrt(2.0 * eps) * noise
        params.project_parameters()

    one_step()                       # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        one_step()
        n += 1
        el = time.perf_counter() - t0
        if n >= 3 and el >= budget_s:
            break
        if el > 3 * budget_s:
            break
    return dict(value=n / el, unit="SGLD steps/s", cores=1, kind="port",
                sample="{0} full SGLD steps of one chain ({1:.1f} s), NumPy oracle, 1 thread; "
                       "host has {2} cores".format(n, el, os.cpu_count()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--chains-per-gpu", type=int, default=3072)
    ap.add_argument("--model", default="svm", choices=["svm", "garch"])
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-single-chain", action="store_true",
                    help="skip the one-chain-alone latency measurement (profiling runs: keeps every "
                         "launch of the PF kernel the same size, so rocprofv3's per-kernel average is the launch time)")
    ap.add_argument("--cpu-budget", type=float, default=12.0)
    args = ap.parse_args()

    os.environ.setdefault("OMP_NUM_THREADS", "1")
    import torch
    from sgmcmc_ssm_amd import distributed, _capi
    from sgmcmc_ssm_amd.ensemble import ChainEnsemble

    rank, world, local_rank = distributed.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus {0} but WORLD_SIZE={1}".format(args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    dev_index = local_rank % torch.cuda.device_count()     # 1:1 on a full node; folds ranks when rehearsing on one GPU
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    p0, y, prior, cfg = make_workload(args.model)
    C = args.chains_per_gpu
    lo, _ = distributed.chain_range(rank, C)
    ens = ChainEnsemble(args.model, y, p0, num_chains=C, N=N_PART, pf="poyiadjis_N", kernel=cfg["kernel"],
                        epsilon=cfg["epsilon"], prior=prior, subsequence_length=cfg["S"],
                        buffer_length=cfg["B"], dtype=args.dtype, seed=2024, chain_offset=lo, device=dev_index)

    for _ in range(args.warmup):
        ens.step(1)
    torch.cuda.synchronize(dev)
    distributed.barrier()
    torch.cuda.synchronize(dev)

    # K timed steps; HIP events (torch.cuda.Event on the launch stream) bracket each PF launch
    stream = torch.cuda.current_stream(dev)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        if ens.steps_done > 0 and ens._set_windows():
            ens.desc_dev.copy_(torch.from_numpy(ens._desc.view(np.uint8).reshape(ens.C, -1)), non_blocking=True)
        ev[k][0].record(stream)
        ens.launch_pf(stream)
        ev[k][1].record(stream)
        ens.launch_update(stream)
        ens.steps_done += 1
    torch.cuda.synchronize(dev)
    distributed.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    elapsed = distributed.max_over_ranks(elapsed, device=dev)

    samples = ens.gather_samples()            # the one collective (RCCL all_gather), outside the timing
    torch.cuda.synchronize(dev)
    theta = samples.cpu().numpy()
    if not np.all(np.isfinite(theta)):
        raise SystemExit("non-finite parameters after the run")

    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    # latency of ONE chain alone on the GPU (the reference's unit: one chain, one step at a time)
    single = None
    if rank == 0 and not args.no_single_chain:
        one = ChainEnsemble(args.model, y, p0, num_chains=1, N=N_PART, pf="poyiadjis_N", kernel=cfg["kernel"],
                            epsilon=cfg["epsilon"], prior=prior, subsequence_length=cfg["S"],
                            buffer_length=cfg["B"], dtype=args.dtype, seed=7, chain_offset=10 ** 6, device=dev_index)
        one.step(2)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        one.step(5)
        torch.cuda.synchronize(dev)
        single = 5.0 / (time.perf_counter() - t1)
    window_T = T_SERIES if cfg["S"] == -1 else (cfg["S"] + 2 * cfg["B"])
    wsize = 8 if args.dtype == "f64" else 4
    bytes_per_pt = 2 * (cfg["n"] + 1 + cfg["h"]) * wsize
    alg_bytes = float(C) * window_T * N_PART * bytes_per_pt        # per launch
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    if os.path.exists(tpath):
        try:
            rec = json.load(open(tpath)).get("{0}_{1}_C{2}".format(args.model, args.dtype, C))
            traffic = rec["bytes_per_launch"] if rec else None
        except Exception:
            traffic = None

    # what actually bounds the LDS-resident kernel: VALU issue (4 cycles per wave64 instruction per
    # SIMD, 1024 SIMDs at 2.4 GHz), from the committed SQ_INSTS_VALU pass of this workload
    valu = None
    vpath = os.path.join(ROOT, "profiles", "valu_issue.json")
    if os.path.exists(vpath):
        try:
            rec = json.load(open(vpath)).get("{0}_{1}_C{2}".format(args.model, args.dtype, C))
            if rec:
                insts = float(rec["SQ_INSTS_VALU_per_launch"])
                valu = {"insts_per_launch": insts, "issue_cycles_per_inst": 4, "simds": 1024, "clock_ghz": 2.4,
                        "issue_frac": insts * 4.0 / (1024 * 2.4e9 * kern_ms * 1e-3)}
        except Exception:
            valu = None

    if rank == 0:
        total_steps = float(C) * world * args.steps
        line = {
            "metric": "SGLD steps/sec (T=1000, N=1000 particles)",
            "value": total_steps / elapsed,
            "unit": "SGLD steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": ("SVM synthetic T=1000 N=1000, SGLD full sequence (S=-1), poyiadjis_N, prior kernel"
                             if args.model == "svm" else
                             "GARCH synthetic T=1000 N=1000, SGLD buffered PF S=16 B=4, poyiadjis_N, optimal kernel"),
                "chains_per_gpu": C,
                "chains_total": C * world,
                "rng": "device (xoshiro128++ per lane keyed by Philox4x32-10)",
                "kernel_variant": ens.ctx.variant_name(args.model, cfg["kernel"], args.dtype, "philox", N_PART),
                "parallelism": "independent chains, {0} GPU(s) x {1} chains, RCCL all_gather of samples".format(world, C),
            },
            "per_chain_steps_per_s": args.steps / elapsed,
            "single_chain_alone_steps_per_s": single,
            "us_per_pf_timestep": kern_ms * 1e3 / window_T,
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "kernel_ms": kern_ms,
                "algorithmic_bytes_per_launch": alg_bytes,
                "note": "state is LDS-resident by design: measured HBM traffic is far below the algorithmic bytes; "
                        "the binding resource is VALU issue (see valu_issue)",
                "valu_issue": valu,
            },
        }
        if world == 1 and not args.no_single_chain:
            line["parity"] = grad_error_vs_reference()
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.model, p0, y, prior, cfg, budget_s=args.cpu_budget)
            line["speedup_vs_cpu_1core"] = line["value"] / line["cpu_baseline"]["value"]
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
