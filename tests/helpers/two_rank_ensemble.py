"""One rank of the two-rank ChainEnsemble rehearsal (tests/test_gpu_two_ranks.py): started by
`python -m torch.distributed.run --nproc-per-node 2`, both ranks on the ONE visible GPU under gloo.
Each rank owns a ChainEnsemble of C chains with chain_offset = rank * C, runs K SGLD steps and the ranks
all_gather their samples (the job's only collective); rank 0 writes the gathered [world * C, P] arrays."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "stochastic-gradient-mcmc-for-non-linear-state-models---mth422_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def cases():
    """(name, model, ChainEnsemble kwargs, steps) -- shared with the single-process side of the test."""
    return [("svm_full", "svm", dict(N=256, epsilon=0.05, seed=31), 3),
            ("svm_buffered_host", "svm", dict(N=256, epsilon=0.05, seed=32, subsequence_length=16, buffer_length=4,
                                              window_sampling="host"), 4),
            ("garch_buffered_device", "garch", dict(N=300, epsilon=0.01, seed=33, subsequence_length=16, buffer_length=4,
                                                    window_sampling="device"), 4),
            ("lgssm_full", "lgssm", dict(N=100, epsilon=0.05, seed=34), 3)]


def series(model, T=150):
    from test_host_logic import default_params, GEN
    np.random.seed(17)
    return GEN[model](T=T, parameters=default_params(model))["observations"], default_params(model)


def run_case(model, kw, steps, C, offset):
    from sgmcmc_ssm_amd.ensemble import ChainEnsemble
    y, p = series(model)
    ens = ChainEnsemble(model, y, p, num_chains=C, chain_offset=offset, **kw)
    ens.step(steps)
    ens.synchronize()
    return ens


def main():
    out_path, C = sys.argv[1], int(sys.argv[2])
    os.environ.setdefault("PFG_DIST_BACKEND", "gloo")       # two ranks on one GPU: RCCL refuses, the rendezvous runs on gloo
    import torch
    from sgmcmc_ssm_amd import distributed
    rank, world, _ = distributed.init_from_env()
    torch.cuda.set_device(0)
    lo, _ = distributed.chain_range(rank, C)
    res = {}
    for name, model, kw, steps in cases():
        ens = run_case(model, kw, steps, C, lo)
        res[name] = ens.gather_samples().cpu().numpy()          # [world * C, P], global chain order, on every rank
        g, ll = ens.last_gradient_statistics()
        res[name + "/grad_local"] = g
    distributed.barrier()
    if rank == 0:
        np.savez(out_path, world=world, **res)
    distributed.barrier()
    import torch.distributed as dist
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
