"""Multi-GPU plumbing: one process per GPU, chains sharded, one gather.

The particle-filter path shards over *independent units* (chains, minibatch windows,
sequences): no data-path collective exists.  Rank r owns global chains
[r*chains_per_rank, (r+1)*chains_per_rank) -- weak scaling -- and Philox stream ids are the
GLOBAL chain indices, so a chain's trajectory does not depend on how many GPUs run the job.
The only communication is `gather_samples` (RCCL all_gather over xGMI on GPUs; gloo on CPU
for tests): [chains_per_rank, P] f64 per rank, i.e. a few KB -- latency-bound."""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* if WORLD_SIZE > 1.
    Returns (rank, world_size, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("PFG_DIST_BACKEND")     # rehearsal override (e.g. gloo on one GPU)
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"   # "nccl" is RCCL on ROCm
        kwargs = {}
        if backend == "nccl":
            index = local_rank % max(1, torch.cuda.device_count())
            torch.cuda.set_device(index)
            # bind the communicator to this rank's GPU: barrier() otherwise guesses the device from the global rank
            kwargs["device_id"] = torch.device("cuda", index)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        try:
            dist.init_process_group(backend=backend, rank=rank, world_size=world, **kwargs)
        except TypeError:                       # a torch without the device_id argument
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def chain_range(rank, chains_per_rank):
    """Global chain indices owned by `rank` (contiguous block)."""
    start = int(rank) * int(chains_per_rank)
    return start, start + int(chains_per_rank)


def gather_samples(local):
    """[C, P] per rank -> [world*C, P] on every rank, rank-major (= global chain order)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return local.clone()
    local = local.contiguous()
    dev = local.device
    if dist.get_backend() == "gloo" and dev.type != "cpu":
        local = local.cpu()                   # gloo rehearsal of a GPU job: stage through the host
    parts = [torch.empty_like(local) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, local)
    return torch.cat(parts, dim=0).to(dev)


def max_over_ranks(value, device=None):
    """max of a python float over ranks (bench timing)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    if dist.get_backend() == "gloo":
        device = None
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
