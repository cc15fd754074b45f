"""N>1 path on CPU: world_size-2 gloo processes shard chains and gather samples."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sgmcmc_ssm_amd import distributed


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, C, P, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, _ = distributed.init_from_env(backend="gloo")
    lo, hi = distributed.chain_range(r, C)
    # each rank's "samples": a deterministic function of the GLOBAL chain index
    local = torch.tensor([[g * 10.0 + j for j in range(P)] for g in range(lo, hi)], dtype=torch.float64)
    allsamp = distributed.gather_samples(local)
    tmax = distributed.max_over_ranks(1.0 + r)
    distributed.barrier()
    q.put((r, lo, hi, allsamp.numpy(), tmax))
    dist.destroy_process_group()


def test_two_rank_gather_matches_single_process():
    world, C, P = 2, 3, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, C, P, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = np.array([[g * 10.0 + j for j in range(P)] for g in range(world * C)])
    ranges = set()
    for r, lo, hi, allsamp, tmax in res:
        np.testing.assert_array_equal(allsamp, expect)     # same on every rank, global chain order
        assert tmax == 2.0
        ranges.add((lo, hi))
    assert ranges == {(0, 3), (3, 6)}


def test_single_process_is_identity():
    x = torch.arange(6, dtype=torch.float64).reshape(3, 2)
    assert torch.equal(distributed.gather_samples(x), x)
    assert distributed.max_over_ranks(3.5) == 3.5
    assert distributed.chain_range(3, 5) == (15, 20)
