"""Multi-rank readiness on ONE GPU (no scaling number: a rehearsal).  Two child processes started with
`python -m torch.distributed.run --nproc-per-node 2` share the visible MI355X under gloo; each owns a ChainEnsemble
of C chains with chain_offset = rank * C, runs real SGLD steps, and the ranks all_gather their samples -- exactly what
bench.py's ranks and `ChainEnsemble.gather_samples` do on an 8-GPU node (there over RCCL).  The gathered [2C, P]
samples must equal a single-process run of 2C chains BIT FOR BIT: a chain's trajectory is fixed by its global chain
id (device generator key, host / device window draws, update noise), not by the rank partition.  Full-sequence and
buffered chains, host- and device-side window sampling, three models."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "helpers"))


def test_two_ranks_on_one_gpu_equal_one_process(tmp_path):
    import two_rank_ensemble as tr
    C = 72          # more than 64 windows per launch on both sides: the same kernel variant (<= 64 takes the latency variant)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "two_ranks.npz")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PFG_DIST_BACKEND="gloo", OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "helpers", "two_rank_ensemble.py"), out, str(C)]
    res = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    got = np.load(out)
    assert int(got["world"]) == 2
    for name, model, kw, steps in tr.cases():
        whole = tr.run_case(model, kw, steps, 2 * C, 0)           # this process: all 2C chains on the same GPU
        ref = whole.theta()
        assert got[name].shape == ref.shape == (2 * C, ref.shape[1])
        assert np.array_equal(got[name], ref), (name, np.max(np.abs(got[name] - ref)))
        assert len({tuple(r) for r in ref}) == 2 * C              # distinct chains
        g, _ = whole.last_gradient_statistics()
        assert np.array_equal(got[name + "/grad_local"], g[:C])   # rank 0's last gradients = chains [0, C) of the whole
