"""bench.py's multi-rank entry: `--gpus N` without a launcher starts N ranks itself, before the parent
touches a GPU.  Exercised on CPU with the gloo rehearsal mode (the rank logic: chain ranges, barrier,
max over ranks, gather of ChainEnsemble-shaped [C, P] samples in global chain order)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=240):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                          env=env, timeout=timeout)


def test_gpus_2_spawns_two_ranks_and_gathers_in_chain_order():
    res = _run(["--gpus", "2", "--cpu-rehearsal", "--chains-per-gpu", "5"])
    assert res.returncode == 0, res.stdout + res.stderr
    lines = [json.loads(l) for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout                      # rank 0 prints ONE line
    line = lines[0]
    assert line["ranks"] == 2 and line["chains_total"] == 10 and line["gathered_in_global_chain_order"] is True
    assert line["elapsed_max_s"] >= 0.02                    # the max over ranks (rank 1 sleeps 20 ms)


def test_gpus_must_match_world_size():
    res = _run(["--gpus", "2", "--cpu-rehearsal"], env_extra={"WORLD_SIZE": "3", "RANK": "0"})
    assert res.returncode != 0 and "WORLD_SIZE=3" in (res.stdout + res.stderr)


def test_single_process_rehearsal():
    res = _run(["--gpus", "1", "--cpu-rehearsal", "--chains-per-gpu", "4"])
    assert res.returncode == 0, res.stdout + res.stderr
    line = [json.loads(l) for l in res.stdout.splitlines() if l.startswith("{")][0]
    assert line["ranks"] == 1 and line["chains_total"] == 4
