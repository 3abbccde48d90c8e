"""bench.py --gpus N must start N ranks by itself (the driver's SCALE command is `python bench.py --gpus N`) and relay
rank 0's JSON line as its only stdout line.  The launch / rendezvous / relay path runs here on the CPU (gloo); the
reference has no counterpart (Project 1/solver.py:1367 run_case is one process)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]


def run_bench(args, env_extra=None, drop=("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")):
    env = {k: v for k, v in os.environ.items() if k not in drop}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(REPO / "bench.py")] + args, env=env, capture_output=True, text=True, timeout=300)


@pytest.mark.parametrize("n", [2, 3])
def test_gpus_n_starts_n_ranks_and_relays_one_line(n):
    r = run_bench(["--gpus", str(n), "--rehearse-launch"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["ranks_seen_by_all_reduce"] == n and out["self_launched"] is True


def test_single_gpu_runs_in_process():
    r = run_bench(["--gpus", "1", "--rehearse-launch"])
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip())
    assert out["n_gpus"] == 1 and out["self_launched"] is False


def test_launcher_world_size_must_match_gpus():
    """Under an external launcher (WORLD_SIZE set) a different --gpus is an error, not a silently different run."""
    r = run_bench(["--gpus", "4", "--rehearse-launch"], env_extra={"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, drop=())
    assert r.returncode != 0 and "WORLD_SIZE = 2" in r.stderr


def test_failing_rank_fails_the_parent():
    """Without a GPU every rank of the real bench exits non-zero: the parent must report it and print no JSON line."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a machine without a GPU")
    r = run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu"])
    assert r.returncode != 0
    assert r.stdout.strip() == ""
    assert "exited with code" in r.stderr
