"""bench.py's own launcher (VERDICT r01 "What's missing" #1): `python bench.py --gpus N` with no
external torchrun must start N ranks itself, rendezvous on 127.0.0.1, and print exactly one JSON
line from rank 0.  Exercised here without a GPU through NP_BENCH_DRYRUN=1 (gloo, no kernels: the
line says "dry_run": true and carries no throughput)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def run_bench(args, extra_env=None, timeout=240):
    env = dict(os.environ, NP_BENCH_DRYRUN="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], env=env, timeout=timeout,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


@pytest.mark.parametrize("n", [2, 3])
def test_self_launch_prints_one_json_line(n):
    p = run_bench(["--gpus", str(n), "--steps", "4", "--warmup", "1"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, p.stdout
    j = json.loads(lines[0])
    assert j["dry_run"] is True and j["n_gpus"] == n and j["steps"] == 4 and j["warmup"] == 1
    assert j["ranks_seen"] == n          # every spawned rank took part in the max-over-ranks collective


def test_external_launcher_env_is_respected():
    """With RANK set (torch.distributed.run's convention) bench.py must NOT spawn: it is one rank."""
    p = run_bench(["--gpus", "1", "--steps", "2", "--warmup", "0"],
                  {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "NP_BENCH_FORCE_DIST": "1",
                   "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29617"})
    assert p.returncode == 0, p.stderr[-2000:]
    j = json.loads(p.stdout.strip())
    assert j["ranks_seen"] == 1


def test_world_size_mismatch_is_an_error():
    p = run_bench(["--gpus", "2", "--steps", "2", "--warmup", "0"],
                  {"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "4", "MASTER_ADDR": "127.0.0.1",
                   "MASTER_PORT": "29618"})
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr


def test_dead_rank_takes_the_job_down_quickly():
    """A rank that exits non-zero must end the job (no sitting in the rendezvous time-out)."""
    p = run_bench(["--gpus", "2", "--steps", "2", "--warmup", "0"], {"NP_BENCH_DRYRUN_FAIL_RANK": "1"}, timeout=90)
    assert p.returncode != 0
    assert "rank 1 failed" in p.stderr
