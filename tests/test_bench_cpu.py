"""CPU: bench.py's multi-rank control flow without a GPU -- `python bench.py --gpus N --dry-run` must start its own N rank
processes (no launcher, WORLD_SIZE unset), rendezvous over gloo on 127.0.0.1, shard the tiny model with tp.shard_llama,
run the path's exchange pattern under the barrier / max-over-ranks timing protocol and print ONE JSON line from rank 0."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n, extra=()):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "1"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--dry-run", "--steps", "3",
                          "--warmup", "1", *extra], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [1, 2, 4])
def test_bench_self_launch_dry_run(n):
    d = _run(n)
    assert d["n_gpus"] == n and d["steps"] == 3 and d["warmup"] == 1 and d["dry_run"] is True
    assert d["config"]["parallelism"] == f"tp{n}"
    assert d["config"]["q_rows_per_rank"] == 1024 // n         # rehearsal model: 1024 q rows split over the ranks
    assert d["scaling"] == ("strong" if n > 1 else "weak") and d["value"] > 0


def test_bench_rejects_world_mismatch():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-run"], env=env,
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "WORLD_SIZE" in (out.stderr + out.stdout)
