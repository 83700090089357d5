"""`python bench.py --gpus N` must start its own ranks when no launcher set WORLD_SIZE (the driver's invocation),
print exactly one JSON line on stdout and exit 0; a failing rank must make the parent exit non-zero.
Runs without a GPU through bench.py's --dry-launch mode (ranks join a gloo all-reduce only)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *extra], env=env, capture_output=True,
                          text=True, timeout=600)


def test_self_launch_two_ranks_prints_one_json_line():
    r = _run(["--gpus", "2", "--dry-launch"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["joined_ranks"] == 2


def test_self_launch_propagates_rank_failure():
    r = _run(["--gpus", "2", "--dry-launch"], {"MI_BENCH_DRY_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
