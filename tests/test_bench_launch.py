"""CPU: `python bench.py --gpus N`, run without a launcher as the driver runs it, must start N ranks itself.
--dry-run brings the ranks up over gloo and counts them; no kernel runs and no GPU is touched."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], env=env, capture_output=True,
                          text=True, timeout=280)


def test_gpus_2_starts_two_ranks():
    r = _run("--gpus", "2", "--dry-run")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout                      # ONE json line, rank 0's
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["dry_run"] is True
    # fixed global batch of 8 views split over the ranks: strong scaling
    assert out["config"] == {"global_batch_views": 8, "views_per_gpu": 4} and out["scaling"] == "strong"


def test_weak_mode_and_single_rank_line():
    r = _run("--gpus", "2", "--dry-run", "--global-batch", "0")
    assert r.returncode == 0, r.stderr[-4000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["config"] == {"global_batch_views": 2, "views_per_gpu": 1} and out["scaling"] == "weak"
    r = _run("--gpus", "1", "--dry-run", env_extra={"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29577"})
    assert r.returncode == 0, r.stderr[-4000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert out["n_gpus"] == 1 and out["config"] == {"global_batch_views": 1, "views_per_gpu": 1}


def test_world_size_mismatch_is_an_error():
    r = _run("--gpus", "2", "--dry-run", env_extra={"WORLD_SIZE": "1", "RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def test_batch_not_divisible_is_an_error():
    r = _run("--gpus", "2", "--dry-run", "--global-batch", "3")
    assert r.returncode != 0


def test_gpus_4_and_8_split_the_batch_of_eight():
    """The launch lines of the scaling runs (N = 4, 8): N ranks come up, one JSON line, 2 and 1 views per rank."""
    for n, per in ((4, 2), (8, 1)):
        r = _run("--gpus", str(n), "--dry-run")
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout
        out = json.loads(lines[0])
        assert out["n_gpus"] == n and out["ranks_seen"] == n and out["scaling"] == "strong"
        assert out["config"] == {"global_batch_views": 8, "views_per_gpu": per}
