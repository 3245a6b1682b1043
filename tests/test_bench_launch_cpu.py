"""bench.py's own multi-rank launcher (`python bench.py --gpus N` with no torch.distributed environment), exercised on CPU over gloo."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, capture_output=True, text=True, timeout=600, cwd=ROOT, env=e)


def test_bench_gpus2_self_launch_dry_run():
    r = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run"])
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                       # exactly ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["dry_run"] is True and out["allreduce_ok"] is True
    assert out["scaling"] == "weak" and out["ms_per_step"] > 0


def test_bench_launcher_propagates_rank_failure():
    """A rank that dies (here: every rank refuses to run without a GPU) must turn into a non-zero exit code of the launcher."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("GPU present: the ranks would run")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--launch-timeout", "300"])
    assert r.returncode != 0


def test_bench_single_rank_dry_run_and_torchrun_env():
    r = _run(["--gpus", "1", "--steps", "2", "--warmup", "0", "--dry-run"])
    assert r.returncode == 0, r.stdout + r.stderr
    assert json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])["n_gpus"] == 1
    # started as a rank by an external launcher whose WORLD_SIZE disagrees with --gpus: refuse
    r = _run(["--gpus", "2", "--dry-run"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0
