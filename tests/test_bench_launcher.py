"""`python bench.py --gpus N` without a launcher starts N ranks itself (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE set)
before the parent touches a GPU; --spawn-dry-run lets the ranks report and leave before they touch one either, so the
launcher path can be checked on a machine without GPUs."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*args, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), capture_output=True, text=True, env=e, timeout=120)


def test_gpus_flag_spawns_one_rank_per_gpu():
    res = _run("--gpus", "2", "--spawn-dry-run")
    assert res.returncode == 0, res.stderr
    out = json.loads(res.stdout.strip().splitlines()[-1])
    assert out["n_gpus"] == 2 and out["failed_ranks"] == []
    assert [(r["rank"], r["local_rank"], r["world_size"]) for r in out["ranks"]] == [(0, 0, 2), (1, 1, 2)]


def test_a_launcher_s_environment_is_respected():
    """Under torch.distributed.run (WORLD_SIZE set) bench.py is one rank and must not spawn anything."""
    res = _run("--gpus", "4", "--spawn-dry-run", env={"RANK": "3", "LOCAL_RANK": "3", "WORLD_SIZE": "4"})
    assert res.returncode == 0, res.stderr
    assert json.loads(res.stdout.strip().splitlines()[-1]) == {"rank": 3, "local_rank": 3, "world_size": 4}
