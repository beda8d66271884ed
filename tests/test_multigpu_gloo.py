"""N > 1 path on CPU: the sharded re-integration (harness/reintegrate.py) with world_size 2 over gloo, using the CPU
oracle as the engine.  The sharded result must equal the single-rank result byte for byte."""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_plan_region(pkg):
    from dslam_amd.harness.reintegrate import plan_region
    lo, groups = plan_region(last_free_block_id=0x800 - 1 - 100, num_local_blocks=0x800, world=2, chunk_blocks=16)
    assert lo == 0x800 - 128 and groups == 4
    lo, groups = plan_region(last_free_block_id=-1, num_local_blocks=0x800, world=4, chunk_blocks=64)
    assert lo == 0 and groups == 8
    with pytest.raises(ValueError):
        plan_region(10, 1000, 3, 64)


@pytest.mark.parametrize("world", [2])
def test_sharded_reintegration_equals_single_rank(tmp_path, world):
    port = _free_port()
    out = tmp_path / "result.txt"
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_gloo_worker.py"), str(r), str(world), str(port), str(out)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    assert out.read_text() == "OK", out.read_text() + "\n" + "\n".join(logs)
