"""N > 1 on hardware, as far as a one-GPU box allows: the sharded re-integration with world_size 2 and 4, EVERY RANK A PROCESS OF
ITS OWN with its own HIP engine on the box's one GPU (RCCL refuses two ranks on one device, so the collective is gloo over
host-staged buffers: harness/reintegrate.py::make_staged_all_gather).  What runs on the device is the production path of a
sharded rank: the sharded per-keyframe kernels or k_reintegrate_blocks<.,1>, the dirty plan, pack and unpack.  Rank 0 also runs
the batch unsharded on the same engine; the gathered map must equal it byte for byte and be the same on every rank."""
import os
import subprocess
import sys

import pytest

from test_multigpu_gloo import _free_port

HERE = os.path.dirname(os.path.abspath(__file__))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,maintenance,batched", [(2, 0, 0), (2, 1, 1), (4, 1, 1)])
def test_ranks_as_processes_sharing_the_gpu(tmp_path, world, maintenance, batched):
    port = _free_port()
    out = tmp_path / "result.txt"
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_gloo_worker.py"), str(r), str(world), str(port), str(out),
                               str(maintenance), str(batched), "hip"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    assert out.read_text() == "OK", out.read_text() + "\n" + "\n".join(logs)
