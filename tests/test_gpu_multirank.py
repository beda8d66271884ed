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


def test_bench_runs_with_two_ranks_sharing_the_gpu():
    """`bench.py --gpus 2` end to end on the one-GPU box (DSLAM_BENCH_REHEARSE=1: both ranks on cuda:0, gloo, host-staged
    exchange): the N > 1 branches of the bench -- barrier + max over ranks, the three sharded re-integration legs, the replica
    check across ranks and rank 0's unsharded repeat -- run across real process boundaries and their self-checks hold.  The
    rates of such a run mean nothing and are not looked at."""
    import json
    env = dict(os.environ, DSLAM_BENCH_REHEARSE="1")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "4",
                        "--no-stress", "--no-extra-rates", "--no-cpu-baseline", "--reint", "8"], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and "rehearsal" in d["config"]
    re_ = d["reintegration"]
    assert "error" not in re_, re_
    assert re_["ranks_seen_by_rccl"] == 2 and re_["map_checksum_equal"] is True and re_["equals_unsharded_run_on_rank0"] is True
    for form in ("reference_calls", "stored_lists", "block_major"):
        assert re_[form]["dirty_blocks"] > 0 and re_[form]["gathered_bytes"] > 0
