"""N > 1 path on CPU: the sharded re-integration (harness/reintegrate.py) with world_size 2 and 4 over gloo, using the
CPU oracle as the engine.  The sharded result must equal the single-rank result byte for byte -- also on a map that has
been decayed and slid, where freed slots have returned to the pool in arbitrary order."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import util

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_dirty_plan_lists_exactly_the_visited_blocks(pkg, synth, oracle):
    """dslam_shard_dirty_plan: per shard the slots a (de-)integration pass walked over since tracking began, ascending;
    pack / unpack move exactly those blocks."""
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl, num_local_blocks=0x800, num_buckets=0x1000, num_excess=0x400)
    s, rs, v = util.run_sequence(oracle, pkg, wl, p, 4)
    before = oracle.download_voxel_blocks(s)
    with pytest.raises(pkg.DslamError):
        oracle.shard_dirty_plan(s, 2, 16)  # tracking was never enabled
    oracle.track_dirty(s, True)
    rgba, mm, M = wl.frame(4)
    oracle.view_update(v, rgba, mm)
    oracle.process_frame(s, v, rs, M, wl.intr)
    after = oracle.download_voxel_blocks(s)
    ids = oracle.download_visible_ids(rs)
    slots = np.sort(oracle.download_hash_table(s)["ptr"][ids])
    slots = slots[slots >= 0]
    world, chunk = 4, 16
    counts = oracle.shard_dirty_plan(s, world, chunk)
    want = [int((((slots // chunk) % world) == r).sum()) for r in range(world)]
    assert counts == want and sum(counts) == len(slots) > 300
    changed = np.nonzero((before.view(np.uint64) != after.view(np.uint64)).any(axis=1))[0]
    assert np.isin(changed, slots).all()
    with pytest.raises(pkg.DslamError):
        oracle.shard_dirty_plan(s, 3, 16)  # 0x800 is not a multiple of 3 * 16
    # pack every shard, wipe the map's blocks, unpack: the visited blocks are back, nothing else is touched
    cap = max(counts)
    recv = np.zeros((world, cap, 4096), np.uint8)
    for r in range(world):
        oracle.shard_dirty_pack(s, r, recv[r].ctypes.data, cap)
    oracle.upload_voxel_blocks(s, 0, np.zeros_like(after))
    oracle.shard_dirty_unpack(s, -1, recv.ctypes.data, cap)
    got = oracle.download_voxel_blocks(s)
    assert np.array_equal(got[slots].view(np.uint64), after[slots].view(np.uint64))
    rest = np.setdiff1d(np.arange(p.num_local_blocks), slots)
    assert not got[rest].view(np.uint64).any()
    oracle.track_dirty(s, False)


@pytest.mark.parametrize("world,maintenance,batched", [(2, 0, 0), (2, 1, 0), (4, 1, 0), (2, 1, 1)])
def test_sharded_reintegration_equals_single_rank(tmp_path, world, maintenance, batched):
    """batched: every rank runs the batch as ONE dslam_reintegrate_batch call (on the oracle: the loop that defines it) on its
    shard; the reference is the unsharded per-keyframe loop with stored lists."""
    port = _free_port()
    out = tmp_path / "result.txt"
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "_gloo_worker.py"), str(r), str(world), str(port), str(out),
                               str(maintenance), str(batched)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
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
