"""Worker for tests/test_multigpu_gloo.py: one rank of the sharded re-integration on the CPU oracle over gloo."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build_map(api, pkg, wl, params, n_frames):
    import util
    return util.run_sequence(api, pkg, wl, params, n_frames)


def make_batch(pkg, synth, reint, wl, n_frames):
    frames, old, new = [], [], []
    for j in range(n_frames):
        rgba, mm, M_old = wl.frame(j)
        T_new = wl.pose(j) @ synth.pose_matrix(synth.look_rotation(0.004 * (j + 1), -0.002), [0.01, 0.002 * j, -0.005])
        frames.append(("host", rgba, mm))
        old.append(M_old)
        new.append(synth.world_to_camera(T_new))
    return reint.Batch(frames, old, new, wl.intr)


def main(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    import __graft_entry__ as ge
    import util
    pkg = ge.load_package()
    from dslam_amd.harness import reintegrate as reint
    from dslam_amd.harness import synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    api = ge.load_oracle().open_oracle(pkg.CApi)
    wl = synth.s_tiny()
    chunk = 16
    params = util.small_params(pkg, wl, num_local_blocks=0x800, num_buckets=0x1000, num_excess=0x400)
    n_frames = 5
    s, rs, v = build_map(api, pkg, wl, params, n_frames)
    batch = make_batch(pkg, synth, reint, wl, n_frames)
    timers = {}
    reint.reintegrate(api, s, v, rs, batch, rank=rank, world=world, chunk_blocks=chunk,
                      all_gather=reint.make_numpy_all_gather(api, s, dist, chunk), timers=timers)
    snap = util.snapshot(api, s, rs)
    ok = True
    msg = ""
    if rank == 0:
        # single-rank reference in the same process
        s1, rs1, v1 = build_map(api, pkg, wl, params, n_frames)
        reint.reintegrate(api, s1, v1, rs1, batch, rank=0, world=1)
        ref = util.snapshot(api, s1, rs1)
        try:
            util.assert_same_state(snap, ref, "sharded vs single")
            assert timers["gathered_bytes"] > 0
            changed = (ref["voxels"]["w_depth"] > 0).sum()
            assert changed > 1000
        except AssertionError as ex:
            ok, msg = False, str(ex)
    # every rank must hold the same gathered map
    digest = np.frombuffer(snap["voxels"].tobytes(), dtype=np.uint64).sum(dtype=np.uint64)
    t = torch.tensor([int(digest) & 0x7FFFFFFFFFFFFFFF], dtype=torch.int64)
    gathered = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(gathered, t)
    if rank == 0:
        if len({int(g.item()) for g in gathered}) != 1:
            ok, msg = False, "ranks hold different voxel arrays after the all-gather"
        with open(out_path, "w") as f:
            f.write("OK" if ok else "FAIL " + msg)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
