"""Worker for tests/test_multigpu_gloo.py: one rank of the sharded re-integration on the CPU oracle over gloo -- and for
tests/test_gpu_multirank.py: the same with the HIP engine, every rank a process of its own on the ONE GPU of the box (engine
"hip": sharded kernels + pack / unpack on the device, the collective staged through the host)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build_map(api, pkg, wl, params, n_frames, maintenance, store=None):
    """Fuse n_frames; with `maintenance` the map is also decayed and slid like BASELINE configs[2] (reference
    DenseSlam.cpp:215-232), so that freed slots have gone back to the pool in arbitrary order and live blocks sit anywhere.
    store: a list that receives a keyframe store holding every keyframe's images and fusion-time visible list."""
    import util
    box = {}

    def keep(i, scene, rs, view):
        if "st" not in box:
            box["st"] = api.create_frame_store(wl.W, wl.H, n_frames)
            api.frame_store_enable_lists(box["st"], scene)
        api.frame_store_put_view(box["st"], i, view)
        api.frame_store_put_visible_list(box["st"], i, scene, rs)
    kw = dict(after_frame=keep) if store is not None else {}
    # (after_frame runs behind decay / slide of the same frame; the list is the one the fusion left in the render state --
    # entries the maintenance took since are skipped by the stored-list de-integration, as the contract says)
    out = util.run_sequence(api, pkg, wl, params, n_frames, **kw) if not maintenance else \
        util.run_sequence(api, pkg, wl, params, n_frames, decay=(1, 2, True), slide=2, **kw)
    if store is not None:
        store.append(box["st"])
    return out


def make_batch(pkg, synth, reint, wl, first, n):
    frames, old, new = [], [], []
    for j in range(first, first + n):
        rgba, mm, M_old = wl.frame(j)
        T_new = wl.pose(j) @ synth.pose_matrix(synth.look_rotation(0.004 * (j + 1), -0.002), [0.01, 0.002 * j, -0.005])
        frames.append(("host", rgba, mm))
        old.append(M_old)
        new.append(synth.world_to_camera(T_new))
    return reint.Batch(frames, old, new, wl.intr)


def main(rank, world, port, out_path, maintenance, batched=False, engine="oracle"):
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
    if engine == "hip":
        torch.cuda.set_device(0)
        api = pkg.open_engine(0)
        gather = lambda sc: reint.make_staged_all_gather(api, sc, dist, api.synchronize)
    else:
        api = ge.load_oracle().open_oracle(pkg.CApi)
        gather = lambda sc: reint.make_numpy_all_gather(api, sc, dist)
    wl = synth.s_tiny()
    chunk = 16
    params = util.small_params(pkg, wl, num_local_blocks=0x800, num_buckets=0x1000, num_excess=0x400)
    n_frames = 12 if maintenance else 5
    stores = [] if batched else None
    s, rs, v = build_map(api, pkg, wl, params, n_frames, maintenance, stores)
    # the keyframes still inside the window are the ones a correction re-fuses
    first, count = (n_frames - 3, 3) if maintenance else (0, n_frames)
    batch = make_batch(pkg, synth, reint, wl, first, count)
    if batched:   # the same keyframes out of the store, as ONE dslam_reintegrate_batch call per rank
        batch = reint.Batch([("store", stores[0], j) for j in range(first, first + count)], batch.old_poses, batch.new_poses, batch.intr)
    timers = {}
    counts = reint.reintegrate(api, s, v, rs, batch, rank=rank, world=world, chunk_blocks=chunk,
                               all_gather=gather(s), timers=timers, batched=batched)
    snap = util.snapshot(api, s, rs)
    ok = True
    msg = ""
    if rank == 0:
        # single-rank reference in the same process
        stores1 = [] if batched else None
        s1, rs1, v1 = build_map(api, pkg, wl, params, n_frames, maintenance, stores1)
        before = util.snapshot(api, s1, rs1)
        batch1 = batch if not batched else reint.Batch([("store", stores1[0], f[2]) for f in batch.frames], batch.old_poses,
                                                       batch.new_poses, batch.intr)
        # (the single-rank reference of the batched run is the per-keyframe loop with stored lists: the call's definition)
        reint.reintegrate(api, s1, v1, rs1, batch1, rank=0, world=1, stored_lists=batched)
        ref = util.snapshot(api, s1, rs1)
        try:
            util.assert_same_state(snap, ref, "sharded vs single")
            changed = int((ref["voxels"].view(np.uint64) != before["voxels"].view(np.uint64)).any(axis=1).sum())
            assert changed > 200, f"the batch changed only {changed} blocks"
            assert timers["dirty_blocks"] >= changed and timers["gathered_bytes"] == world * max(counts) * 4096
            if maintenance:
                # the premise of the old exchange ("used slots are a top range of the pool") must be FALSE here, or the
                # test does not cover the case it exists for
                st = before["stats"]
                used = params.num_local_blocks - 1 - st["last_free_block_id"]
                lowest = int(before["hash"]["ptr"][before["hash"]["ptr"] >= 0].min())
                assert lowest < params.num_local_blocks - used, "live blocks still form a top range"
                assert st["slid_block_count"] + st["decayed_block_count"] > 0
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
    main(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]) != 0,
         len(sys.argv) > 6 and int(sys.argv[6]) != 0, sys.argv[7] if len(sys.argv) > 7 else "oracle")
