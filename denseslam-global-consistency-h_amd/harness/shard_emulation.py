"""What the sharded re-integration (BASELINE configs[4]) will cost per rank at 2 / 4 / 8 GPUs, measured on ONE:
rank r of N is the same program with dslam_scene_set_shard(r, N), so every rank of every world size is run in turn on an
identical map and timed.  The batch's wall time on N GPUs is then max over ranks (compute) + the exchange: dirty-list
plan + pack + unpack are measured here too; the all-gather itself cannot be (one GPU), so its time is modelled from the
bytes a rank receives and the xGMI link rate, and labelled as a model.

    python denseslam-global-consistency-h_amd/harness/shard_emulation.py [keyframes_in_map] [batch]

prints one JSON object (profiles/r03_shard_emulation.json).
"""
import gc
import json
import sys
import time

import numpy as np

XGMI_LINK_GBPS = 153.0  # per link and direction; 7 links per GPU (task statement / MI355X_MICROARCH.md)
BLOCK_BYTES = 4096


def main():
    sys.path.insert(0, ".")
    import __graft_entry__ as ge
    pkg = ge.load_package()
    from dslam_amd.harness import reintegrate as reint
    from dslam_amd.harness import synth
    n_map = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    K = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    wl = synth.s_street(640, 480)
    frames = [wl.frame(i) for i in range(n_map)]
    eng = pkg.open_engine(0)
    params = pkg.SceneParams(num_local_blocks=0x40000, **wl.scene_kwargs)
    scene = eng.create_scene(params)
    view = eng.create_view(wl.W, wl.H)
    store = eng.create_frame_store(wl.W, wl.H, n_map)
    eng.frame_store_enable_lists(store, scene)
    for i, (rgba, mm, M) in enumerate(frames):
        eng.frame_store_put(store, i, rgba, mm)
    ids = list(range(n_map - K, n_map))
    new_poses = [synth.world_to_camera(wl.pose(i) @ synth.pose_matrix(synth.look_rotation(0.002 * (n + 1), 0.0), [0.01 * (n + 1), 0.0, 0.02]))
                 for n, i in enumerate(ids)]
    batch = reint.Batch([("store", store, i) for i in ids], [frames[i][2] for i in ids], new_poses, wl.intr)
    chunk = 64

    def build():
        """the live fusion every replica has done (identical every time: same calls on a reset scene)"""
        eng.reset_scene(scene)
        rs = eng.create_render_state(scene, wl.W, wl.H)
        for i, (rgba, mm, M) in enumerate(frames):
            eng.view_update_from_store(view, store, i, timestamp=float(i))
            eng.process_frame(scene, view, rs, M, wl.intr)
            eng.frame_store_put_visible_list(store, i, scene, rs)
        eng.synchronize()
        return rs

    out = {"workload": f"{wl.name} 640x480, map of {n_map} keyframes, batch = the last {K} de-/re-integrated at corrected poses",
           "chunk_blocks": chunk, "variants": {}}
    for variant, stored in (("allocation pass at the old pose (DeProcessFrame as the reference calls it)", False),
                            ("keyframe's stored visible list (dslam_deprocess_frame_stored)", True),
                            ("block-major batch (dslam_reintegrate_batch)", "batch")):
        rows = {}
        for world in (1, 2, 4, 8):
            per_rank, exch = [], []
            counts = None
            for rank in range(world):
                rs = build()
                eng.track_dirty(scene, True)
                if stored == "batch":
                    eng.reintegrate_batch(scene, view, rs, store, [], [], [], wl.intr)   # (set-up call: the scratch buffers)
                eng.synchronize()
                gc.collect(); gc.freeze()
                t0 = time.perf_counter()
                if world > 1:
                    eng.set_shard(scene, rank, world, chunk)
                eng.set_async(True)
                if stored == "batch":
                    eng.reintegrate_batch(scene, view, rs, store, ids, batch.old_poses, batch.new_poses, wl.intr)
                for k in range(0 if stored == "batch" else len(batch)):
                    eng.view_update_from_store(view, store, ids[k], timestamp=float(k))
                    if stored:
                        eng.deprocess_frame_stored(scene, view, store, ids[k], batch.old_poses[k], wl.intr)
                    else:
                        eng.deprocess_frame(scene, view, rs, batch.old_poses[k], wl.intr)
                    eng.process_frame(scene, view, rs, batch.new_poses[k], wl.intr, is_defusion=True)
                    if stored:
                        eng.frame_store_put_visible_list(store, ids[k], scene, rs)
                eng.synchronize()
                eng.set_async(False)
                t1 = time.perf_counter()
                per_rank.append((t1 - t0) * 1e3)
                # the exchange's local parts: plan, pack, unpack (of the other shards' blocks, from a scratch buffer)
                import torch
                counts = eng.shard_dirty_plan(scene, world, chunk)
                cap = max(1, max(counts))
                recv = torch.zeros((world, cap, BLOCK_BYTES), dtype=torch.uint8, device="cuda")
                eng.synchronize()
                t2 = time.perf_counter()
                counts = eng.shard_dirty_plan(scene, world, chunk)
                eng.shard_dirty_pack(scene, rank, recv[rank].data_ptr(), cap)
                if world > 1:
                    eng.shard_dirty_unpack(scene, rank, recv.data_ptr(), cap)
                eng.synchronize()
                exch.append((time.perf_counter() - t2) * 1e3)
                eng.track_dirty(scene, False)
                eng.set_shard(scene, 0, 1, chunk)
                rs.close()
            cap = max(counts)
            recv_bytes = (world - 1) * cap * BLOCK_BYTES
            rows[str(world)] = {
                "compute_ms_per_rank": [round(x, 3) for x in per_rank], "compute_ms_max": round(max(per_rank), 3),
                "plan_pack_unpack_ms_max": round(max(exch), 3) if world > 1 else 0.0,
                "dirty_blocks_per_shard": counts, "bytes_received_per_rank": recv_bytes,
                # direct all-gather over the 7 point-to-point links (each peer's shard on its own link) / a ring on one link
                "all_gather_ms_model_direct": round(cap * BLOCK_BYTES / (XGMI_LINK_GBPS * 1e9) * 1e3, 4) if world > 1 else 0.0,
                "all_gather_ms_model_ring": round(recv_bytes / (XGMI_LINK_GBPS * 1e9) * 1e3, 4) if world > 1 else 0.0,
            }
        base = rows["1"]["compute_ms_max"]
        for w in rows:
            r = rows[w]
            r["projected_batch_ms"] = round(r["compute_ms_max"] + r["plan_pack_unpack_ms_max"] + r["all_gather_ms_model_direct"], 3)
            r["projected_vs_1_rank_same_variant"] = round(r["projected_batch_ms"] / base, 3)
        out["variants"][variant] = rows
    ref = out["variants"]["allocation pass at the old pose (DeProcessFrame as the reference calls it)"]["1"]["compute_ms_max"]
    for variant in out["variants"]:
        for w in out["variants"][variant]:
            r = out["variants"][variant][w]
            r["projected_vs_1_rank_reference_call"] = round(r["projected_batch_ms"] / ref, 3)
    out["keyframes_per_s_1_rank"] = {v: round(K / (out["variants"][v]["1"]["compute_ms_max"] * 1e-3), 1) for v in out["variants"]}
    out["note"] = ("compute = de-/re-integration of the batch on one rank incl. the allocation passes every rank repeats; "
                   "exchange = measured plan + pack + unpack, plus a MODELLED all-gather (one GPU here)")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
