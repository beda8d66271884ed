"""Round 4, VERDICT item 6 (overlap of the replicated allocation passes with the block work of a sharded rank), measured as far as
it can be without building it: do allocation passes of ONE stream run concurrently with the re-integration batch of a sharded
rank on ANOTHER stream of the same GPU?  Two engines (own streams, scratch, tickets) in one process:
  A: `passes` allocation passes (visible-list only: the map does not change) over the bench's frames, asynchronous;
  B: dslam_reintegrate_batch of 32 keyframes on a scene sharded 1 of `world` (its own 32 passes, then k_reintegrate_blocks<.,1>).
Timed: A alone, B alone, both started back to back.  If the streams overlap, both together take max(A, B); if the block launch
holds the SIMDs (4 waves x 121 VGPRs each), A's passes queue behind it.

    python profiles/experiments/overlap_probe.py [world] [passes]      (GPU box)
"""
import gc
import json
import sys
import time

sys.path.insert(0, ".")
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
from dslam_amd.harness import reintegrate as reint  # noqa: E402
from dslam_amd.harness import synth  # noqa: E402


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    passes = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    n_map, K, chunk = 120, 32, 64
    wl = synth.s_street(640, 480)
    frames = [wl.frame(i) for i in range(n_map)]
    engs = [pkg.open_engine(0), pkg.open_engine(0)]
    params = pkg.SceneParams(num_local_blocks=0x40000, **wl.scene_kwargs)
    objs = []
    for eng in engs:
        scene = eng.create_scene(params)
        view = eng.create_view(wl.W, wl.H)
        store = eng.create_frame_store(wl.W, wl.H, n_map)
        eng.frame_store_enable_lists(store, scene)
        for i, (rgba, mm, M) in enumerate(frames):
            eng.frame_store_put(store, i, rgba, mm)
        objs.append((eng, scene, view, store))
    ids = list(range(n_map - K, n_map))
    new_poses = [synth.world_to_camera(wl.pose(i) @ synth.pose_matrix(synth.look_rotation(0.002 * (n + 1), 0.0), [0.01 * (n + 1), 0.0, 0.02]))
                 for n, i in enumerate(ids)]
    old_poses = [frames[i][2] for i in ids]

    def build(k):
        eng, scene, view, store = objs[k]
        eng.reset_scene(scene)
        rs = eng.create_render_state(scene, wl.W, wl.H)
        for i, (rgba, mm, M) in enumerate(frames):
            eng.view_update_from_store(view, store, i, timestamp=float(i))
            eng.process_frame(scene, view, rs, M, wl.intr)
            eng.frame_store_put_visible_list(store, i, scene, rs)
        eng.synchronize()
        return rs

    def start_a(rs):
        eng, scene, view, store = objs[0]
        for p in range(passes):
            i = ids[p % K]
            eng.view_update_from_store(view, store, i, timestamp=float(p))
            eng.allocate_scene_from_depth(scene, view, rs, new_poses[p % K], wl.intr, only_update_visible_list=True)

    def start_b(rs):
        eng, scene, view, store = objs[1]
        eng.reintegrate_batch(scene, view, rs, store, ids, old_poses, new_poses, wl.intr)

    out = {"world": world, "passes_on_stream_A": passes, "runs": []}
    for rep in range(3):
        row = {}
        for mode in ("A", "B", "AB"):
            rs_a, rs_b = build(0), build(1)
            eb, sb, vb, stb = objs[1]
            eb.reintegrate_batch(sb, vb, rs_b, stb, [], [], [], wl.intr)   # (set-up call: the scratch buffers)
            if world > 1:
                eb.set_shard(sb, 0, world, chunk)
            for e in engs:
                e.synchronize()
                e.set_async(True)
            gc.collect()
            t0 = time.perf_counter()
            if "A" in mode:
                start_a(rs_a)
            t_mid = time.perf_counter()
            if "B" in mode:
                start_b(rs_b)
            t_enq = time.perf_counter()
            for e in engs:
                e.synchronize()
            t1 = time.perf_counter()
            for e in engs:
                e.set_async(False)
            if world > 1:
                eb.set_shard(sb, 0, 1, chunk)
            row[mode] = {"ms": round((t1 - t0) * 1e3, 3), "host_enqueue_ms": round((t_enq - t0) * 1e3, 3), "enqueue_A_ms": round((t_mid - t0) * 1e3, 3)}
            rs_a.close(); rs_b.close()
        row["sum_ms"] = round(row["A"]["ms"] + row["B"]["ms"], 3)
        row["overlap_gain_ms"] = round(row["sum_ms"] - row["AB"]["ms"], 3)
        out["runs"].append(row)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
