"""Timings of the rows next to the hot path: view pre-processing, the keyframe store, the
OnlineCorrection batch with and without it, and the depth tracker.  Wall-clock around synchronous engine calls
(each number is the mean over `reps` calls after one warm-up call); prints one JSON line.

    python denseslam-global-consistency-h_amd/harness/side_bench.py [reps]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def timed(eng, fn, reps):
    fn()
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    eng.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    pkg = ge.load_package()
    from dslam_amd.harness import reintegrate, synth
    eng = pkg.open_engine(0)
    eng.set_async(True)  # calls return at once; `timed` synchronises around the batch
    wl = synth.s_street(640, 480)
    W, H = wl.W, wl.H
    n_frames = 24
    frames = [wl.frame(i) for i in range(n_frames)]
    p = pkg.SceneParams(**wl.scene_kwargs)
    scene = eng.create_scene(p)
    rs, view = eng.create_render_state(scene, W, H), eng.create_view(W, H)
    out = {"image": f"{W}x{H}", "reps": reps, "unit": "us per call (wall clock, engine idle before and after)"}

    rgba, mm, M = frames[0]
    bgr = np.ascontiguousarray(rgba[..., 2::-1])
    out["view_update_rgba_host"] = timed(eng, lambda: eng.view_update(view, rgba, mm), reps)
    out["view_update_bgr_host"] = timed(eng, lambda: eng.view_update_bgr(view, bgr, mm), reps)

    def upd_filtered():
        eng.view_update(view, rgba, mm, bilateral=True)
    out["view_update_rgba_host_bilateral"] = timed(eng, upd_filtered, reps)
    out["bilateral_filter_5_passes"] = out["view_update_rgba_host_bilateral"] - out["view_update_rgba_host"]
    Tpc = synth.pose_matrix(synth.look_rotation(0.01, 0.005), [0.02, -0.01, 0.3]).astype(np.float32)
    out["depth_post_processing_host"] = timed(eng, lambda: eng.depth_post_processing(frames[1][1], mm, Tpc, wl.intr, 0.05, 0.3), reps)

    # map + keyframe store
    db = reintegrate.FusionFrameDatabase(eng, W, H, n_frames)
    for i, (rgba_i, mm_i, M_i) in enumerate(frames):
        eng.view_update(view, rgba_i, mm_i, timestamp=float(i))
        db.insert_from_view(float(i), np.linalg.inv(np.asarray(M_i, np.float64)), view)
        eng.process_frame(scene, view, rs, M_i, wl.intr)
    eng.synchronize()
    out["frame_store_put_view"] = timed(eng, lambda: eng.frame_store_put_view(db.store, 0, view), reps)
    out["view_update_from_store"] = timed(eng, lambda: eng.view_update_from_store(view, db.store, 3), reps)

    # OnlineCorrection's inner pair (DeIntegrate at the old pose + Integrate at the new one) for 8 keyframes:
    # images from the store vs re-uploaded from the host as the reference does
    ks = list(range(8, 16))
    new_pose = {k: (np.asarray(frames[k][2], np.float64) @ synth.pose_matrix(synth.look_rotation(0.002, 0.001), [0.01, 0.0, 0.005])).astype(np.float32) for k in ks}

    def correction(from_store):
        for k in ks:
            if from_store:
                eng.view_update_from_store(view, db.store, db.entries[float(k)][1], timestamp=float(k))
            else:
                eng.view_update(view, frames[k][0], frames[k][1], timestamp=float(k))
            eng.deprocess_frame(scene, view, rs, frames[k][2], wl.intr)
            eng.process_frame(scene, view, rs, new_pose[k], wl.intr, is_defusion=True)
            # and back, so every repetition sees the same map
            eng.deprocess_frame(scene, view, rs, new_pose[k], wl.intr)
            eng.process_frame(scene, view, rs, frames[k][2], wl.intr, is_defusion=True)
    out["online_correction_8kf_x2_from_store"] = timed(eng, lambda: correction(True), max(3, reps // 10))
    out["online_correction_8kf_x2_host_upload"] = timed(eng, lambda: correction(False), max(3, reps // 10))

    # depth tracker on the S-room map (the indoor, ICP-sized case)
    wr = synth.s_room(640, 480)
    pr = pkg.SceneParams(**wr.scene_kwargs)
    sr = eng.create_scene(pr)
    rsr, vr = eng.create_render_state(sr, wr.W, wr.H), eng.create_view(wr.W, wr.H)
    for i in range(4):
        rgba_i, mm_i, M_i = wr.frame(i)
        eng.view_update(vr, rgba_i, mm_i, timestamp=float(i))
        eng.process_frame(sr, vr, rsr, M_i, wr.intr)
    M3 = wr.frame(3)[2]
    out["create_icp_maps_with_download"] = timed(eng, lambda: eng.create_icp_maps(sr, rsr, M3, wr.intr), max(3, reps // 5))
    out["create_icp_maps_device_only"] = timed(eng, lambda: eng.create_icp_maps(sr, rsr, M3, wr.intr, download=False), max(3, reps // 5))
    rgba4, mm4, _ = wr.frame(4)
    eng.view_update(vr, rgba4, mm4, timestamp=4.0)
    res_holder = {}

    def track():
        res_holder["r"] = eng.track_camera(vr, rsr, M3, M3, wr.intr)[1]
    out["track_camera"] = timed(eng, track, max(3, reps // 5))
    out["track_camera_iterations"] = res_holder["r"].iterations
    # meshing export (SaveCurrSceneToMesh) of the street map built above and of the room map
    import ctypes as C
    for tag, sc in (("street", scene), ("room", sr)):
        n = C.c_int(0)
        for colour in (0, 1):
            out[f"mesh_scene_{tag}" + ("_colour" if colour else "")] = timed(
                eng, lambda: eng._call("mesh_scene", eng._engine, sc.ptr, C.c_int(0), C.c_int(colour), C.byref(n)), max(3, reps // 10))
        out[f"mesh_triangles_{tag}"] = n.value
    print(json.dumps({k: (round(v, 2) if isinstance(v, float) else v) for k, v in out.items()}))


if __name__ == "__main__":
    main()
