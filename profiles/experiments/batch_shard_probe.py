"""k_reintegrate_blocks under sharding: the batch of harness/shard_emulation.py as rank 0 of `world` ranks (run under
rocprofv3 --kernel-trace --stats to see the kernel's own duration).  python batch_shard_probe.py <world>"""
import os
import sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import __graft_entry__ as ge
pkg = ge.load_package()
from dslam_amd.harness import synth
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n_map, K = 120, 32
wl = synth.s_street(640, 480)
frames = [wl.frame(i) for i in range(n_map)]
eng = pkg.open_engine(0)
scene = eng.create_scene(pkg.SceneParams(num_local_blocks=0x40000, **wl.scene_kwargs))
view = eng.create_view(wl.W, wl.H)
rs = eng.create_render_state(scene, wl.W, wl.H)
store = eng.create_frame_store(wl.W, wl.H, n_map)
eng.frame_store_enable_lists(store, scene)
for i, (rgba, mm, M) in enumerate(frames):
    eng.frame_store_put(store, i, rgba, mm)
    eng.view_update_from_store(view, store, i, timestamp=float(i))
    eng.process_frame(scene, view, rs, M, wl.intr)
    eng.frame_store_put_visible_list(store, i, scene, rs)
ids = list(range(n_map - K, n_map))
new = [synth.world_to_camera(wl.pose(i) @ synth.pose_matrix(synth.look_rotation(0.002 * (n + 1), 0.0), [0.01 * (n + 1), 0.0, 0.02])) for n, i in enumerate(ids)]
old = [frames[i][2] for i in ids]
if world > 1:
    eng.track_dirty(scene, True)
    eng.set_shard(scene, 0, world, 64)
eng.reintegrate_batch(scene, view, rs, store, ids, old, new, wl.intr)
eng.synchronize()
print("ok", world)
