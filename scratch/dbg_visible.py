import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import __graft_entry__ as ge, util
pkg = ge.load_package(); from dslam_amd.harness import synth
gpu = pkg.open_engine(0); orc = ge.load_oracle().open_oracle(pkg.CApi)
big = len(sys.argv) > 1
if big:
    wlb = synth.s_room(); pb = pkg.SceneParams(num_local_blocks=0x10000, **wlb.scene_kwargs)
    util.run_sequence(gpu, pkg, wlb, pb, 1)
wl = synth.s_tiny(96, 72); p = util.small_params(pkg, wl)
out = {}
for name, api in (("gpu", gpu), ("oracle", orc), ("gpu2", gpu)):
    s, rs, v = util.run_sequence(api, pkg, wl, p, 4)
    M = synth.world_to_camera(wl.pose(2) @ synth.pose_matrix(synth.look_rotation(0.05, -0.03), [0.03, 0.01, -0.02]))
    img = api.get_image(s, rs, M, wl.intr, pkg.IMAGE_DEPTH)
    out[name] = (api.download_visible_ids(rs), api.download_hash_table(s), api.download_range_image(rs), img)
v0, h0, r0, i0 = out["gpu"]; v1, h1, r1, i1 = out["oracle"]; v2, h2, r2, i2 = out["gpu2"]
print("hash equal", np.array_equal(h0, h1), len(v0), len(v1), len(v2))
print("gpu-only", np.setdiff1d(v0, v1), "oracle-only", np.setdiff1d(v1, v0), "gpu vs gpu2", np.array_equal(v0, v2))
print("range equal", np.array_equal(r0, r1), "img equal", np.array_equal(i0, i1), np.abs(i0-i1).max())
if not np.array_equal(r0, r1):
    bad = np.argwhere(r0 != r1); print(len(bad), bad[:10], r0[tuple(bad[0][:2])], r1[tuple(bad[0][:2])])
