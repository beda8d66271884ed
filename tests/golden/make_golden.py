"""Generates tests/golden/s_tiny_3frames.npz with the CPU oracle.

The reference ships no golden vectors for this path (parity unpinned), so these fixtures pin the
ORACLE's behaviour over time (any later change to oracle or kernels that alters results is caught), not the
reference's.  Inputs are the synthetic S-tiny frames; outputs are the map state after 3 fused frames and the raycast.

    python tests/golden/make_golden.py
"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

N_FRAMES = 3
N_BLOCKS = 48


def run(api, pkg, synth, inputs=None):
    """Fuse the frames and raycast.  With `inputs` (a loaded fixture) the stored frames / poses are replayed, so the
    check does not depend on regenerating bit-identical synthetic images."""
    wl = synth.s_tiny()
    params = pkg.SceneParams(num_local_blocks=0x800, num_buckets=0x1000, num_excess=0x400, **wl.scene_kwargs)
    s = api.create_scene(params)
    rs = api.create_render_state(s, wl.W, wl.H)
    v = api.create_view(wl.W, wl.H)
    if inputs is not None:
        frames = [(inputs["rgba"][i], inputs["depth_mm"][i], inputs["poses"][i]) for i in range(N_FRAMES)]
    else:
        frames = [wl.frame(i) for i in range(N_FRAMES)]
    stats = []
    for i, (rgba, mm, M) in enumerate(frames):
        api.view_update(v, rgba, mm, timestamp=float(i))
        api.process_frame(s, v, rs, M, wl.intr)
        st = api.stats(s, rs)
        stats.append([st["last_free_block_id"], st["last_free_excess_id"], st["no_visible_entries"]])
    h = api.download_hash_table(s)
    occ = np.nonzero(h["ptr"] >= -1)[0].astype(np.int32)
    vis = api.download_visible_ids(rs)
    top = params.num_local_blocks - N_BLOCKS
    vox = api.download_voxel_blocks(s, top, N_BLOCKS)  # the first N_BLOCKS slots handed out (top of the free stack)
    M_last = frames[-1][2]
    rs_free = api.create_render_state(s, wl.W, wl.H)
    depth = api.get_image(s, rs_free, M_last, wl.intr, pkg.IMAGE_DEPTH)
    colour = api.get_image(s, rs_free, M_last, wl.intr, pkg.IMAGE_COLOUR_FROM_VOLUME)
    shaded = api.get_image(s, rs_free, M_last, wl.intr, pkg.IMAGE_SHADED)
    rng = api.download_range_image(rs_free)[:(wl.H + 7) // 8, :(wl.W + 7) // 8]
    # meshing export (SaveCurrSceneToMesh): triangle count, the first triangles, and a digest of the whole mesh
    mesh_pos, mesh_col = api.mesh_scene(s, colour=True)
    digest = hashlib.sha256(mesh_pos.tobytes() + mesh_col.tobytes()).digest()
    return dict(
        rgba=np.stack([f[0] for f in frames]), depth_mm=np.stack([f[1] for f in frames]),
        poses=np.stack([f[2] for f in frames]), intr=np.asarray(wl.intr, np.float32),
        scene_params=np.array([params.voxel_size, params.mu, params.max_w, params.frustum_min, params.frustum_max,
                               params.num_local_blocks, params.num_buckets, params.num_excess], np.float64),
        stats=np.array(stats, np.int32), occupied_idx=occ, occupied_pos=h["pos"][occ], occupied_ptr=h["ptr"][occ],
        occupied_offset=h["offset"][occ], visible_ids=vis, voxels_top=vox.view(np.uint64).reshape(N_BLOCKS, 512),
        raycast_depth=depth, raycast_colour=colour, raycast_shaded=shaded, range_corner=rng,
        mesh_count=np.array([len(mesh_pos)], np.int32), mesh_head=mesh_pos[:256].copy(), mesh_head_colour=mesh_col[:256].copy(),
        mesh_sha256=np.frombuffer(digest, np.uint8).copy())


if __name__ == "__main__":
    pkg = ge.load_package()
    from dslam_amd.harness import synth
    api = ge.load_oracle().open_oracle(pkg.CApi)
    out = run(api, pkg, synth)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "s_tiny_3frames.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
