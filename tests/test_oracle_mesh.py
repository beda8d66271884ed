"""The meshing export on the CPU oracle (SaveCurrSceneToMesh -> ITMMeshingEngine::MeshScene, reference
DenseSlam.cpp:638-643).  The reference holds no mesh fixture (parity unpinned), so the mesh is
checked against the map it comes from: vertices sit on cube edges, the surface coincides with what the raycaster
sees, triangles face one way, the triangle list saturates like upstream's, cubes next to missing blocks are skipped."""
import numpy as np
import pytest

import util


def _fused(pkg, synth, api, frames=4, **over):
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl, **over)
    s, rs, v = util.run_sequence(api, pkg, wl, p, frames)
    return wl, p, s, rs, v


def _candidate_cubes(pos, voxel):
    """[n, 8, 3] integer (0,0,0)-corners of the cubes a triangle can belong to: the cube holding its centroid, or --
    when the centroid sits on a cube face (a corner sample was exactly 0) -- the cubes on either side."""
    c = pos.astype(np.float64).mean(1) / voxel
    lo, hi = np.floor(c - 1e-3).astype(int), np.floor(c + 1e-3).astype(int)
    pick = np.array([[(k >> a) & 1 for a in range(3)] for k in range(8)])
    return np.where(pick[None] == 1, hi[:, None, :], lo[:, None, :])


def test_vertices_lie_on_cube_edges_and_mesh_is_deterministic(pkg, synth, oracle):
    wl, p, s, rs, v = _fused(pkg, synth, oracle)
    pos, col = oracle.mesh_scene(s, colour=True)
    assert len(pos) > 10000 and pos.dtype == np.float32 and col.shape == pos.shape
    again, _ = oracle.mesh_scene(s)
    assert np.array_equal(pos, again)
    # a vertex is a point on an axis-aligned cube edge: at least two of its coordinates are whole voxels
    g = pos.reshape(-1, 3).astype(np.float64) / p.voxel_size
    whole = np.abs(g - np.round(g)) < 1e-3
    assert (whole.sum(1) >= 2).all()
    assert col.min() >= 0.0 and col.max() <= 1.0 and col.std() > 0.05  # the procedural texture came through


def test_mesh_surface_is_the_surface_the_raycaster_sees(pkg, synth, oracle):
    scipy_spatial = pytest.importorskip("scipy.spatial")
    wl, p, s, rs, v = _fused(pkg, synth, oracle)
    pos, _ = oracle.mesh_scene(s)
    M = wl.frame(3)[2]
    pts, nrm = oracle.create_icp_maps(s, rs, M, wl.intr)
    hit = pts[..., 3] > 0
    assert hit.mean() > 0.5
    tree = scipy_spatial.cKDTree(pos.reshape(-1, 3))
    d, _ = tree.query(pts[hit][:, :3])
    # every raycast hit has a mesh vertex within one voxel diagonal (both are the TSDF zero crossing)
    assert np.percentile(d, 99) < 1.8 * p.voxel_size and d.max() < 3.0 * p.voxel_size
    # orientation: with the table's winding the geometric normal points from the free side (sdf > 0) into the surface,
    # i.e. away from the camera that observed it, for essentially every triangle
    n = np.cross(pos[:, 1] - pos[:, 0], pos[:, 2] - pos[:, 0])
    cam = np.linalg.inv(M)[:3, 3]
    away = ((pos.mean(1) - cam) * n).sum(1)
    ok = np.abs(away) > 1e-12
    frac = (away[ok] > 0).mean()
    assert frac > 0.97 or frac < 0.03, frac


def test_triangle_list_saturates_like_upstream(pkg, synth, oracle):
    wl, p, s, rs, v = _fused(pkg, synth, oracle, frames=2)
    full, _ = oracle.mesh_scene(s)
    assert len(full) > 500
    capped, _ = oracle.mesh_scene(s, max_triangles=100)
    assert len(capped) == 99 and np.array_equal(capped, full[:99])  # `if (n < noMaxTriangles - 1) n++`
    one, _ = oracle.mesh_scene(s, max_triangles=1)
    assert len(one) == 0
    big, _ = oracle.mesh_scene(s, max_triangles=len(full) + 1)
    assert np.array_equal(big, full)


def test_empty_scene_and_order(pkg, synth, oracle):
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl)
    s = oracle.create_scene(p)
    pos, col = oracle.mesh_scene(s, colour=True)
    assert pos.shape == (0, 3, 3) and col.shape == (0, 3, 3)
    # order = hash entries ascending: the block of every triangle, looked up again, has a non-decreasing entry index
    s, rs, v = util.run_sequence(oracle, pkg, wl, p, 2)
    pos, _ = oracle.mesh_scene(s)
    h = oracle.download_hash_table(s)
    entry_of = {tuple(h["pos"][i][:3]): i for i in np.nonzero(h["ptr"] >= 0)[0]}
    cur = 0
    for cands in np.floor_divide(_candidate_cubes(pos, p.voxel_size), 8):
        idx = sorted(entry_of[tuple(b)] for b in cands if tuple(b) in entry_of)
        nxt = [i for i in idx if i >= cur]
        assert nxt, "triangle belongs to a block that precedes the previous triangle's block"
        cur = nxt[0]


def test_cubes_touching_missing_blocks_are_skipped(pkg, synth, oracle):
    """With swapping, blocks parked on the host have ptr = -1: they produce no triangles and neither do the cubes of
    resident blocks that reach into them (findPointNeighbors fails)."""
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl, use_swapping=1)
    s, rs, v = util.run_sequence(oracle, pkg, wl, p, 3)
    before, _ = oracle.mesh_scene(s)
    # look away so the next frames swap the first blocks out
    far = wl.frame(0)[2].copy()
    far[:3, 3] += np.array([30.0, 0.0, 0.0], np.float32)
    rgba, mm, _ = wl.frame(0)
    for i in range(3):
        oracle.view_update(v, rgba, np.zeros_like(mm), timestamp=10.0 + i)
        oracle.process_frame(s, v, rs, far, wl.intr)
    h = oracle.download_hash_table(s)
    assert (h["ptr"] == -1).sum() > 0, "nothing was swapped out"
    after, _ = oracle.mesh_scene(s)
    assert len(after) < len(before)
    live = {tuple(x[:3]) for x in h["pos"][h["ptr"] >= 0]}
    for cands in _candidate_cubes(after, p.voxel_size):
        # some cube this triangle can come from has both its near and its far corner in resident blocks
        assert any(tuple(b // 8) in live and tuple((b + 1) // 8) in live for b in cands)
