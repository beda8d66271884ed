"""S-stress: every block of the pool allocated on a dense lattice and visible, so one integrate
launch streams the whole voxel block array (V = num_local_blocks; 2.15 GB read+write at the default 0x40000 blocks,
far past the 256 MiB Infinity Cache).  Used for the steady-state HBM roofline point of k_integrate.

    python -m ... stress  (see main) prints one JSON line.
"""
import json
import sys

import numpy as np


def build_lattice_state(pkg, n_side, num_buckets, num_excess):
    """Host-side hash table holding an n_side^3 lattice of blocks (positions offset so the camera sees them)."""
    n = n_side ** 3
    nb = num_buckets
    idx = np.arange(n)
    bx = (idx % n_side) - n_side // 2
    by = ((idx // n_side) % n_side) - n_side // 2
    bz = (idx // (n_side * n_side)) + 40  # lattice starts 40 blocks in front of the camera
    h = ((bx.astype(np.int64) * 73856093) ^ (by.astype(np.int64) * 19349669) ^ (bz.astype(np.int64) * 83492791)) & (nb - 1)
    table = np.zeros(nb + num_excess, dtype=pkg.HASH_ENTRY_DTYPE)
    table["ptr"] = -2
    tail = np.full(nb, -1, dtype=np.int64)  # last entry index of each bucket chain
    next_excess = 0
    for i in range(n):
        b = int(h[i])
        if tail[b] < 0:
            t = b
        else:
            t = nb + next_excess
            table["offset"][tail[b]] = next_excess + 1
            next_excess += 1
            if next_excess > num_excess:
                raise RuntimeError("excess list too small for the lattice")
        table["pos"][t] = (bx[i], by[i], bz[i])
        table["ptr"][t] = i
        tail[b] = t
    visible = np.nonzero(table["ptr"] >= 0)[0].astype(np.int32)
    # pools: every voxel-block slot is used; remaining excess slots stay on the free list in ascending order
    excess_list = np.arange(num_excess, dtype=np.int32)
    free_ex = np.arange(next_excess, num_excess, dtype=np.int32)
    excess_list[:len(free_ex)] = free_ex[::-1]
    return table, visible, excess_list, len(free_ex) - 1


def run(pkg, eng, n_side=64, iterations=20, W=640, H=480, slots_in_list_order=False, as_process_frame=False):
    """slots_in_list_order: the r-th visible entry holds voxel-block slot r (the blocks are then visited in memory order);
    default: slots in lattice order = random with respect to the list (hash) order -- the worse case for HBM page locality.
    as_process_frame: the kernel as dslam_process_frame launches it -- the frame's visible list queued on the ring in the
    same launch, as every per-frame fusion has it (the production launch); default: as dslam_integrate_into_scene launches
    it (no ring push).  The frustum then ends in front of the 30 m depth plane, so that ProcessFrame's allocation pass marks
    nothing on the full pool (integration has no frustum test: every voxel is still updated)."""
    n = n_side ** 3
    params = pkg.SceneParams(voxel_size=0.01, mu=0.04, max_w=100, frustum_min=0.2, frustum_max=20.0 if as_process_frame else 100.0,
                             num_local_blocks=n, num_buckets=0x100000, num_excess=0x20000)
    scene = eng.create_scene(params)
    rs = eng.create_render_state(scene, W, H)
    view = eng.create_view(W, H)
    table, visible, excess_list, last_free_ex = build_lattice_state(pkg, n_side, params.num_buckets, params.num_excess)
    if slots_in_list_order:
        table["ptr"][visible] = np.arange(len(visible), dtype=np.int32)
    eng.upload_scene_state(scene, hash_table=table, allocation_list=np.arange(n, dtype=np.int32), last_free_block_id=-1,
                           excess_list=excess_list, last_free_excess_id=last_free_ex)
    eng.upload_visible_ids(rs, visible)
    # constant far depth: every voxel in front of it is updated (eta > mu -> newF = 1), none is skipped
    depth_mm = np.full((H, W), 30000, dtype=np.int16)
    rgba = np.full((H, W, 4), 128, dtype=np.uint8)
    eng.view_update(view, rgba, depth_mm)
    M = np.eye(4, dtype=np.float32)
    intr = np.array([100.0, 100.0, (W - 1) / 2.0, (H - 1) / 2.0], np.float32)  # wide FOV: the lattice projects inside
    # one warm-up launch, then `iterations` launches timed one by one with events attached to the dispatch packets
    # (the kernel's own start-to-end interval, what rocprofv3 reports; bench.py times the headline launch the same way)
    call = (lambda: eng.process_frame(scene, view, rs, M, intr)) if as_process_frame else (lambda: eng.integrate_into_scene(scene, view, rs, M, intr))
    call()
    call()
    eng.synchronize()
    eng.kernel_timer_enable(True)
    for _ in range(iterations):
        call()
    total_ms, launches, blocks = eng.kernel_timer_read()
    eng.kernel_timer_enable(False)
    ms, nvis = total_ms / max(1, launches), blocks // max(1, launches)
    vox = eng.download_voxel_blocks(scene, 0, 64)
    updated = float((vox["w_depth"] > 0).mean())
    alg_bytes = 8212.0 * nvis + 8.0 * W * H
    return {"workload": f"S-stress lattice {n_side}^3" + (", slots in list order" if slots_in_list_order else "") +
                        (", launched by ProcessFrame (visible-list ring push on)" if as_process_frame else ", launched by IntegrateIntoScene (no ring push)"),
            "visible_blocks": nvis, "ms_per_launch": ms,
            "algorithmic_bytes_per_launch": alg_bytes, "achieved_GBps": alg_bytes / (ms * 1e-3) / 1e9,
            "frac_of_8TBps": alg_bytes / (ms * 1e-3) / 1e9 / 8000.0, "voxels_updated_frac_sample": updated,
            "iterations": iterations}


if __name__ == "__main__":
    sys.path.insert(0, ".")
    import __graft_entry__ as ge
    pkg = ge.load_package()
    eng = pkg.open_engine(0)
    n_side = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    out = run(pkg, eng, n_side=n_side, as_process_frame=True)
    out["no_push"] = run(pkg, pkg.open_engine(0), n_side=n_side)
    out["slots_in_list_order"] = run(pkg, pkg.open_engine(0), n_side=n_side, slots_in_list_order=True, as_process_frame=True)
    print(json.dumps(out))
