"""Keyframe store + DenseSlam::OnlineCorrection (reference DenseSlam.cpp:156-158, 298-432).

 * the keyframe store keeps RGB + depth of every fused keyframe in HBM; its contents are byte-exact copies;
 * the Python scheduler (harness/reintegrate.py FusionFrameDatabase) driven over the HIP engine and over the CPU
   oracle ends in byte-identical maps;
 * the C++ scheduler a maintainer links against (itmlib/DenseSLAM/OnlineCorrection.h) run by the driver harness
   picks the same keyframes in the same order and ends in the same map as the oracle replaying that schedule."""
import ctypes as C
import os
import struct
import subprocess

import numpy as np
import pytest

import util
from test_gpu_itmlib_shim import HARNESS, fnv1a

pytestmark = pytest.mark.gpu


def _inv_abi(oracle, M):
    """float32 cofactor inverse with the engine's operation order (row-major numpy in / out)."""
    m = np.ascontiguousarray(np.asarray(M, np.float32).T).ravel()
    out = np.zeros(16, np.float32)
    assert oracle.lib.oracle_invert_matrix(m.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.POINTER(C.c_float))) == 0
    return out.reshape(4, 4).T.copy()


def _perturbed(synth, Twc, k):
    d = synth.pose_matrix(synth.look_rotation(0.004 * k, -0.002 * k), [0.003 * k, -0.002 * k, 0.004 * k])
    return (np.asarray(Twc, np.float64) @ d).astype(np.float32)


def test_frame_store_round_trip(gpu, oracle, synth):
    wl = synth.s_tiny()
    for api in (gpu, oracle):
        fs = api.create_frame_store(wl.W, wl.H, 3)
        v = api.create_view(wl.W, wl.H)
        rgba0, mm0, _ = wl.frame(0)
        rgba1, mm1, _ = wl.frame(1)
        api.frame_store_put(fs, 0, rgba0, mm0)
        api.frame_store_put_bgr(fs, 2, rgba1[..., 2::-1], mm1)
        api.view_update(v, rgba1, mm1)
        api.frame_store_put_view(fs, 1, v)
        got = [api.frame_store_get(fs, s) for s in range(3)]
        assert np.array_equal(got[0][0], rgba0) and np.array_equal(got[0][1], mm0)
        assert np.array_equal(got[1][0], rgba1) and np.array_equal(got[1][1], mm1)
        want = rgba1.copy(); want[..., 3] = 255
        assert np.array_equal(got[2][0], want) and np.array_equal(got[2][1], mm1)
        api.view_update_from_store(v, fs, 0)
        assert np.array_equal(api.download_view_rgba(v), rgba0)
        d = api.download_view_depth(v)
        assert np.array_equal(d[mm0 > 0], mm0[mm0 > 0].astype(np.float32) * np.float32(0.001))
        with pytest.raises(Exception):
            api.frame_store_put(fs, 3, rgba0, mm0)


def _keyframe_sets(synth, wl, n_frames):
    """ORB-SLAM2's keyframes as seen at each fused frame: all previous keyframes, some with optimised (moved) poses,
    keyframe 3 reported bad once.  Keyframe 4 never reaches ORB-SLAM2's map, so the pass that follows its insertion
    takes it out again (flaginfo stays 0: DenseSlam.cpp:412-426); a keyframe that disappears later (2, from frame 5
    on) stays, because the reference never clears flaginfo."""
    Twc = [np.linalg.inv(np.asarray(wl.frame(i)[2], np.float64)).astype(np.float32) for i in range(n_frames)]
    current = [t.copy() for t in Twc]
    sets = []
    for i in range(n_frames):
        if i >= 3:  # a "bundle adjustment" moved a few keyframes by distinct amounts
            for rank, j in enumerate(range(max(0, i - 4), i)):
                current[j] = _perturbed(synth, current[j], 1 + 1.37 * rank + 0.61 * (i % 3) + 0.083 * j)
        kfs = []
        for j in range(i + 1):
            if (j == 2 and i >= 5) or j == 4:
                continue
            kfs.append((float(j), current[j].copy(), j == 3 and i == 6))
        sets.append(kfs)
    return Twc, sets


def test_python_scheduler_gpu_matches_oracle(pkg, synth, gpu, oracle):
    from dslam_amd.harness import reintegrate
    wl = synth.s_tiny()
    n_frames = 9
    p = util.small_params(pkg, wl)
    Twc, sets = _keyframe_sets(synth, wl, n_frames)
    logs, snaps = {}, {}
    for name, api in (("gpu", gpu), ("oracle", oracle)):
        s = api.create_scene(p)
        rs, v = api.create_render_state(s, wl.W, wl.H), api.create_view(wl.W, wl.H)
        db = reintegrate.FusionFrameDatabase(api, wl.W, wl.H, n_frames, pose_to_M=lambda T: _inv_abi(oracle, T))
        log = []
        for i in range(n_frames):
            rgba, mm, _ = wl.frame(i)
            api.view_update(v, rgba, mm, timestamp=float(i))
            slot = db.insert_from_view(float(i), Twc[i], v)
            order, culled = db.online_correction(s, v, rs, wl.intr, sets[i], correction_num=2, start_to_correction_num=3)
            if float(i) in db.entries:
                api.view_update_from_store(v, db.store, slot, timestamp=float(i))
                api.process_frame(s, v, rs, _inv_abi(oracle, Twc[i]), wl.intr)
            if len(db) > 6:
                api.slide_window(s, rs, 6)
                for _ in range(2):
                    api.slide_window_defusion_part(s, rs, 6, (6 - 3) * 2)
                db.slide_window_pose(6)
            log.append((order, culled, len(db)))
        logs[name], snaps[name] = log, util.snapshot(api, s, rs)
    assert logs["gpu"] == logs["oracle"]
    assert sum(len(o) for o, _, _ in logs["gpu"]) >= 8, "the schedule should re-fuse keyframes"
    assert [c for _, c, _ in logs["gpu"] if c] == [[4.0]], "keyframe 4 (never in ORB-SLAM2's map) is culled, 2 is not"
    util.assert_same_state(snaps["gpu"], snaps["oracle"], "after online correction")


@pytest.mark.parametrize("batched", [False, True], ids=["per_keyframe_calls", "one_batch_call"])
def test_cpp_scheduler_matches_oracle_replay(pkg, synth, oracle, tmp_path, batched):
    """batched: FusionFrameDataBase::OnlineCorrectionBatched -- the re-fusions of a correction as ONE dslam_reintegrate_batch call,
    de-integrating from the visible lists kept with the keyframes -- against the oracle's replay of the same schedule."""
    from dslam_amd.harness import reintegrate
    assert os.path.exists(HARNESS), "run python __graft_entry__.py (build) first"
    wl = synth.s_tiny()
    n_frames, corr_num, start_num, max_age = 9, 2, 3, 6
    p = util.small_params(pkg, wl, num_local_blocks=0x800, num_buckets=0x1000, num_excess=0x400)
    frames = [wl.frame(i) for i in range(n_frames)]
    # the harness derives Twc = M^-1 with the float32 cofactor inverse; hand ORB-SLAM2's keyframes the same bits
    Twc = [_inv_abi(oracle, M) + np.float32(0.0) for _, _, M in frames]
    _, sets = _keyframe_sets(synth, wl, n_frames)
    for i, kfs in enumerate(sets):  # unmoved keyframes must carry exactly the fused pose
        sets[i] = [(ts, (Twc[int(ts)] if np.array_equal(T, np.linalg.inv(np.asarray(frames[int(ts)][2], np.float64)).astype(np.float32)) else T), bad)
                   for ts, T, bad in kfs]
    fin, fkf, fout = tmp_path / "frames.bin", tmp_path / "keyframes.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        f.write(struct.pack("<3i", wl.W, wl.H, n_frames))
        for rgba, mm, M in frames:
            f.write(rgba.tobytes()); f.write(mm.tobytes()); f.write(pkg.mat_to_abi(M).tobytes())
        f.write(np.asarray(wl.intr, np.float32).tobytes())
        f.write(struct.pack("<4f", p.voxel_size, p.mu, p.frustum_min, p.frustum_max))
        f.write(struct.pack("<4i", p.max_w, p.num_local_blocks, p.num_buckets, p.num_excess))
    with open(fkf, "wb") as f:
        f.write(struct.pack("<2i", corr_num, start_num))
        for kfs in sets:
            f.write(struct.pack("<i", len(kfs)))
            for ts, T, bad in kfs:
                f.write(struct.pack("<d", ts)); f.write(pkg.mat_to_abi(T).tobytes()); f.write(struct.pack("<i", int(bad)))
    env = dict(os.environ, DRIVER_HARNESS_BATCHED="1") if batched else dict(os.environ)
    res = subprocess.run([HARNESS, str(fin), str(fout), "1", str(max_age), str(fkf)], capture_output=True, text=True, timeout=120, env=env)
    assert res.returncode == 0, res.stdout + res.stderr

    raw = open(fout, "rb").read()
    npx = wl.W * wl.H
    off = 32 + npx * 8 + 64  # header, depth + colour images, tracked pose
    cpp_log = []
    for i in range(n_frames):
        (n,) = struct.unpack_from("<i", raw, off); off += 4
        order = list(struct.unpack_from(f"<{n}d", raw, off)); off += 8 * n
        culled, size = struct.unpack_from("<2i", raw, off); off += 8
        cpp_log.append((order, culled, size))
    assert off + 40 == len(raw)  # the rest is the trailer (raycast image sums + host mirrors)

    # the oracle replays DenseSlam::ProcessFrame with the Python scheduler (float64 pose errors, same rules)
    s = oracle.create_scene(p)
    rs, v = oracle.create_render_state(s, wl.W, wl.H), oracle.create_view(wl.W, wl.H)
    to_M = lambda T: _inv_abi(oracle, np.asarray(T, np.float32) + np.float32(0.0))
    db = reintegrate.FusionFrameDatabase(oracle, wl.W, wl.H, n_frames, pose_to_M=to_M)
    if batched:
        db.enable_visible_lists(s)
    py_log = []
    for i in range(n_frames):
        rgba, mm, _ = frames[i]
        oracle.view_update(v, rgba, mm, timestamp=float(i))
        slot = db.insert_from_view(float(i), Twc[i], v)
        # unmoved keyframes give pose differences of rounding size (never exactly the identity in the C++ float
        # arithmetic either, unless bit-identical); they rank last and are never selected with these parameters
        order, culled = db.online_correction(s, v, rs, wl.intr, sets[i], corr_num, start_num, batched=batched)
        if float(i) in db.entries:
            oracle.view_update_from_store(v, db.store, slot, timestamp=float(i))
            oracle.process_frame(s, v, rs, to_M(Twc[i]), wl.intr)
            if batched:
                db.keep_visible_list(float(i), s, rs)
        if len(db) > max_age:
            oracle.slide_window(s, rs, max_age)
            for _ in range(corr_num):
                oracle.slide_window_defusion_part(s, rs, max_age, (max_age - start_num) * corr_num)
            db.slide_window_pose(max_age)
        oracle.decay(s, rs, 1, 2, True)
        py_log.append((order, len(culled), len(db)))
    assert cpp_log == py_log
    assert sum(len(o) for o, _, _ in cpp_log) >= 8 and sum(c for _, c, _ in cpp_log) == 1

    st = oracle.stats(s, rs)
    last_free, no_vis, used_bytes, decayed = struct.unpack_from("<4i", raw, 0)
    h_hash, h_vox = struct.unpack_from("<2Q", raw, 16)
    assert last_free == st["last_free_block_id"] and no_vis == st["no_visible_entries"] and decayed == st["decayed_block_count"]
    assert h_hash == fnv1a(oracle.download_hash_table(s).tobytes())
    assert h_vox == fnv1a(oracle.download_voxel_blocks(s).tobytes())
