"""Map checkpoint / restore through the C ABI's bulk read-back and upload calls (harness/checkpoint.py): a restored
map is the same bytes, and fusion continues from it exactly as from the original -- also across engines (saved on
the HIP engine, restored into the CPU oracle and the other way round)."""
import numpy as np
import pytest

import util

# the visible-list history of decay / sliding window is not part of a map checkpoint: a restored map starts a new one
RING_STATS = ("frame_counter", "fusion_fifo_len", "defusion_fifo_len")


def _fuse(api, wl, scene, rs, view, frames):
    for i in frames:
        rgba, mm, M = wl.frame(i)
        api.view_update(view, rgba, mm, timestamp=float(i))
        api.process_frame(scene, view, rs, M, wl.intr)


def _roundtrip(pkg, synth, src, dst, tmp_path):
    from dslam_amd.harness import checkpoint
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl)
    s0 = src.create_scene(p)
    rs0, v0 = src.create_render_state(s0, wl.W, wl.H), src.create_view(wl.W, wl.H)
    _fuse(src, wl, s0, rs0, v0, range(4))
    info = checkpoint.save_map(src, s0, tmp_path / "map.npz")
    assert info["used_blocks"] > 300
    s1 = checkpoint.load_map(dst, pkg, tmp_path / "map.npz")
    a, b = util.snapshot(src, s0), util.snapshot(dst, s1)
    util.assert_same_state(a, b, "restored map", ignore_stats=RING_STATS)
    # both continue with the same frames; the restored side has a fresh render state, so the first frame re-derives
    # the visible list -- give the original a fresh one too (same situation as the reference creating a new view)
    rs0b = src.create_render_state(s0, wl.W, wl.H)
    rs1, v1 = dst.create_render_state(s1, wl.W, wl.H), dst.create_view(wl.W, wl.H)
    _fuse(src, wl, s0, rs0b, v0, range(4, 8))
    _fuse(dst, wl, s1, rs1, v1, range(4, 8))
    util.assert_same_state(util.snapshot(src, s0, rs0b), util.snapshot(dst, s1, rs1), "continued after restore",
                           ignore_stats=RING_STATS)
    d0 = src.get_image(s0, src.create_render_state(s0, wl.W, wl.H), wl.frame(7)[2], wl.intr, pkg.IMAGE_DEPTH)
    d1 = dst.get_image(s1, dst.create_render_state(s1, wl.W, wl.H), wl.frame(7)[2], wl.intr, pkg.IMAGE_DEPTH)
    assert np.abs(d0 - d1).max() <= 1e-4 and (d0 > 0).sum() > 500


def test_checkpoint_roundtrip_oracle(pkg, synth, oracle, tmp_path):
    _roundtrip(pkg, synth, oracle, oracle, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("direction", ["gpu_to_gpu", "gpu_to_oracle", "oracle_to_gpu"])
def test_checkpoint_roundtrip_gpu(pkg, synth, gpu, oracle, tmp_path, direction):
    src, dst = {"gpu_to_gpu": (gpu, gpu), "gpu_to_oracle": (gpu, oracle), "oracle_to_gpu": (oracle, gpu)}[direction]
    _roundtrip(pkg, synth, src, dst, tmp_path)
