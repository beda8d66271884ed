"""Pipelining frames across PCIe (dslam_engine_set_async + page-locked caller images + fences): uploads run on the
engine's copy stream under the previous frame's kernels, the render kernel stores the output image in page-locked
host memory, a fence tells the caller when.  Results must equal the synchronous call sequence of the reference's
driver (InfiniTamDriver.cpp:280-288, DenseSlam.cpp:210-232, DenseSlam.h:146-164) byte for byte."""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("workload,n_frames", [("s_tiny", 14), ("s_street", 6), ("s_street", 48)])   # (48: the bench's loop at length, every image compared)
def test_pipelined_frames_equal_synchronous_calls(pkg, synth, gpu, workload, n_frames):
    wl = synth.s_tiny() if workload == "s_tiny" else synth.s_street(640, 480)
    p = util.small_params(pkg, wl) if workload == "s_tiny" else pkg.SceneParams(num_local_blocks=0x8000 if n_frames < 20 else 0x20000, **wl.scene_kwargs)
    frames = [wl.frame(i) for i in range(n_frames)]

    # reference behaviour: every call synchronous, pageable images
    s0 = gpu.create_scene(p)
    rs0, free0, v0 = gpu.create_render_state(s0, wl.W, wl.H), gpu.create_render_state(s0, wl.W, wl.H), gpu.create_view(wl.W, wl.H)
    want = []
    for i, (rgba, mm, M) in enumerate(frames):
        gpu.view_update(v0, rgba, mm, timestamp=float(i))
        gpu.process_frame(s0, v0, rs0, M, wl.intr)
        want.append(gpu.get_image(s0, free0, M, wl.intr, pkg.IMAGE_DEPTH))
    ref = util.snapshot(gpu, s0, rs0)

    # pipelined: page-locked inputs (all frames, as a capture ring would hold them), a ring of 3 page-locked outputs
    s1 = gpu.create_scene(p)
    rs1, free1, v1 = gpu.create_render_state(s1, wl.W, wl.H), gpu.create_render_state(s1, wl.W, wl.H), gpu.create_view(wl.W, wl.H)
    rgba_p = gpu.host_alloc((n_frames, wl.H, wl.W, 4), np.uint8)
    mm_p = gpu.host_alloc((n_frames, wl.H, wl.W), np.int16)
    for i, (rgba, mm, M) in enumerate(frames):
        rgba_p[i], mm_p[i] = rgba, mm
    R = 3
    out_p = gpu.host_alloc((R, wl.H, wl.W), np.float32)
    fences = [gpu.fence_create() for _ in range(R)]
    got = [None] * n_frames
    gpu.set_async(True)
    try:
        for i, (rgba, mm, M) in enumerate(frames):
            if i >= R:  # the consumer takes image i - R before its buffer is rendered into again
                gpu.fence_wait(fences[i % R])
                got[i - R] = out_p[i % R].copy()
            gpu.view_update(v1, rgba_p[i], mm_p[i], timestamp=float(i))
            gpu.process_frame(s1, v1, rs1, M, wl.intr)
            gpu.get_image(s1, free1, M, wl.intr, pkg.IMAGE_DEPTH, out=out_p[i % R])
            gpu.fence_record(fences[i % R])
        for i in range(max(0, n_frames - R), n_frames):
            gpu.fence_wait(fences[i % R])
            assert gpu.fence_query(fences[i % R])
            got[i] = out_p[i % R].copy()
        gpu.synchronize()
    finally:
        gpu.set_async(False)
    util.assert_same_state(util.snapshot(gpu, s1, rs1), ref, "pipelined vs synchronous")
    for i in range(n_frames):
        assert np.array_equal(got[i], want[i]), f"output image of frame {i}"
    assert (want[-1] > 0).sum() > 500
    for f in fences:
        f.close()
    for a in (rgba_p, mm_p, out_p):
        gpu.host_free(a)


def test_fence_that_was_never_recorded_has_passed(gpu):
    f = gpu.fence_create()
    gpu.fence_wait(f)
    assert gpu.fence_query(f)
    f.close()


@pytest.mark.parametrize("pattern", ["fence_destroyed_while_lent", "view_read_after_the_fence", "one_fence_for_all",
                                     "bilateral_filter_behind_the_fence"])
def test_callers_fence_as_consumed_mark(pkg, synth, gpu, pattern):
    """The pipelined upload takes a caller's fence recorded behind the last view-reading call as the landing buffer's
    "consumed" mark instead of recording an event of its own (dslam_engine::last_fence).  Call patterns around that:
    the fence is destroyed while a view still waits on its event; a call reads the view AFTER the fence was recorded
    (the fence then says nothing about that read); one fence re-recorded every frame; the update itself enqueues a kernel
    that reads the landing buffer (the bilateral filter) and two updates follow each other with no other view-reading call
    in between (round-2 ADVICE: that kernel must count as a read, or the fence recorded before it is taken as the buffer's
    "consumed" mark and the next refill may land under the filter)."""
    wl = synth.s_tiny()
    p = util.small_params(pkg, wl)
    n_frames = 12
    frames = [wl.frame(i) for i in range(n_frames)]
    s0 = gpu.create_scene(p)
    rs0, v0 = gpu.create_render_state(s0, wl.W, wl.H), gpu.create_view(wl.W, wl.H)
    bilateral = pattern == "bilateral_filter_behind_the_fence"
    for i, (rgba, mm, M) in enumerate(frames):
        gpu.view_update(v0, rgba, mm, timestamp=float(i), bilateral=bilateral)
        gpu.process_frame(s0, v0, rs0, M, wl.intr)
    ref = util.snapshot(gpu, s0, rs0)

    s1 = gpu.create_scene(p)
    rs1, v1 = gpu.create_render_state(s1, wl.W, wl.H), gpu.create_view(wl.W, wl.H)
    store = gpu.create_frame_store(wl.W, wl.H, n_frames)
    rgba_p = gpu.host_alloc((n_frames, wl.H, wl.W, 4), np.uint8)
    mm_p = gpu.host_alloc((n_frames, wl.H, wl.W), np.int16)
    for i, (rgba, mm, M) in enumerate(frames):
        rgba_p[i], mm_p[i] = rgba, mm
    one = gpu.fence_create()
    gpu.set_async(True)
    try:
        for i, (rgba, mm, M) in enumerate(frames):
            if bilateral:
                # fence, then two filtered updates back to back (the first one is overwritten: frame i - 1 again), nothing
                # else reading the view in between
                gpu.fence_record(one)
                gpu.view_update(v1, rgba_p[max(i - 1, 0)], mm_p[max(i - 1, 0)], timestamp=float(i), bilateral=True)
                gpu.view_update(v1, rgba_p[i], mm_p[i], timestamp=float(i), bilateral=True)
                gpu.process_frame(s1, v1, rs1, M, wl.intr)
                continue
            gpu.view_update(v1, rgba_p[i], mm_p[i], timestamp=float(i))
            gpu.process_frame(s1, v1, rs1, M, wl.intr)
            if pattern == "fence_destroyed_while_lent":
                f = gpu.fence_create()
                gpu.fence_record(f)
                if i % 2:  # destroyed right away on odd frames, after the next upload borrowed it on even ones
                    f.close()
                else:
                    if i + 1 < n_frames:
                        gpu.view_update(v1, rgba_p[i], mm_p[i], timestamp=float(i))  # (same frame again: borrows f)
                    f.close()
            elif pattern == "view_read_after_the_fence":
                gpu.fence_record(one)
                gpu.frame_store_put_view(store, i, v1)  # reads the landing buffer behind the fence
            else:
                gpu.fence_record(one)
        gpu.synchronize()
    finally:
        gpu.set_async(False)
    util.assert_same_state(util.snapshot(gpu, s1, rs1), ref, pattern)
    if pattern == "view_read_after_the_fence":
        v2 = gpu.create_view(wl.W, wl.H)
        for i in (0, n_frames // 2, n_frames - 1):
            gpu.view_update_from_store(v2, store, i)
            assert np.array_equal(gpu.download_view_raw_depth(v2), frames[i][1]), f"stored depth image {i}"
            assert np.array_equal(gpu.download_view_rgba(v2), frames[i][0]), f"stored colour image {i}"
    one.close()
    store.close()
    for a in (rgba_p, mm_p):
        gpu.host_free(a)


def test_fence_outlives_its_engine(pkg):
    """A fence closed after the engine it was created on (round-2 ADVICE: dslam_fence_destroy used to follow the fence's engine
    pointer): the engine's destruction takes its outstanding fences along, closing the handle afterwards is a no-op."""
    eng = pkg.open_engine(0)
    f, g = eng.fence_create(), eng.fence_create()
    eng.fence_record(f)
    eng.synchronize()
    assert eng.fence_query(f)
    eng.close()
    f.close()
    g.close()
