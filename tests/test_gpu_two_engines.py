"""Two engines of one process, both in async mode, fusing at the same time (round-2 ADVICE: the in-launch look-backs of
round 2 needed the whole grid of a sweep co-resident and assumed the device to themselves).  Since round 3 a tile is
taken by ticket and only waits for workgroups that are already running, and the spins are bounded: kernels of another
stream on the same device can delay a pass but not wedge it.  Both maps must equal the oracle's byte for byte."""
import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu


def test_two_async_engines_fuse_concurrently(pkg, synth, gpu, oracle):
    wl = synth.s_street(320, 240)
    p = pkg.SceneParams(num_local_blocks=0x8000, **wl.scene_kwargs)
    second = pkg.open_engine(0)   # its own streams, scratch, ticket counters
    engines = [gpu, second]
    objs = []
    for eng in engines:
        s = eng.create_scene(p)
        objs.append((eng, s, eng.create_render_state(s, wl.W, wl.H), eng.create_view(wl.W, wl.H), eng.create_render_state(s, wl.W, wl.H)))
    n = 24
    frames = [wl.frame(i) for i in range(n)]
    try:
        for eng in engines:
            eng.set_async(True)
        for i in range(n):      # the calls of the two engines interleave; nothing waits in between
            for k, (eng, s, rs, v, free) in enumerate(objs):
                rgba, mm, M = frames[i] if k == 0 else frames[n - 1 - i]
                eng.view_update(v, rgba, mm, timestamp=float(i))
                eng.process_frame(s, v, rs, M, wl.intr)
                if i % 4 == 3:
                    eng.get_image(s, free, M, wl.intr, pkg.IMAGE_DEPTH, download=False)
                if i % 6 == 5:
                    eng.decay(s, rs, 2, 3, True)
    finally:
        for eng in engines:
            eng.synchronize()
            eng.set_async(False)
    for k, (eng, s, rs, v, free) in enumerate(objs):
        os_ = oracle.create_scene(p)
        ors, ov = oracle.create_render_state(os_, wl.W, wl.H), oracle.create_view(wl.W, wl.H)
        for i in range(n):
            rgba, mm, M = frames[i] if k == 0 else frames[n - 1 - i]
            oracle.view_update(ov, rgba, mm, timestamp=float(i))
            oracle.process_frame(os_, ov, ors, M, wl.intr)
            if i % 6 == 5:
                oracle.decay(os_, ors, 2, 3, True)
        util.assert_same_state(util.snapshot(eng, s, rs), util.snapshot(oracle, os_, ors), f"engine {k}")
        assert eng.stats(s, rs)["no_visible_entries"] > 500
