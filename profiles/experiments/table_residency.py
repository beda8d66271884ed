# sensitivity of k_render to the hash table's cache residency: same frames, smaller bucket array (NOT a valid config)
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from dslam_amd.harness import synth
buckets = int(sys.argv[1], 0)
wl = synth.s_street(640, 480)
n = 70
frames = [wl.frame(i) for i in range(n)]
eng = pkg.open_engine(0)
p = pkg.SceneParams(num_buckets=buckets, num_excess=0x20000, **wl.scene_kwargs)
s = eng.create_scene(p); rs = eng.create_render_state(s, wl.W, wl.H); rsf = eng.create_render_state(s, wl.W, wl.H)
v = eng.create_view(wl.W, wl.H)
eng.set_async(True)
t0 = None
for i, (rgba, mm, M) in enumerate(frames):
    if i == 10:
        eng.synchronize(); t0 = time.perf_counter()
    eng.view_update(v, rgba, mm, timestamp=float(i))
    eng.process_frame(s, v, rs, M, wl.intr)
    eng.get_image(s, rsf, M, wl.intr, pkg.IMAGE_DEPTH, download=False)
eng.synchronize()
print('buckets', hex(buckets), 'us/frame (host-upload incl.)', (time.perf_counter() - t0) / (n - 10) * 1e6, eng.stats(s, rs))
