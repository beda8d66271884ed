"""Wall time of every call of the frame loop per frame index (which call of which frames is slow?).
    python profiles/experiments/per_call_times.py <workload> <frames> [sync|async] [timer]
sync: every call returns when its work is done (pageable host frames in); async: device-resident frames, calls only enqueue -- the
columns are then the host's enqueue times, and the line at the end is the loop's wall time per frame; timer: with the bench's
per-launch kernel timer on."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import __graft_entry__ as ge
pkg = ge.load_package()
from dslam_amd.harness import synth
import torch
name, n = sys.argv[1], int(sys.argv[2])
mode = sys.argv[3] if len(sys.argv) > 3 else "sync"
wl = getattr(synth, name)(640, 480)
frames = [wl.frame(i) for i in range(n)]
eng = pkg.open_engine(0)
scene = eng.create_scene(pkg.SceneParams(num_local_blocks=0x40000, **wl.scene_kwargs))
view = eng.create_view(wl.W, wl.H)
rs, rs_free = eng.create_render_state(scene, wl.W, wl.H), eng.create_render_state(scene, wl.W, wl.H)
dev = [(torch.from_numpy(f[0]).cuda(), torch.from_numpy(f[1]).cuda()) for f in frames] if mode == "async" else None
torch.cuda.synchronize()
eng.set_async(mode == "async")
if "timer" in sys.argv:
    eng.kernel_timer_enable(True)
t = np.zeros((n, 3))
t0 = time.perf_counter()
for i in range(n):
    rgba, mm, M = frames[i]
    a = time.perf_counter()
    if mode == "async":
        eng.view_update_device(view, dev[i][0].data_ptr(), dev[i][1].data_ptr(), timestamp=float(i))
    else:
        eng.view_update(view, rgba, mm, timestamp=float(i))
    b = time.perf_counter(); eng.process_frame(scene, view, rs, M, wl.intr)
    c = time.perf_counter(); eng.get_image(scene, rs_free, M, wl.intr, pkg.IMAGE_DEPTH, download=False)
    d = time.perf_counter()
    t[i] = (b - a, c - b, d - c)
eng.synchronize()
total = (time.perf_counter() - t0) * 1e6 / n
eng.set_async(False)
t *= 1e6
for lo in range(0, n, 20):
    print("frames %3d-%3d  view_update %6.1f  process_frame %6.1f  get_image %6.1f us" % ((lo, min(n, lo + 20) - 1) + tuple(t[lo:lo + 20].mean(axis=0))))
print("loop: %.1f us per frame (%s)" % (total, mode))
