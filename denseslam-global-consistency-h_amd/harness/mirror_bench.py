"""What the reference's UNCHANGED caller gets: the keyframe loops of DenseSlam::ProcessFrame (DenseSlam.cpp:210-232) driven
through the C++ ITMLib mirror (itmlib/, the classes InfiniTamDriver derives from and calls), by the native program
itmlib/tests/driver_harness -- every frame copied into the driver's own images (CvToItm's role), every call issued as the
reference issues it, the raycast image read through GetData as ItmDepthToCv does.

Loops (120 S-street keyframes at 640x480, timed over the last 50, i.e. with the window full; SURVEY 8d's parameters):
    plain            UpdateView + IntegrateLocalMap
    plain_raycast    ... + one free-camera depth raycast read back per keyframe: bench.py's step through the mirror
    decay            ... + Decay(3, 30, forceAll)                                   (DenseSlam.cpp:227-232)
    decay_window     ... + SlideWindow(50)                                          (DenseSlam.cpp:215-225)
    *_raycast        ... + one free-camera depth raycast read back per keyframe     (SaveRaycastDepth, DenseSlam.cpp:573-603)
    *_swapping       the same with ITMLibSettings::useSwapping
each with the mirror's deferred completion (the default) and with DSLAM_MIRROR_SYNC=1 (every call waits: the round-3 shape).

    python denseslam-global-consistency-h_amd/harness/mirror_bench.py [--keyframes 120] [--time-from 70] [--json-only]
"""
import argparse
import json
import os
import re
import struct
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

HARNESS = os.path.join(ROOT, "denseslam-global-consistency-h_amd", "itmlib", "tests", "driver_harness")
LINE = re.compile(r"driver_harness loop: (\d+) keyframes in ([\d.]+) ms \(([\d.]+) us per keyframe; host time inside the calls: "
                  r"UpdateView ([\d.]+) of which image fill ([\d.]+), fusion \+ window \+ decay ([\d.]+), raycast ([\d.]+); (\d+) bytes in use")


def write_frames(path, pkg, wl, n):
    p = pkg.SceneParams(**wl.scene_kwargs)
    with open(path, "wb") as f:
        f.write(struct.pack("<3i", wl.W, wl.H, n))
        for i in range(n):
            rgba, mm, M = wl.frame(i)
            f.write(rgba.tobytes()); f.write(mm.tobytes()); f.write(pkg.mat_to_abi(M).tobytes())
        f.write(np.asarray(wl.intr, np.float32).tobytes())
        f.write(struct.pack("<4f", p.voxel_size, p.mu, p.frustum_min, p.frustum_max))
        f.write(struct.pack("<4i", p.max_w, p.num_local_blocks or 0x40000, p.num_buckets or 0x100000, p.num_excess or 0x20000))


def run_loop(frames, out, decay, window, raycast, swapping, sync, time_from):
    env = dict(os.environ, DRIVER_HARNESS_DECAY="3,30", DRIVER_HARNESS_TIME_FROM=str(time_from))
    if raycast:
        env["DRIVER_HARNESS_RAYCAST_EACH_FRAME"] = "1"
    if swapping:
        env["DSLAM_USE_SWAPPING"] = "1"
    if sync:
        env["DSLAM_MIRROR_SYNC"] = "1"
    res = subprocess.run([HARNESS, frames, out, "1" if decay else "0", str(window)], capture_output=True, text=True, timeout=600, env=env)
    if res.returncode != 0:
        raise RuntimeError("driver_harness failed:\n" + res.stdout + res.stderr)
    m = LINE.search(res.stdout)
    if not m:
        raise RuntimeError("no timing line in:\n" + res.stdout)
    return {"us_per_keyframe": float(m.group(3)), "keyframes_timed": int(m.group(1)),
            "host_us_in_calls": {"UpdateView": float(m.group(4)), "of_which_image_fill": float(m.group(5)),
                                 "fusion_window_decay": float(m.group(6)), "raycast": float(m.group(7))},
            "us_per_keyframe_without_image_fill": round(float(m.group(3)) - float(m.group(5)), 1),
            "blocks_in_use_end": int(m.group(8)) // 4096}


def measure(keyframes=120, time_from=70, repeats=2, width=640, height=480, loops=None, modes=("deferred", "synchronous")):
    pkg = ge.load_package()
    from dslam_amd.harness import synth
    wl = synth.s_street(width, height)
    out = {"what": "keyframe loops of DenseSlam::ProcessFrame through the C++ ITMLib mirror (itmlib/tests/driver_harness), "
                   f"{keyframes} S-street keyframes {width}x{height}, timed over the last {keyframes - time_from}; us per keyframe, wall clock, "
                   "best of %d runs" % repeats,
           "modes": {"deferred": "the mirror as shipped: calls enqueue, calls that hand data to the host wait",
                     "synchronous": "DSLAM_MIRROR_SYNC=1: every call waits (the round-3 mirror without the per-call counter read-back)"}}
    table = {
        "plain": dict(decay=0, window=-1, raycast=0, swapping=0),
        "plain_raycast": dict(decay=0, window=-1, raycast=1, swapping=0),   # the bench's step: UpdateView + IntegrateLocalMap + GetImage
        "decay": dict(decay=1, window=-1, raycast=0, swapping=0),
        "decay_raycast": dict(decay=1, window=-1, raycast=1, swapping=0),
        "decay_window": dict(decay=1, window=50, raycast=0, swapping=0),
        "decay_window_raycast": dict(decay=1, window=50, raycast=1, swapping=0),
        "decay_window_swapping": dict(decay=1, window=50, raycast=0, swapping=1),
        "decay_window_swapping_raycast": dict(decay=1, window=50, raycast=1, swapping=1),
    }
    with tempfile.TemporaryDirectory() as tmp:
        frames, fout = os.path.join(tmp, "frames.bin"), os.path.join(tmp, "out.bin")
        write_frames(frames, pkg, wl, keyframes)
        for name, kw in table.items():
            if loops and name not in loops:
                continue
            row = {}
            for mode, sync in (("deferred", 0), ("synchronous", 1)):
                if mode not in modes:
                    continue
                runs = [run_loop(frames, fout, sync=sync, time_from=time_from, **kw) for _ in range(repeats)]
                row[mode] = min(runs, key=lambda r: r["us_per_keyframe"])
            out[name] = row
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--keyframes", type=int, default=120)
    ap.add_argument("--time-from", type=int, default=70)
    ap.add_argument("--repeats", type=int, default=2)
    ap.add_argument("--loops", default="")
    a = ap.parse_args()
    print(json.dumps(measure(a.keyframes, a.time_from, a.repeats, loops=[x for x in a.loops.split(",") if x] or None)))


if __name__ == "__main__":
    main()
