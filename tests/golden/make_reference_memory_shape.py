#!/usr/bin/env python3
"""Extracts the shape numbers of the reference's four memory logs (/root/reference/memory*.txt: the only recorded
behaviour of its decay / sliding-window path) into tests/golden/reference_memory_shape.json -- data, computed by the same
reduction (harness/memory_curves.py::shape_metrics) that the tests apply to this engine's curves.  The x axis of the
reference logs is the GUI frame number (every frame logs, keyframe or not); only ratios are kept.
Run in the build container:  python tests/golden/make_reference_memory_shape.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

REF = "/root/reference"


def first_run(path):
    rows = [l.split() for l in open(path) if l.strip()]
    fr = np.array([int(r[0]) for r in rows])
    v = np.array([float(r[1]) for r in rows])
    back = np.nonzero(np.diff(fr) < 0)[0]  # memory.txt has a second, shorter run appended (ios::app)
    end = back[0] + 1 if len(back) else len(rows)
    return v[:end]


def main():
    ge.load_package()
    from dslam_amd.harness import memory_curves as mc
    curves = {name: first_run(os.path.join(REF, name + ".txt")) for name in mc.MODES}
    # the un-windowed logs end where the run ended: the pool is full there (10.2151 and 10.2001 of 10.24)
    exhausted = {"memory": len(curves["memory"]), "memory_decay": len(curves["memory_decay"]),
                 "memory_slide_window": None, "memory_decay_slide_window": None}
    shape = mc.shape_metrics(curves, exhausted)
    out = {"source": "memory.txt, memory_decay.txt, memory_slide_window.txt, memory_decay_slide_window.txt of the reference "
                     "(labels: scripts/memoryDraw.py:12-13); first run of each file",
           "lines": {k: int(len(v)) for k, v in curves.items()},
           "last_value": {k: float(v[-1]) for k, v in curves.items()},
           "shape": shape}
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "reference_memory_shape.json"), "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
