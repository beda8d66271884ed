#!/usr/bin/env python3
"""Per-tile timeline of k_alloc_sweep from the dump DSLAM_DBG_SWEEP=<file> makes (8 s_memtime stamps per tile, shader
clock, 2.4 ticks per ns on MI355X; the counters of different XCDs are not aligned, so only differences within a tile
mean anything).  usage: python profiles/experiments/sweep_timeline.py dump.bin [out.json]"""
import json
import sys

import numpy as np

NAMES = ["start", "requests_published", "A_done", "B_start", "commit_words_seen", "visible_count_published",
         "ranks_known", "end"]
TICKS_PER_US = 2400.0


def main():
    d = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8).astype(np.float64)
    d = (d - d[:, 0:1]) / TICKS_PER_US
    seg = np.diff(d, axis=1)
    out = {"tiles": int(d.shape[0]), "unit": "us since the tile's own start",
           "mean": {n: round(float(x), 2) for n, x in zip(NAMES, d.mean(0))},
           "max": {n: round(float(x), 2) for n, x in zip(NAMES, d.max(0))},
           "segment_mean": {f"{a}->{b}": round(float(x), 2) for a, b, x in zip(NAMES[:-1], NAMES[1:], seg.mean(0))},
           "last_tile": {n: round(float(x), 2) for n, x in zip(NAMES, d[-1])}}
    txt = json.dumps(out, indent=1)
    print(txt)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")


if __name__ == "__main__":
    main()
