"""Per-tile timeline of the round-3 allocation sweep (DSLAM_DBG_SWEEP dump: 8 s_memtime stamps per tile, 100 MHz clock):
0 kernel entry, 1 ticket taken, 2 words + late-mark check, 3 counts published, 4 look-back done, 5 requests committed,
6 entries from other tiles known, 7 list written.  python profiles/experiments/sweep_timeline3.py <dump>"""
import sys
import numpy as np
d = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8).astype(np.int64)
t0 = d[:, 0].min()
us = (d - t0) / 100.0
names = ["entry", "ticket", "words+late", "published", "lookback", "requests", "newx", "list"]
print("tile " + " ".join(n.rjust(10) for n in names))
for i, row in enumerate(us):
    print(f"{i:4d} " + " ".join(f"{x:10.2f}" for x in row))
print("max  " + " ".join(f"{x:10.2f}" for x in us.max(axis=0)))
print("mean step durations:", " ".join(f"{n}:{x:.2f}" for n, x in zip(names[1:], np.diff(us, axis=1).mean(axis=0))))
