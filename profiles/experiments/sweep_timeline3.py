"""Per-tile timeline of the round-3 allocation sweep (DSLAM_DBG_SWEEP dump: 8 s_memtime stamps per tile; the counter runs
at the shader clock, ~2.4 GHz, and is per XCD: only differences inside a tile mean something; printed in units of 100 ticks
= ~42 ns): 0 kernel entry, 1 ticket taken, 2 words + late-mark check, 3 counts published, 4 look-back done, 5 requests
committed, 6 entries from other tiles known, 7 list written.
    python profiles/experiments/sweep_timeline3.py <dump> [out.json]"""
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

if len(sys.argv) > 2:
    import json
    dur = np.diff(d, axis=1) / 2400.0   # us at 2.4 GHz
    json.dump({"kernel": "k_alloc_sweep", "source": "DSLAM_DBG_SWEEP dump of the 60th pass of the bench loop (s_memtime at the shader clock, taken as 2.4 GHz)",
               "tiles": int(len(d)), "phases": names[1:],
               "phase_us_mean_over_tiles": {n: round(float(x), 2) for n, x in zip(names[1:], dur.mean(axis=0))},
               "tile_total_us": {"mean": round(float(dur.sum(axis=1).mean()), 2), "max": round(float(dur.sum(axis=1).max()), 2),
                                 "slowest_tile": int(dur.sum(axis=1).argmax())},
               "phase_us_of_the_slowest_tile": {n: round(float(x), 2) for n, x in zip(names[1:], dur[dur.sum(axis=1).argmax()])}},
              open(sys.argv[2], "w"), indent=1)
