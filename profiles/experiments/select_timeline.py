"""Per-tile timeline of the fused ordered selection (k_bits_select) behind GetImage's FindVisibleBlocks (DSLAM_DBG_SELECT dump:
8 s_memtime stamps per tile, shader clock taken as 2.4 GHz, per XCD: only differences inside a tile mean something):
0 kernel entry, 1 ticket taken, 2 words + first scan, 3 list expanded, 4 tested, 5 count published, 6 look-back done, 7 emitted.
    python profiles/experiments/select_timeline.py <dump> [out.json]"""
import sys
import numpy as np
d = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8).astype(np.int64)
names = ["entry", "ticket", "words+scan", "expanded", "tested", "published", "lookback", "emitted"]
busy = d[:, 7] != 0
dur = np.diff(d, axis=1) / 2400.0
print("tiles", len(d), "with work", int(busy.sum()))
print("tile " + " ".join(n.rjust(10) for n in names[1:]))
for i in np.nonzero(busy)[0]:
    print(f"{i:4d} " + " ".join(f"{x:10.2f}" for x in dur[i]))
print("mean " + " ".join(f"{x:10.2f}" for x in dur[busy].mean(axis=0)))
tot = (d[busy, 7] - d[busy, 0]) / 2400.0
print("tile total: mean %.2f max %.2f (tile %d)" % (tot.mean(), tot.max(), np.nonzero(busy)[0][tot.argmax()]))
if len(sys.argv) > 2:
    import json
    json.dump({"kernel": "k_bits_select<SelFrustum<true>>", "source": "DSLAM_DBG_SELECT dump of the 60th GetImage of the bench loop (s_memtime, 2.4 GHz)",
               "tiles": int(len(d)), "tiles_with_candidates": int(busy.sum()), "phases": names[1:],
               "phase_us_mean_over_busy_tiles": {n: round(float(x), 2) for n, x in zip(names[1:], dur[busy].mean(axis=0))},
               "tile_total_us": {"mean": round(float(tot.mean()), 2), "max": round(float(tot.max()), 2)}}, open(sys.argv[2], "w"), indent=1)
