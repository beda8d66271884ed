"""Timeline of k_mark's pixel workgroups (DSLAM_DBG_MARK dump: per workgroup 4 stamps of its first wave, s_memtime taken as 2.4 GHz;
the counters of different CUs are not aligned, so only differences inside a workgroup mean something): entry -> depth pixel arrived ->
walk over -> exit.   python profiles/experiments/mark_timeline.py <dump> [pixel_workgroups = 1200]"""
import sys
import numpy as np
d = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4)
n_pix = int(sys.argv[2]) if len(sys.argv) > 2 else 1200
re, pix = d[:len(d) - n_pix].astype(np.int64), d[len(d) - n_pix:]
steps = (pix[:, 3] >> np.uint64(56)).astype(int)
t = pix.astype(np.int64)
t3 = (pix[:, 3] & np.uint64((1 << 56) - 1)).astype(np.int64)
a, b, c = (t[:, 2] - t[:, 0]) / 2400.0, (t3 - t[:, 2]) / 2400.0, (t[:, 1] - t3) / 2400.0
print("re-test workgroups %d: life mean %.2f max %.2f us" % (len(re), ((re[:, 1] - re[:, 0]) / 2400.0).mean(), ((re[:, 1] - re[:, 0]) / 2400.0).max()))
print("pixel workgroups %d (first wave): depth %.2f  walk %.2f  tail %.2f us (means); total mean %.2f p99 %.2f max %.2f" %
      (n_pix, a.mean(), b.mean(), c.mean(), (a + b + c).mean(), np.percentile(a + b + c, 99), (a + b + c).max()))
for st in sorted(set(steps)):
    m = steps == st
    print("  wave_steps %d: n %4d  depth %.2f walk %.2f tail %.2f" % (st, m.sum(), a[m].mean(), b[m].mean(), c[m].mean()))
