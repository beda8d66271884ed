# per-wave dump of k_render (DSLAM_DBG_WAVETIME) condensed: usage render_wave_compare.py file.bin ...
import sys, numpy as np
for f in sys.argv[1:]:
    d = np.fromfile(f, dtype=np.uint64).reshape(-1, 6).astype(np.float64)
    d = d[d[:, 0] > 0]
    us = d[:, 0] / 2400.0
    plain = (d[:, 0] - d[:, 3] - d[:, 4] - d[:, 5]).sum() / max(1.0, (d[:, 1] - d[:, 2]).sum())
    slow = d[:, 3].sum() / max(1.0, d[:, 2].sum())
    k = np.argsort(-us)[:5]
    print(f, 'waves', len(d), 'lifetime us mean %.1f p99 %.1f max %.1f; sum/1024 SIMDs %.1f us; cycles per plain iteration %.0f, per straddling iteration %.0f; setup %.0f tail %.0f'
          % (us.mean(), np.percentile(us, 99), us.max(), us.sum() / 1024, plain, slow, d[:, 4].mean(), d[:, 5].mean()))
    print('   longest waves: us', np.round(us[k], 1), 'iters', d[k, 1], 'straddling', d[k, 2], 'their cycles', d[k, 3])
