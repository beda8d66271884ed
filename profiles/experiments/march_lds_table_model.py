# Round 4, VERDICT item 2 -- PRICED BEFORE BUILDING: a complete per-cell block table in LDS for k_render.
# Every variant of DESIGN 4b kept a global hash probe (one 16-byte read of the 19 MB table, Infinity-Cache latency, ~1000 cycles)
# on a march step's critical path.  Idea: the range-image pass already visits every (block, 8x8-pixel cell) pair; let it append
# (pos, ptr) to a per-cell list, and let each k_render workgroup (= one cell) build an open-addressed LDS table from its list
# before it marches: a step's lookup -- hit AND miss -- is then an LDS probe, only the tap load goes to memory.
# This script feeds the same per-step trace as march_trace_model.py (oracle hook) into the same calibrated cost model with
#   probe round trip (1000 cycles) -> LDS probe (LDS_PROBE cycles; the straddling cell's resolve likewise)
#   + table build per workgroup (one global round trip for the list + LDS inserts)
# and prints the modelled k_render time against the current one.  CPU only.
import sys, ctypes as C
import os; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np
import __graft_entry__ as ge
pkg = ge.load_package()
from dslam_amd.harness import synth
import util
o_ = ge.load_oracle(); o_.build(); oracle = o_.open_oracle(pkg.CApi, threads=8)
wl = synth.s_street(640, 480)
p = pkg.SceneParams(**wl.scene_kwargs)
NF = int(sys.argv[1]) if len(sys.argv) > 1 else 12
s, rs, v = util.run_sequence(oracle, pkg, wl, p, NF)
L = 128
buf = np.zeros((wl.H * wl.W, L), np.uint8)
oracle.lib.oracle_raycast_trace_buffer(buf.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int(L))
oracle.set_threads(1)
rsf = oracle.create_render_state(s, wl.W, wl.H)
M_last = wl.frame(NF - 1)[2]
oracle.get_image(s, rsf, M_last, wl.intr, pkg.IMAGE_DEPTH)
oracle.lib.oracle_raycast_trace_buffer(None, C.c_int(0))
t = buf.reshape(wl.H // 8, 8, wl.W // 8, 8, L).transpose(0, 2, 1, 3, 4).reshape(-1, 64, L)  # [tile, lane, iter]
RT = 1000

def cost(t, lds=None, build=0):
    active = t > 0
    probe = (t == 3) | (t == 6) | (t == 7) | (t == 1)
    slow_new = (t == 4) | (t == 6); slow_same = (t == 5) | (t == 7)
    slow = slow_new | slow_same
    any_active = active.any(1); any_found = (t >= 2).any(1); any_probe = probe.any(1); any_slow = slow.any(1)
    instr = any_active * 400 + any_found * 320 + any_slow * 1050
    if lds is None:
        rts = any_probe * 1 + any_found * 1 + any_slow * 2       # probe | taps | resolve + second tap fetch
        c = (rts * RT + instr) * any_active
    else:
        probe_cyc, extra_instr = lds
        rts = any_found * 1 + any_slow * 1                        # taps | second tap fetch
        c = (rts * RT + instr + any_probe * (probe_cyc + extra_instr * 4) + any_slow * (2 * probe_cyc + 2 * extra_instr * 4)) * any_active
    return c.sum(1) + build

# how long the per-cell lists are: visible blocks whose projected box touches the cell (the pairs the range-image pass visits)
vis = oracle.download_visible_ids(rsf)
table = oracle.download_hash_table(s)
pos = table["pos"][vis].astype(np.float64) * 8 * p.voxel_size
corners = np.array([[x, y, z] for x in (0, 1) for y in (0, 1) for z in (0, 1)], np.float64) * 8 * p.voxel_size
M = np.asarray(M_last, np.float64)
fx, fy, cx, cy = wl.intr
cnt = np.zeros((wl.H // 8, wl.W // 8), np.int64)
behind = 0
for b in pos:
    pc = (M[:3, :3] @ (b[None, :] + corners).T).T + M[:3, 3]
    if (pc[:, 2] < 1e-3).any():
        behind += 1
        cnt += 1           # conservative: such a block goes on every cell's list (or the cell falls back)
        continue
    u = fx * pc[:, 0] / pc[:, 2] + cx; w = fy * pc[:, 1] / pc[:, 2] + cy
    x0, x1 = int(np.floor(max(0, u.min()) / 8)), int(np.floor(min(wl.W - 1, u.max()) / 8))
    y0, y1 = int(np.floor(max(0, w.min()) / 8)), int(np.floor(min(wl.H - 1, w.max()) / 8))
    if x1 >= x0 and y1 >= y0:
        cnt[y0:y1 + 1, x0:x1 + 1] += 1
print('visible blocks %d (with a corner behind the camera: %d); (block, cell) pairs %d; per-cell list length: mean %.1f, p50 %d, p90 %d, p99 %d, max %d'
      % (len(vis), behind, cnt.sum(), cnt.mean(), np.percentile(cnt, 50), np.percentile(cnt, 90), np.percentile(cnt, 99), cnt.max()))
n_list = cnt.reshape(-1)

c0 = cost(t)
print('model now: longest tile %.1f us (sum over tiles / 1024 SIMDs %.1f us)' % (c0.max() / 2400, c0.sum() / 2400 / 1024))
for probe_cyc in (150, 300, 500):
    for extra in (20, 60):
        # build: one global round trip for the list + ceil(n / 64) rounds of (load 8 B, hash, LDS insert with linear probing ~ 200 cycles)
        build = RT + np.ceil(n_list / 64.0) * 250 + 300
        c1 = cost(t, lds=(probe_cyc, extra), build=build)
        k = int(np.argmax(c1))
        print('LDS table (probe %3d cycles, +%2d instructions per probing step): longest tile %.1f us = %.0f %% of now; sum %.0f %%; its list %d blocks, build %.1f us'
              % (probe_cyc, extra, c1.max() / 2400, 100.0 * c1.max() / c0.max(), 100.0 * c1.sum() / c0.sum(), n_list[k], build[k] / 2400))
# how much of the longest tiles' time is the probe round trip
order = np.argsort(-c0)[:5]
probe = (t == 3) | (t == 6) | (t == 7) | (t == 1)
for k in order:
    n = int((t[k] > 0).any(0).sum())
    print('tile %d: %d wave-steps, some lane probes in %d, straddles in %d; list length %d' % (k, n, probe[k].any(0).sum(), (t[k] >= 4).any(0).sum(), n_list[k]))
