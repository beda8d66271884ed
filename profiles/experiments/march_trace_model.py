# CPU-only analysis: per-step trace of every ray of one S-street raycast (oracle hook oracle_raycast_trace_buffer), fed
# into a wave-level cost model (round trips + 4 cycles per instruction of the union path): wave tile shapes, a 2x2x2
# neighbourhood cache for straddling cells, and the lone-ray bound.  Results quoted in DESIGN.md 4b.
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
oracle.get_image(s, rsf, wl.frame(NF - 1)[2], wl.intr, pkg.IMAGE_DEPTH)
oracle.lib.oracle_raycast_trace_buffer(None, C.c_int(0))
t = buf.reshape(wl.H // 8, 8, wl.W // 8, 8, L).transpose(0, 2, 1, 3, 4).reshape(-1, 64, L)  # [tile, lane, iter]
# cost model per wave step (cycles): RT = 1000, instr: plain 180*4, slow adds ~ (5780-2730-2000)
RT = 1000
def cost(t, use_cache, merged=False, merged_extra_instr=0):
    active = t > 0
    miss = (t == 1); probe = (t == 3) | (t == 6) | (t == 7) | (t == 1)
    slow_new = (t == 4) | (t == 6); slow_same = (t == 5) | (t == 7)
    slow = slow_new | slow_same
    any_active = active.any(1)
    any_found = ((t >= 2)).any(1)
    any_probe = probe.any(1)
    any_slow = slow.any(1)
    need_resolve = (slow_new.any(1)) if use_cache else any_slow
    rts = any_probe * 1 + any_found * 1 + any_slow * 1 + need_resolve * 1
    instr = any_active * 400 + any_found * 320 + any_slow * 1050
    if merged:  # SDF shadow addressed by hash-entry index: the bucket-head probe and the nearest-voxel read of a lane share
        # one round trip (the read is speculative on "the head is the block", true for every found block off the excess
        # lists); the union path of a wave step then has ONE round trip where it had probe -> read
        rts = (any_probe | any_found) * 1 + any_slow * 1 + need_resolve * 1
        instr = instr + any_probe * merged_extra_instr
    return (rts * RT + instr) * any_active
c0 = cost(t, False).sum(1); c1 = cost(t, True).sum(1)
iters = (t > 0).any(1).sum(1)
order = np.argsort(-c0)[:10]
print('tiles', len(c0), 'max iters', iters.max(), 'mean iters', iters.mean())
print('longest tiles: iters', iters[order], 'cost now', c0[order], 'with cache', c1[order])
for k in order[:3]:
    tt = t[k]; act = (tt > 0).any(0)
    n = act.sum()
    kinds = [(tt[:, :n] == q).any(0).sum() for q in range(1, 8)]
    print('tile', k, 'iters', n, 'wave-steps with any lane of kind 1..7:', kinds)
    slow_new = ((tt == 4) | (tt == 6)).any(0)[:n]; slow_any = (tt >= 4).any(0)[:n]
    print('   slow wave-steps', slow_any.sum(), 'of which need resolve with cache', slow_new.sum())
print('model kernel time now %.1f us, with cache %.1f us (2.4 GHz)' % (c0.max() / 2400, c1.max() / 2400))
for extra in (0, 80, 160):
    cm = cost(t, False, merged=True, merged_extra_instr=extra * 4).sum(1)
    print('merged probe+read round trip (+%d instructions on a probing step): model kernel time %.1f us (%.0f %% of now); sum over tiles %.3g vs %.3g'
          % (extra, cm.max() / 2400, 100.0 * cm.max() / c0.max(), cm.sum(), c0.sum()))
print('--- wave shapes (model) ---')
img = buf.reshape(wl.H, wl.W, L)
for tw, th in ((8, 8), (16, 4), (32, 2), (64, 1), (4, 16), (2, 32)):
    tt = img.reshape(wl.H // th, th, wl.W // tw, tw, L).transpose(0, 2, 1, 3, 4).reshape(-1, 64, L)
    c = cost(tt, False).sum(1)
    it = (tt > 0).any(1).sum(1)
    print('%2dx%-2d  max tile cost %7d cycles = %.1f us; sum of tile costs %.3g; max iters %d' % (tw, th, c.max(), c.max() / 2400, c.sum(), it.max()))
# lone-ray bound
one = img.reshape(-1, 1, L)
c = cost(one, False).sum(1)
print('lone ray bound: %.1f us' % (c.max() / 2400))

# --- transitions per wave step (for the by-entry experiment: speculative tap loads are warm only where the entry holds a
# block; a lane that steps from a block into empty space sends them to a cold bucket, and the wave waits for them)
prev = np.concatenate([np.zeros_like(t[:, :, :1]), t[:, :, :-1]], axis=2)
found_now = t >= 2; miss_now = t == 1; found_prev = prev >= 2; miss_prev = (prev == 1) | (prev == 0)
probe_now = (t == 1) | (t == 3) | (t == 6) | (t == 7)
f2m = (found_prev & miss_now).any(1); m2f = (miss_prev & found_now).any(1); f2f = (found_prev & found_now & probe_now).any(1)
m2m = (miss_prev & miss_now).any(1)
for k in order[:5]:
    n = int((t[k] > 0).any(0).sum())
    print('tile %d: %d wave-steps; some lane block->empty in %d, empty->block in %d, block->other block in %d, empty->empty in %d'
          % (k, n, f2m[k, :n].sum(), m2f[k, :n].sum(), f2f[k, :n].sum(), m2m[k, :n].sum()))

# --- two-phase march with regrouping (priced only): every tile marches at most S1 steps; rays still going are queued
# with their state and continued by waves that hold only g of them (less union-path cost per step: a ray alone pays one
# round trip on most steps, 64 together pay two on nearly all).  Total = longest phase-1 tile + hand-over + longest group.
print('--- two-phase march, regrouped tails (model; hand-over = one launch boundary + state store/load = 14000 cycles) ---')
HAND = 14000
for S1 in (8, 12, 16, 24, 32):
    c_head = cost(t[:, :, :S1], False).sum(1)
    alive = t[:, :, S1] > 0 if S1 < L else np.zeros(t.shape[:2], bool)
    line = 'S1 = %2d: phase 1 %.1f us, %d rays (%.1f %%) continue;' % (S1, c_head.max() / 2400, alive.sum(), 100.0 * alive.mean())
    for g in (4, 8, 16, 32, 64):
        worst = 0; groups = 0
        for k in np.nonzero(alive.any(1))[0]:
            lanes = np.nonzero(alive[k])[0]
            for a in range(0, len(lanes), g):
                sub = t[k][lanes[a:a + g], S1:][None]
                worst = max(worst, int(cost(sub, False).sum()))
                groups += 1
        line += '  g=%d: %d waves, total %.1f us' % (g, groups, (c_head.max() + HAND + worst) / 2400)
    print(line)

print('--- waves that hold fewer rays (half-empty wave64s; model) ---')
for tw, th in ((8, 8), (8, 4), (4, 8), (4, 4), (8, 2), (2, 2)):
    tt = img.reshape(wl.H // th, th, wl.W // tw, tw, L).transpose(0, 2, 1, 3, 4).reshape(-1, tw * th, L)
    c = cost(tt, False).sum(1)
    print('%dx%d (%d rays per wave, %d waves): longest %.1f us, sum of wave costs / 1024 SIMDs %.1f us' % (tw, th, tw * th, len(c), c.max() / 2400, c.sum() / 2400 / 1024))
