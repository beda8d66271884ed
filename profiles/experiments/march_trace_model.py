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
def cost(t, use_cache):
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
