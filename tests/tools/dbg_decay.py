import sys, os, time
sys.path.insert(0, "/root/repo"); os.chdir(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import __graft_entry__ as g
pkg = g.load_package()
from dslam_amd.harness import synth
import util
gpu = pkg.open_engine(0)
wl = synth.s_street(320, 240)
p = pkg.SceneParams(num_local_blocks=0x10000, **wl.scene_kwargs)
s, rs, v = util.run_sequence(gpu, pkg, wl, p, 12, decay=(2, 6, True), slide=None)
print("done", gpu.stats(s, rs))
