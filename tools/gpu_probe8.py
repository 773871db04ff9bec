import os, sys, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from common import M, oracle_scene, rmse   # noqa: E402
ctx = M.Context(0)
hs = M.HostScene("file:coffee", 200, 112); seeds = M.launch_seeds(3)
o, ost = oracle_scene(hs).render(seeds)
for var in (2, 3):
    ctx.set_option("kernel_variant", var); ctx.load(hs); ctx.accum_clear()
    t0 = time.time(); st = ctx.render_counted(seeds); dt = time.time() - t0; g = ctx.accum_read()
    print("variant", var, "rmse vs oracle", rmse(g / 3, o / 3), "rays", st.rays, ost.rays, "wall %.3fs" % dt, flush=True)
hs = M.HostScene("file:coffee", 640, 360); seeds = M.launch_seeds(4)
ref = None
for var in (2, 3):
    ctx.set_option("kernel_variant", var); ctx.load(hs); ctx.accum_clear(); ctx.kernel_time(reset=True)
    t0 = time.time(); ctx.render(seeds); dt = time.time() - t0; ms, n = ctx.kernel_time(); g = ctx.accum_read()
    if ref is None: ref = g
    print("640x360x4 variant", var, "%.2f ms wall %.3fs same=%s" % (ms, dt, np.array_equal(g, ref)), flush=True)
