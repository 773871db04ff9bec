import os, sys, hashlib
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from common import M, oracle_scene, rmse   # noqa: E402
ctx = M.Context(0)
# parity first (small)
hs = M.HostScene("file:coffee", 200, 112); seeds = M.launch_seeds(3)
for var in (0, 1, 2, 3):
    ctx.set_option("kernel_variant", var); ctx.load(hs); ctx.accum_clear(); st = ctx.render_counted(seeds); g = ctx.accum_read()
    if var == 0: o, ost = oracle_scene(hs).render(seeds)
    print("variant", var, "rmse vs oracle", rmse(g / 3, o / 3), "rays", st.rays, ost.rays, flush=True)
W, H = 1920, 1080
hs = M.HostScene("file:coffee", W, H)
spp = int(os.environ.get("SPP", "32"))
seeds = M.launch_seeds(spp)
ctx.load(hs)
ctx.set_option("kernel_variant", 2)
ctx.accum_clear(); st = ctx.render_counted(seeds)
rays = st.rays
B = 128 * st.nodeFetches + 48 * st.triTests + 108 * st.closestHits + 72 * st.lightLoads + 24 * W * H
ref = None
def run(tag):
    global ref
    best = 1e9
    for rep in range(2):
        ctx.accum_clear(); ctx.kernel_time(reset=True); ctx.render(seeds); ms, n = ctx.kernel_time(); best = min(best, ms)
    img = ctx.accum_read()
    if ref is None: ref = img
    print("%-40s %.2f ms  %.1f Mrays/s  %.2f TB/s(alg) same=%s" % (tag, best, rays / best / 1e3, B / best / 1e9, np.array_equal(img, ref)), flush=True)
ctx.set_option("kernel_variant", 2); ctx.set_option("swap_lanes", 16); ctx.set_option("starve_lanes", 32); run("v2 swap16 starve32")
ctx.set_option("kernel_variant", 3)
for bpc in (2,):
    ctx.set_option("blocks_per_cu", bpc)
    for swap in (8, 16, 24):
        for starve in (8, 16, 32):
            ctx.set_option("swap_lanes", swap); ctx.set_option("starve_lanes", starve)
            run("v3 bpc%d swap%d starve%d" % (bpc, swap, starve))
ctx.set_option("swap_lanes", 16); ctx.set_option("starve_lanes", 16)
ctx.accum_clear(); st = ctx.render_counted(seeds)
