import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from common import M   # noqa: E402
ctx = M.Context(0)
W, H = 1920, 1080
hs = M.HostScene("file:coffee", W, H)
spp = int(os.environ.get("SPP", "64"))
seeds = M.launch_seeds(spp)
ctx.load(hs)
ctx.accum_clear(); st = ctx.render_counted(seeds)
rays = st.rays
B = 128 * st.nodeFetches + 48 * st.triTests + 108 * st.closestHits + 72 * st.lightLoads + 24 * W * H
def run(tag):
    best = 1e9
    for rep in range(2):
        ctx.accum_clear(); ctx.kernel_time(reset=True); ctx.render(seeds); ms, n = ctx.kernel_time(); best = min(best, ms)
    print("%-30s %.2f ms  %.1f Mrays/s  %.2f TB/s(alg)" % (tag, best, rays / best / 1e3, B / best / 1e9), flush=True)
for swap in (12, 16, 24, 32):
    for starve in (8, 16, 24, 32):
        ctx.set_option("swap_lanes", swap); ctx.set_option("starve_lanes", starve)
        run("swap%d starve%d" % (swap, starve))
