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
ctx.set_option("kernel_variant", 1)
ctx.accum_clear(); st = ctx.render_counted(seeds)
rays = st.rays
B = 128 * st.nodeFetches + 48 * st.triTests + 108 * st.closestHits + 72 * st.lightLoads + 24 * W * H
print("spp", spp, "rays", rays, "rays/sample %.3f" % (rays / st.samples), "bytes/ray %.1f" % (B / rays), "trav util %.3f batch fill %.1f" % (
    st.activeLaneSteps / max(1, 64 * st.traversalSteps), st.shadeBatchLanes / max(1, st.shadeBatches)))
def run(tag):
    ctx.accum_clear(); ctx.kernel_time(reset=True); ctx.render(seeds); ms, n = ctx.kernel_time(); red = ctx.reduce_time()
    print("%-44s %.2f ms (+reduce %.2f)  %.1f Mrays/s  %.2f TB/s(alg)" % (tag, ms, red, rays / ms / 1e3, B / ms / 1e9))
for P, bpc, refill, starve in ((128, 2, 32, 48), (128, 3, 32, 48), (128, 2, 16, 32), (192, 2, 32, 48), (128, 2, 48, 56)):
    ctx.set_option("pool_slots", P); ctx.set_option("blocks_per_cu", bpc); ctx.set_option("refill_lanes", refill); ctx.set_option("starve_lanes", starve)
    run("v1 P%d bpc%d refill%d starve%d" % (P, bpc, refill, starve))
ctx.set_option("kernel_variant", 0)
for bpc, thr in ((2, 16), (2, 24), (2, 8), (4, 16)):
    ctx.set_option("blocks_per_cu", bpc); ctx.set_option("exit_threshold", thr)
    run("v0 bpc%d thr%d" % (bpc, thr))
