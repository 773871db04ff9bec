import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from common import M   # noqa: E402
ctx = M.Context(0)
W, H = 1920, 1080
hs = M.HostScene("file:coffee", W, H)
spp = int(os.environ.get("SPP", "32"))
seeds = M.launch_seeds(spp)
ref = None
def run(tag):
    global ref
    ctx.accum_clear(); ctx.kernel_time(reset=True); ctx.render(seeds); ms, n = ctx.kernel_time()
    img = ctx.accum_read()
    if ref is None: ref = img
    print("%-52s %.2f ms  %.1f Mrays/s  %.2f TB/s(alg) same=%s" % (tag, ms, rays / ms / 1e3, B / ms / 1e9, np.array_equal(img, ref)), flush=True)
for leaf in (4, 2, 8):
    ctx.set_option("leaf_size", leaf)
    ctx.load(hs)
    ctx.set_option("kernel_variant", 1)
    ctx.accum_clear(); st = ctx.render_counted(seeds)
    rays = st.rays
    B = 128 * st.nodeFetches + 48 * st.triTests + 108 * st.closestHits + 72 * st.lightLoads + 24 * W * H
    print("leaf", leaf, "bytes/ray %.1f" % (B / rays), "step util %.3f batch fill %.1f" % (st.activeLaneSteps / max(1, 64 * st.traversalSteps), st.shadeBatchLanes / max(1, st.shadeBatches)))
    for lt in (8, 16, 24, 32):
        ctx.set_option("leaf_threshold", lt)
        for refill, starve in ((16, 32), (32, 48)):
            ctx.set_option("refill_lanes", refill); ctx.set_option("starve_lanes", starve)
            ctx.set_option("kernel_variant", 1); run("v1 leaf%d lt%d refill%d starve%d" % (leaf, lt, refill, starve))
        ctx.set_option("kernel_variant", 0); run("v0 leaf%d lt%d" % (leaf, lt))
