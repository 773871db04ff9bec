"""Probe for the pool kernel (variant 1): parity vs oracle + option sweep on coffee full HD."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from common import M, O, oracle_scene, rmse   # noqa: E402

ctx = M.Context(0)
for kind, kw, res, spp in [("file:coffee", {}, (96, 54), 2), ("file:coffee", {}, (200, 112), 3)]:
    hs = M.HostScene(kind, res[0], res[1], **kw)
    seeds = M.launch_seeds(spp)
    for var in (0, 1):
        ctx.set_option("kernel_variant", var)
        ctx.load(hs); ctx.accum_clear()
        st = ctx.render_counted(seeds)
        g = ctx.accum_read()
        o, ost = oracle_scene(hs).render(seeds)
        print(kind, res, "variant", var, "rmse", rmse(g / spp, o / spp), "rays", st.rays, ost.rays, "hits", st.closestHits, ost.closestHits)
W, H = 1920, 1080
hs = M.HostScene("file:coffee", W, H)
seeds = M.launch_seeds(8)
ctx.set_option("kernel_variant", 1)
ctx.load(hs)
ref = None
for P in (128, 192, 256):
    ctx.set_option("pool_slots", P)
    for bpc in (2, 3):
        if P == 256 and bpc == 3: continue
        ctx.set_option("blocks_per_cu", bpc)
        for refill, starve in ((16, 32), (8, 16), (32, 48), (4, 8), (24, 40)):
            ctx.set_option("refill_lanes", refill); ctx.set_option("starve_lanes", starve)
            ctx.accum_clear(); st = ctx.render_counted(seeds)
            rays = st.rays
            B = 128 * st.nodeFetches + 48 * st.triTests + 108 * st.closestHits + 72 * st.lightLoads + 24 * W * H
            ctx.accum_clear(); ctx.kernel_time(reset=True); ctx.render(seeds); ms, n = ctx.kernel_time()
            img = ctx.accum_read()
            if ref is None: ref = img
            print("P %3d bpc %d refill %2d starve %2d: %.2f ms %.1f Mrays/s %.2f TB/s(alg) | trav util %.3f batch fill %.1f | same=%s" % (
                P, bpc, refill, starve, ms, rays / ms / 1e3, B / ms / 1e9, st.activeLaneSteps / max(1, 64 * st.traversalSteps),
                st.shadeBatchLanes / max(1, st.shadeBatches), np.array_equal(img, ref)))
ctx.set_option("kernel_variant", 0); ctx.set_option("blocks_per_cu", 2); ctx.set_option("exit_threshold", 16)
ctx.accum_clear(); ctx.kernel_time(reset=True); ctx.render(seeds); ms, n = ctx.kernel_time()
print("variant 0: %.2f ms  same=%s" % (ms, np.array_equal(ctx.accum_read(), ref)))
