"""First-contact probe for the GPU box: parity + timing sweep, prints everything."""
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from common import M, O, oracle_scene, rmse, hostsim_bvh   # noqa: E402


def main():
    ctx = M.Context(0)
    print("num_cus", ctx.get_option("num_cus"))
    for kind, kw, res, spp in [("spheres", dict(farg=0.5), (160, 90), 4), ("cornell_quads", {}, (64, 64), 4),
                               ("random_spheres", dict(iarg=97), (96, 54), 2), ("file:coffee", {}, (96, 54), 2)]:
        hs = M.HostScene(kind, res[0], res[1], **kw)
        seeds = M.launch_seeds(spp)
        ctx.load(hs); ctx.accum_clear()
        st = ctx.render_counted(seeds)
        g = ctx.accum_read()
        o, ost = oracle_scene(hs).render(seeds)
        print(kind, "rmse", rmse(g / spp, o / spp), "max", float(np.abs(g - o).max()) / spp, "mean", float(g.mean()) / spp, float(o.mean()) / spp)
        print("   gpu", st.as_dict()); print("   orc", ost.as_dict())
    # timing: coffee full HD
    W, H = 1920, 1080
    hs = M.HostScene("file:coffee", W, H)
    for leaf in (4, 2, 8, 1):
        ctx.set_option("leaf_size", leaf)
        ctx.load(hs)
        a = ctx.accel_info()
        print("leaf", leaf, "nodes", a.nNodes, "depth", a.treeDepth, "build ms %.3f" % a.buildMs)
        seeds = M.launch_seeds(8)
        ctx.accum_clear(); st = ctx.render_counted(seeds)
        rays = st.rays
        B = 128 * st.nodeFetches + 48 * st.triTests + 108 * st.closestHits + 72 * st.lightLoads + 24 * W * H
        print("   counters", st.as_dict(), "bytes/ray %.1f" % (B / rays), "lane util %.3f" % (st.activeLaneSteps / max(1, 64 * st.traversalSteps)))
        for thr in (1, 16, 32, 40, 48, 56, 64):
            for bpc in (2, 4):
                ctx.set_option("exit_threshold", thr); ctx.set_option("blocks_per_cu", bpc)
                ctx.accum_clear(); ctx.kernel_time(reset=True)
                ctx.render(seeds)
                ms, n = ctx.kernel_time()
                print("   thr %2d bpc %d: %.2f ms  %.1f Mrays/s  %.2f TB/s(alg)" % (thr, bpc, ms, rays / ms / 1e3, B / ms / 1e9))
    # random spheres 1280x720
    hs = M.HostScene("random_spheres", 1280, 720, iarg=497)
    ctx.load(hs); seeds = M.launch_seeds(8)
    ctx.accum_clear(); st = ctx.render_counted(seeds)
    for bpc in (2, 4, 6):
        ctx.set_option("blocks_per_cu", bpc)
        ctx.accum_clear(); ctx.kernel_time(reset=True); ctx.render(seeds); ms, n = ctx.kernel_time()
        print("random_spheres bpc %d: %.2f ms %.1f Mrays/s analytic tests/s %.3g" % (bpc, ms, st.rays / ms / 1e3, st.analyticTests / ms * 1e3))


if __name__ == "__main__":
    main()
