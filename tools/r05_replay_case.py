import os, sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from common import M, O, oracle_scene, rmse, hostsim_render
ctx = M.Context(0)
hs = M.HostScene("million_standin", 200, 112, iarg=3000)
seeds = M.launch_seeds(2, 67078)
o, ost = oracle_scene(hs).render(seeds)
ob, obst = oracle_scene(hs, brute_force_tris=True).render(seeds)
print("oracle (its tree): closest hits", ost.closestHits, "rays", ost.rays, "| oracle brute force:", obst.closestHits, obst.rays, "rmse tree-vs-brute %.3g" % rmse(o, ob))
for nf in (128, 64):
    hi, hc = hostsim_render(hs, seeds, node_format=nf)
    print("hostsim (CPU build of the kernel code, device's tree) node format %d: closest hits %d rays %d rmse vs oracle %.3g" % (nf, hc["closestHits"], hc["primaryRays"] + hc["bounceRays"] + hc["shadowRays"], rmse(hi / 2, o / 2)))
for opts in ("kernel_variant=3", "kernel_variant=0", "kernel_variant=4,node_format=64", "kernel_variant=4,node_format=128", "kernel_variant=3,leaf_size=2", "kernel_variant=3,leaf_size=1", "kernel_variant=3,builder=0", "kernel_variant=3,leaf_size=8"):
    c = M.Context(0)
    for kv in opts.split(","):
        k, v = kv.split("="); c.set_option(k, int(v))
    c.load(hs); c.accum_clear(); st = c.render_counted(seeds); g = c.accum_read()
    d = np.abs(g.astype(np.float64) - o.astype(np.float64)).max(axis=-1)
    db = np.abs(g.astype(np.float64) - ob.astype(np.float64)).max(axis=-1)
    print("%-36s closest hits %d rays %d | vs oracle tree: rmse %.3g, %d pixels differ %s | vs oracle brute force: %d pixels differ" % (opts, st.closestHits, st.rays, rmse(g / 2, o / 2), int((d > 1e-5).sum()), np.argwhere(d > 1e-5)[:3].tolist(), int((db > 1e-5).sum())))
    c.close()
