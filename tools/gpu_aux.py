# Variant 4 with borrowed slots for deep paths' shadow rays (option aux_depth): results must not change, timing of
# a one-eighth share of the headline frame.
import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from common import M
import numpy as np
ctx = M.Context(0)
ctx.set_option("watchdog_ms", 20000)
for kind, kw, res, spp in (("file:coffee", {}, (64, 36), 2), ("file:coffee", {}, (320, 180), 3), ("coffee_pot_standin", {}, (200, 112), 2),
                           ("million_standin", dict(iarg=50000), (160, 90), 2), ("spheres", {}, (128, 72), 4)):
    hs = M.HostScene(kind, res[0], res[1], **kw); seeds = M.launch_seeds(spp)
    ctx.set_option("kernel_variant", 3); ctx.load(hs); ctx.accum_clear(); ctx.render_counted(seeds); ref = ctx.accum_read()
    ctx.set_option("kernel_variant", 4)
    for ad in (0, 1, 2, 5, 16):
        for counted in (True, False):
            ctx.set_option("aux_depth", ad); ctx.load(hs); ctx.accum_clear()
            try:
                (ctx.render_counted if counted else ctx.render)(seeds); got = ctx.accum_read()
                print(kind, res, "aux_depth", ad, "counted" if counted else "plain", "identical:", np.array_equal(ref, got), "maxdiff", float(np.abs(ref - got).max()), flush=True)
            except Exception as e:
                print(kind, "aux_depth", ad, "FAILED", e, flush=True)
