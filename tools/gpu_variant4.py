import sys, hashlib
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from common import M
import numpy as np
ctx = M.Context(0)
ctx.set_option("watchdog_ms", 20000)
for kind, kw, res, spp in (("file:coffee", {}, (64, 36), 1), ("file:coffee", {}, (320, 180), 3), ("coffee_pot_standin", {}, (200, 112), 2), ("million_standin", dict(iarg=50000), (160, 90), 2)):
    hs = M.HostScene(kind, res[0], res[1], **kw); seeds = M.launch_seeds(spp)
    out = {}
    for v in (3, 4):
        ctx.set_option("kernel_variant", v); ctx.load(hs); ctx.accum_clear()
        try:
            st = ctx.render_counted(seeds); out[v] = (ctx.accum_read(), st.rays)
        except Exception as e:
            print(kind, "variant", v, "FAILED", e); out[v] = (None, 0)
    if out[4][0] is not None:
        print(kind, res, "identical:", np.array_equal(out[3][0], out[4][0]), "rays", out[3][1], out[4][1], "maxdiff", float(np.abs(out[3][0]-out[4][0]).max()))
