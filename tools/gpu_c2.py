"""BASELINE config 1 (random spheres, no acceleration structure) and the Cornell box of quads: time + hash."""
import os, sys, hashlib
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from common import M   # noqa: E402
ctx = M.Context(0)
for o in os.environ.get("OPTS", "").split(","):
    if "=" in o:
        k, v = o.split("="); ctx.set_option(k, int(v))
for kind, kw, res, spp in (("random_spheres", dict(iarg=497), (1280, 720), 64), ("cornell_quads", {}, (256, 256), 64), ("spheres", {}, (1280, 720), 64)):
    hs = M.HostScene(kind, res[0], res[1], **kw); seeds = M.launch_seeds(spp)
    ctx.load(hs); ctx.accum_clear(); st = ctx.render_counted(seeds)
    best = 1e9
    for rep in range(3):
        ctx.accum_clear(); ctx.kernel_time(reset=True); ctx.render(seeds); ms, n = ctx.kernel_time(); best = min(best, ms)
    print("%-16s %dx%d spp %d: %.2f ms  %.1f Mrays/s  %.3g analytic tests/s  hash %s" % (kind, res[0], res[1], spp, best, st.rays / best / 1e3, st.analyticTests / best * 1e3, hashlib.md5(ctx.accum_read().tobytes()).hexdigest()[:10]), flush=True)
