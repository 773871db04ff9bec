"""Launch time against samples per launch: t(spp) = fixed + spp * slope.  `fixed` is what a launch pays once (start-up and the
tail after the last work item), the slope is the steady-state rate.  SCENE=million_standin|dining_standin|coffee"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import minimaloptix_amd as M
kind = os.environ.get("SCENE", "million_standin")
kw = dict(iarg=1000000) if kind == "million_standin" else dict(iarg=6) if kind == "dining_standin" else {}
hs = M.HostScene("file:coffee" if kind == "coffee" else kind, 1920, 1080, **kw)
ctx = M.Context(0); ctx.set_option("watchdog_ms", 60000)
for o in os.environ.get("OPTS", "").split(","):
    if "=" in o:
        k, v = o.split("="); ctx.set_option(k, int(v))
ctx.load(hs)
xs, ys = [], []
for spp in (8, 16, 32, 64):
    seeds = M.launch_seeds(spp)
    ctx.accum_clear(); st = ctx.render_counted(seeds)
    best = 1e9
    for rep in range(3):
        ctx.accum_clear(); ctx.kernel_time(reset=True); ctx.render(seeds); ms, n = ctx.kernel_time(); best = min(best, ms)
    xs.append(spp); ys.append(best)
    print("%s spp %3d: %8.2f ms  %7.1f Mrays/s  counted span %.1f ms tail %.1f ms" % (kind, spp, best, st.rays / best / 1e3,
          ctx.get_option("counted_span_us") / 1e3, ctx.get_option("counted_tail_us") / 1e3), flush=True)
slope, fixed = np.polyfit(xs, ys, 1)
print("%s: t = %.2f ms + spp * %.3f ms  (steady state %.1f Mrays/s)" % (kind, fixed, slope, st.rays / 64 / slope / 1e3))
