"""The drain kernel against the packet kernel alone on one scene: image bits, ray / hit / sample counts with drain_below 64 against 0 (MOPTIX_DEBUG=1 adds how
many paths were handed over and who finished how many samples).  SCENE= SIZE=WxH SPP= BASE=<first launch seed index> OPTS=name=value,... REPS="""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import numpy as np
import minimaloptix_amd as M
W, H = (int(x) for x in os.environ.get("SIZE", "320x180").split("x"))
hs = M.HostScene(os.environ.get("SCENE", "coffee_pot_standin"), W, H)
seeds = M.launch_seeds(int(os.environ.get("SPP", "5")), int(os.environ.get("BASE", "864")))
ctx = M.Context(0)
ctx.set_option("kernel_variant", 4)
for o in os.environ.get("OPTS", "").split(","):
    if "=" in o: ctx.set_option(o.split("=")[0], int(o.split("=")[1]))
def run(db):
    ctx.set_option("drain_below", db); ctx.load(hs); ctx.accum_clear(); st = ctx.render_counted(seeds); return ctx.accum_read(), st
a0, s0 = run(0)
for rep in range(int(os.environ.get('REPS', '2'))):
    a, st = run(64)
    d = np.abs(a.astype(np.float64) - a0).max(axis=-1)
    ys, xs = np.nonzero(d > 0)
    print("rep %d: %d pixels differ; rays %d vs %d closest hits %d vs %d shadow %d vs %d bounce %d vs %d samples %d vs %d" % (rep, len(ys), st.rays, s0.rays, st.closestHits, s0.closestHits, st.shadowRays, s0.shadowRays, st.bounceRays, s0.bounceRays, st.samples, s0.samples))
    if len(ys):
        print("   y range %d..%d x range %d..%d; sum(a-a0) = %.3f; mean |d| %.3f; a>a0 in %d, a<a0 in %d" % (ys.min(), ys.max(), xs.min(), xs.max(), float((a - a0).sum()), d[d > 0].mean(), int(((a - a0).sum(axis=-1) > 0).sum()), int(((a - a0).sum(axis=-1) < 0).sum())))
