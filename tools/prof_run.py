"""Tiny driver for rocprofv3: one warm-up + N timed renders of a scene (no torch import)."""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import minimaloptix_amd as M   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--scene", default="file:coffee")
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--spp", type=int, default=8)
ap.add_argument("--iarg", type=int, default=0)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--opt", action="append", default=[], help="name=value")
a = ap.parse_args()
ctx = M.Context(0)
for o in a.opt:
    k, v = o.split("=")
    ctx.set_option(k, int(v))
hs = M.HostScene(a.scene, a.width, a.height, iarg=a.iarg)
ctx.load(hs)
seeds = M.launch_seeds(a.spp)
ctx.accum_clear(); ctx.render(seeds[:1])
ctx.kernel_time(reset=True)
for r in range(a.reps):
    ctx.accum_clear(); ctx.render(seeds)
ms, n = ctx.kernel_time()
print("scene %s %dx%d spp %d: %.3f ms per render (%d launches)" % (a.scene, a.width, a.height, a.spp, ms / max(1, n), n))
