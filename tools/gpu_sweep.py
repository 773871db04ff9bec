"""Option sweep on coffee full HD (SPP env, default 64): prints ms per configuration."""
import os, sys, itertools
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from common import M   # noqa: E402
ctx = M.Context(0)
W, H = 1920, 1080
hs = M.HostScene("file:coffee", W, H)
spp = int(os.environ.get("SPP", "64"))
seeds = M.launch_seeds(spp)
# SWEEP="leaf_size=2,4,6,8;swap_lanes=16,24,32" -> cartesian product
axes = []
for part in os.environ.get("SWEEP", "swap_lanes=16,24,32,40;starve_lanes=8,16,24").split(";"):
    k, vs = part.split("=")
    axes.append([(k, int(v)) for v in vs.split(",")])
for combo in itertools.product(*axes):
    for k, v in combo:
        ctx.set_option(k, v)
    ctx.load(hs)
    best = 1e9
    for rep in range(3):
        ctx.accum_clear(); ctx.kernel_time(reset=True); ctx.render(seeds); ms, n = ctx.kernel_time(); best = min(best, ms)
    print(" ".join("%s=%d" % kv for kv in combo), "-> %.2f ms" % best, flush=True)
