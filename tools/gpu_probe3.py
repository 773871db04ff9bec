import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from common import M   # noqa: E402
ctx = M.Context(0)
W, H = 1920, 1080
hs = M.HostScene("file:coffee", W, H)
seeds = M.launch_seeds(8)
ctx.load(hs)
for P, refill, starve in ((128, 16, 32), (128, 4, 8), (256, 16, 32)):
    ctx.set_option("pool_slots", P); ctx.set_option("refill_lanes", refill); ctx.set_option("starve_lanes", starve)
    ctx.accum_clear(); st = ctx.render_counted(seeds)
    print(P, refill, starve, st.as_dict())
