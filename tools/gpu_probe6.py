import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from common import M   # noqa: E402
ctx = M.Context(0)
W, H = 1920, 1080
hs = M.HostScene("file:coffee", W, H)
spp = int(os.environ.get("SPP", "32"))
seeds = M.launch_seeds(spp)
ctx.load(hs)
ctx.set_option("kernel_variant", 1)
ctx.accum_clear(); st = ctx.render_counted(seeds)
rays = st.rays
B = 128 * st.nodeFetches + 48 * st.triTests + 108 * st.closestHits + 72 * st.lightLoads + 24 * W * H
img0 = ctx.accum_read()
import hashlib
print(os.environ.get("MOPTIX_DEVICE_LIB", "default"), "hash", hashlib.md5(img0.tobytes()).hexdigest()[:12])
for bpc in (2, 3, 4):
    for lt, refill, starve in ((24, 16, 32), (16, 16, 32)):
        ctx.set_option("blocks_per_cu", bpc); ctx.set_option("leaf_threshold", lt); ctx.set_option("refill_lanes", refill); ctx.set_option("starve_lanes", starve)
        best = 1e9
        for rep in range(2):
            ctx.accum_clear(); ctx.kernel_time(reset=True); ctx.render(seeds); ms, n = ctx.kernel_time(); best = min(best, ms)
        print("  bpc %d lt %d refill %d: %.2f ms  %.1f Mrays/s  %.2f TB/s(alg)" % (bpc, lt, refill, best, rays / best / 1e3, B / best / 1e9), flush=True)
