"""Quick A/B: coffee full HD at SPP (default 64), default options; prints time + hash."""
import os, sys, hashlib
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from common import M   # noqa: E402
ctx = M.Context(0)
W, H = 1920, 1080
hs = M.HostScene("file:coffee", W, H)
spp = int(os.environ.get("SPP", "64"))
seeds = M.launch_seeds(spp)
for o in os.environ.get("OPTS", "").split(","):
    if "=" in o:
        k, v = o.split("="); ctx.set_option(k, int(v))
if os.environ.get("PART"):                      # PART=r/n: rank r's share of an n-way tile split
    r_, n_ = os.environ["PART"].split("/"); ctx.set_partition(int(r_), int(n_))
ctx.load(hs)
if os.environ.get("COLD"):                      # the FIRST frame of a context: no depth history orders its work (kernel code already loaded)
    r_, n_ = (os.environ.get("PART") or "0/1").split("/")
    ctx.set_partition(0, 3); ctx.load(hs); ctx.render(seeds[:2])          # another partition: loads the kernels, and the switch back forgets the history
    cold = []
    for rep in range(3):
        ctx.set_partition(0, 3); ctx.load(hs); ctx.render(seeds[:1])
        ctx.set_partition(int(r_), int(n_)); ctx.load(hs)
        ctx.accum_clear(); ctx.kernel_time(reset=True); ctx.render(seeds); cold.append(ctx.kernel_time()[0])
    print("cold frames (no history), spp %d: %s ms" % (spp, " ".join("%.2f" % m for m in cold)), flush=True)
for _ in range(int(os.environ.get("WARM", "0"))):   # history for the tile order
    ctx.accum_clear(); ctx.render(seeds)
ctx.accum_clear(); st = ctx.render_counted(seeds)
rays = st.rays
B = 64 * st.nodeFetches + 48 * st.triTests + 108 * st.closestHits + 72 * st.lightLoads + 24 * W * H
best = 1e9
for rep in range(3):
    ctx.accum_clear(); ctx.kernel_time(reset=True); ctx.render(seeds); ms, n = ctx.kernel_time(); best = min(best, ms)
img = ctx.accum_read()
print("spp %d: %.2f ms  %.1f Mrays/s  %.2f TB/s(alg)  hash %s" % (spp, best, rays / best / 1e3, B / best / 1e9, hashlib.md5(img.tobytes()).hexdigest()[:10]), flush=True)
