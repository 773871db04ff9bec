"""Many small renders: reports the slowest one (a scheduler that stalls would show up as an outlier)."""
import os, sys, time, random
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from common import M   # noqa: E402
ctx = M.Context(0)
for o in os.environ.get("OPTS", "").split(","):      # e.g. OPTS=kernel_variant=4,aux_depth=1
    if "=" in o:
        k, v = o.split("="); ctx.set_option(k, int(v))
rng = random.Random(3)
worst = (0, None)
n = int(os.environ.get("N", "300"))
hs_cache = {}
t_all = time.time()
for i in range(n):
    w, h = rng.choice([(8, 8), (16, 8), (64, 36), (101, 37), (200, 112), (320, 180), (640, 360)])
    spp = rng.choice([1, 1, 2, 3, 7])
    key = (w, h)
    if key not in hs_cache:
        hs_cache[key] = M.HostScene("file:coffee", w, h)
    ctx.set_partition(rng.randrange(2), 2) if rng.random() < 0.3 else ctx.set_partition(0, 1)
    ctx.load(hs_cache[key]); ctx.accum_clear()
    t = time.time(); ctx.render(M.launch_seeds(spp, i)); dt = time.time() - t
    if dt > worst[0]: worst = (dt, (w, h, spp, i))
print("renders %d total %.1f s, slowest %.1f ms at %s" % (n, time.time() - t_all, worst[0] * 1e3, worst[1]))
