"""Ray counts of the ranks' shares of the coffee frame under the tile split (load balance of the static partition)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from common import M   # noqa: E402
ctx = M.Context(0)
hs = M.HostScene("file:coffee", 1920, 1080)
seeds = M.launch_seeds(int(os.environ.get("SPP", "16")))
for n in (4, 8):
    rays = []
    for r in range(n):
        ctx.set_partition(r, n); ctx.load(hs); ctx.accum_clear(); st = ctx.render_counted(seeds); rays.append(st.rays)
    mean = sum(rays) / n
    print("N=%d rays per rank relative to the mean: %s  (max %+.2f %%)" % (n, " ".join("%.4f" % (x / mean) for x in rays), 100 * (max(rays) / mean - 1)), flush=True)
