import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import minimaloptix_amd as M
ctx = M.Context(0)
ctx.set_option("kernel_variant", 4); ctx.set_option("blocks_per_cu", int(os.environ.get("BPC", "3"))); ctx.set_option("watchdog_ms", 20000)
hs = M.HostScene("file:coffee", 1920, 1080)
seeds = M.launch_seeds(int(os.environ.get("SPP", "64")))
ctx.load(hs)
ctx.accum_clear(); st = ctx.render_counted(seeds)
best = 1e9
for rep in range(3):
    ctx.accum_clear(); ctx.kernel_time(reset=True); ctx.render(seeds); ms, n = ctx.kernel_time(); best = min(best, ms)
print("%s BPC=%s: %.2f ms, rays %.3g (%.1f Mrays/s), node fetches %.3g (%.2f G/s), tri tests %.3g, per ray %.2f nodes %.2f tris" % (
    os.environ.get("MOPTIX_DEVICE_LIB"), os.environ.get("BPC"), best, st.rays, st.rays / best / 1e3, st.nodeFetches, st.nodeFetches / best / 1e6, st.triTests,
    st.nodeFetches / st.rays, st.triTests / st.rays), flush=True)
