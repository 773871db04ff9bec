"""Times coffee 1920x1080 x SPP (default 64) on the GPU under a list of option settings, one line each.
   python tools/sweep_run.py "leaf_size=2" "leaf_size=6" "swap_lanes=24,starve_lanes=12"     (an empty string = the defaults)"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import minimaloptix_amd as M
hs = M.HostScene("file:coffee", 1920, 1080)
seeds = M.launch_seeds(int(os.environ.get("SPP", "64")))
for spec in (sys.argv[1:] or [""]):
    ctx = M.Context(0)
    ctx.set_option("kernel_variant", 4); ctx.set_option("watchdog_ms", 20000)
    for kv in filter(None, spec.split(",")):
        k, v = kv.split("="); ctx.set_option(k, int(v))
    ctx.load(hs)
    ctx.accum_clear(); st = ctx.render_counted(seeds)
    best = 1e9
    for rep in range(3):
        ctx.accum_clear(); ctx.kernel_time(reset=True); ctx.render(seeds); ms, n = ctx.kernel_time(); best = min(best, ms)
    print("%-40s %.2f ms  %.1f Mrays/s  per ray %.2f nodes %.2f tris" % (spec or "(defaults)", best, st.rays / best / 1e3, st.nodeFetches / st.rays, st.triTests / st.rays), flush=True)
    del ctx
