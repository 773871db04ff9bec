"""Trace-kernel time of the triangle scenes (coffee, the dining-room and the million-triangle stand-ins, the coffee scene with the
stand-in pot) under the device library named by MOPTIX_DEVICE_LIB: the A/B harness for changes to the traversal.
   MOPTIX_DEVICE_LIB=libmoptix_n128.so python tools/scene_times.py"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import minimaloptix_amd as M
ctx = M.Context(0); ctx.set_option("watchdog_ms", 60000)
if os.environ.get("NODE_FORMAT"):
    ctx.set_option("node_format", int(os.environ["NODE_FORMAT"]))
for kind, kw, res, spp in [("file:coffee", {}, (1920, 1080), 64), ("dining_standin", dict(iarg=6), (1920, 1080), 16),
                           ("million_standin", dict(iarg=1000000), (1920, 1080), 16), ("coffee_pot_standin", {}, (1920, 1080), 32)]:
    hs = M.HostScene(kind, res[0], res[1], **kw); seeds = M.launch_seeds(spp)
    ctx.load(hs); a = ctx.accel_info()
    ctx.accum_clear(); st = ctx.render_counted(seeds)
    best = 1e9
    for rep in range(3):
        ctx.accum_clear(); ctx.kernel_time(reset=True); ctx.render(seeds); ms, n = ctx.kernel_time(); best = min(best, ms)
    print("%-16s %-20s %7.2f ms %7.1f Mrays/s  per ray %.2f nodes %.2f tris | tris %d nodes %d build %.2f ms | variant %d node format %d" % (
        os.environ.get("MOPTIX_DEVICE_LIB", "libmoptix.so") + " " + os.environ.get("NODE_FORMAT", ""), kind, best, st.rays / best / 1e3, st.nodeFetches / st.rays, st.triTests / st.rays, a.nTriangles, a.nNodes, a.buildMs, ctx.get_option("kernel_variant_used"), ctx.get_option("node_format_used")), flush=True)
