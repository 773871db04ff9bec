"""BASELINE configs 4 and 5 (dining-room and glass-knot stand-ins) under option sets: trace-kernel time + image hash.
   OPTSETS="kernel_variant=3;kernel_variant=4,node_format=64;kernel_variant=4,node_format=128" python tools/gpu_c5.py"""
import hashlib, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import minimaloptix_amd as M
scenes = [("million_standin", dict(iarg=1000000), (1920, 1080), 16), ("dining_standin", dict(iarg=6), (1920, 1080), 16)]
for kind, kw, res, spp in scenes:
    hs = M.HostScene(kind, res[0], res[1], **kw); seeds = M.launch_seeds(spp)
    for optset in os.environ.get("OPTSETS", "kernel_variant=3;kernel_variant=4,node_format=64;kernel_variant=4,node_format=128").split(";"):
        ctx = M.Context(0); ctx.set_option("watchdog_ms", 60000)
        for o in optset.split(","):
            if "=" in o:
                k, v = o.split("="); ctx.set_option(k, int(v))
        ctx.load(hs)
        ctx.accum_clear(); st = ctx.render_counted(seeds)
        best = 1e9
        for rep in range(3):
            ctx.accum_clear(); ctx.kernel_time(reset=True); ctx.render(seeds); ms, n = ctx.kernel_time(); best = min(best, ms)
        img = ctx.accum_read()
        print("%-16s %-40s %7.2f ms %7.1f Mrays/s  per ray %.2f nodes %.2f tris  variant %d nodes %d B  hash %s" % (
            kind, optset, best, st.rays / best / 1e3, st.nodeFetches / st.rays, st.triTests / st.rays, ctx.get_option("kernel_variant_used"),
            ctx.get_option("node_format_used") if ctx.get_option("kernel_variant_used") == 4 else 128, hashlib.md5(img.tobytes()).hexdigest()[:10]), flush=True)
        ctx.close()
