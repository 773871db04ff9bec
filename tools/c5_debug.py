import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import minimaloptix_amd as M
ctx = M.Context(0)
for o in os.environ.get("OPTS", "kernel_variant=4,node_format=64").split(","):
    k, v = o.split("="); ctx.set_option(k, int(v))
hs = M.HostScene("million_standin", 1920, 1080, iarg=1000000); seeds = M.launch_seeds(int(os.environ.get("SPP", "16")))
ctx.load(hs); ctx.accum_clear(); st = ctx.render_counted(seeds)
print("rays", st.rays, "samples", st.samples, "rays/sample %.2f" % (st.rays / st.samples), "span us", ctx.get_option("counted_span_us"), "tail us", ctx.get_option("counted_tail_us"))
