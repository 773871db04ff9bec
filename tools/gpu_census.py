"""Counting build of one scene (MOPTIX_DEBUG=1 prints the phase clocks, slot-row counts and the lane census of the passes):
   MOPTIX_DEBUG=1 SCENE=million_standin IARG=1000000 SPP=16 OPTS=kernel_variant=4,node_format=64 python tools/gpu_census.py"""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import minimaloptix_amd as M
ctx = M.Context(0)
for o in os.environ.get("OPTS", "").split(","):
    if "=" in o:
        k, v = o.split("="); ctx.set_option(k, int(v))
hs = M.HostScene(os.environ.get("SCENE", "file:coffee"), int(os.environ.get("WIDTH", "1920")), int(os.environ.get("HEIGHT", "1080")), iarg=int(os.environ.get("IARG", "0")))
seeds = M.launch_seeds(int(os.environ.get("SPP", "16")))
ctx.load(hs); ctx.accum_clear(); st = ctx.render_counted(seeds)
print("rays", st.rays, "samples", st.samples, "rays/sample %.2f" % (st.rays / st.samples), "variant", ctx.get_option("kernel_variant_used"),
      "span us", ctx.get_option("counted_span_us"), "tail us", ctx.get_option("counted_tail_us"))
