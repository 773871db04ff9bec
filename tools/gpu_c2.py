"""BASELINE config 2 (random spheres, no acceleration structure) 1280x720 at 64 spp: time, rate, image hash."""
import hashlib, os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import minimaloptix_amd as M
ctx = M.Context(0)
for o in os.environ.get("OPTS", "").split(","):
    if "=" in o:
        k, v = o.split("="); ctx.set_option(k, int(v))
hs = M.HostScene("random_spheres", 1280, 720, iarg=497); seeds = M.launch_seeds(64)
ctx.load(hs); ctx.accum_clear(); st = ctx.render_counted(seeds)
best = 1e9
for rep in range(3):
    ctx.accum_clear(); ctx.kernel_time(reset=True); ctx.render(seeds); ms, n = ctx.kernel_time(); best = min(best, ms)
print(os.environ.get("OPTS", ""), "%s random_spheres 1280x720x64: %.2f ms %.1f Mrays/s %.3g primitive tests/s hash %s variant %d" % (os.environ.get("MOPTIX_DEVICE_LIB", "libmoptix.so"), best, st.rays / best / 1e3,
      st.analyticTests / best * 1e3, hashlib.md5(ctx.accum_read().tobytes()).hexdigest()[:10], ctx.get_option("kernel_variant_used")))
