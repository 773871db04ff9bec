"""C4 / C5 stand-ins at 1920x1080x16: kernel variant and slot options (OPTS sets, one per line of CASES)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from common import M   # noqa: E402
ctx = M.Context(0)
scenes = {"c4": ("dining_standin", dict(iarg=6)), "c5": ("million_standin", dict(iarg=1000000))}
seeds = M.launch_seeds(int(os.environ.get("SPP", "16")))
for name in os.environ.get("SCENES", "c5,c4").split(","):
    kind, kw = scenes[name]
    hs = M.HostScene(kind, 1920, 1080, **kw)
    for case in os.environ.get("CASES", "kernel_variant=3;kernel_variant=4,aux_depth=0,slots_in_use=512;kernel_variant=4,aux_depth=16,slots_in_use=512;kernel_variant=4,aux_depth=16,slots_in_use=448").split(";"):
        for o in case.split(","):
            k, v = o.split("="); ctx.set_option(k, int(v))
        ctx.load(hs)
        best = 1e9
        for rep in range(3):
            ctx.accum_clear(); ctx.kernel_time(reset=True); ctx.render(seeds); ms, n = ctx.kernel_time(); best = min(best, ms)
        print("%s %-60s %.1f ms (variant used %d)" % (name, case, best, ctx.get_option("kernel_variant_used")), flush=True)
