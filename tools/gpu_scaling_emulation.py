"""Strong-scaling estimate on ONE GPU: renders each rank's tile partition of the coffee frame in turn and
reports max-over-ranks time against the 1-GPU frame (load balance + per-launch constants; no communication)."""
import os, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from common import M   # noqa: E402
ctx = M.Context(0)
W, H = 1920, 1080
spp = int(os.environ.get("SPP", "256"))
hs = M.HostScene("file:coffee", W, H)
seeds = M.launch_seeds(spp)
for o in os.environ.get("OPTS", "").split(","):
    if "=" in o:
        k, v = o.split("="); ctx.set_option(k, int(v))
NS = [int(x) for x in os.environ.get("NS", "2,4,8").split(",")]
PIPE = os.environ.get("PIPE", "0") == "1"          # two frames in flight: two contexts render alternate frames of the share (bench.py's default from 8 ranks on)
ctx2 = M.Context(0) if PIPE else None
if PIPE:                                           # two frames in flight run without the drain kernel (bench.py GpuFrame.set_mode says why)
    ctx.set_option("drain_below", 0); ctx2.set_option("drain_below", 0)
if ctx2:
    for o in os.environ.get("OPTS", "").split(","):
        if "=" in o:
            k, v = o.split("="); ctx2.set_option(k, int(v))
def timed_pipelined():
    import time
    cs = [ctx, ctx2]
    for c in cs:                                                      # history for the work order, code loaded
        c.accum_clear(); c.render(seeds)
    k = int(os.environ.get("FRAMES", "12"))
    t0 = time.perf_counter()
    for i in range(k):
        c = cs[i % 2]
        c.sync(); c.accum_clear(); c.render_async(seeds)
    for c in cs:
        c.sync()
    return (time.perf_counter() - t0) * 1e3 / k
def timed():
    if PIPE:
        return timed_pipelined()
    best = 1e9
    for rep in range(int(os.environ.get('REPS', '4'))):
        ctx.accum_clear(); ctx.kernel_time(reset=True); ctx.render(seeds); ms, n = ctx.kernel_time(); best = min(best, ms + ctx.reduce_time())
    return best
ctx.load(hs)
if ctx2: ctx2.load(hs)
PIPE, pipe_wanted = False, PIPE                     # the one-GPU frame is the reference either way: one frame at a time
t1 = timed()
PIPE = pipe_wanted
print("N=1: %.1f ms" % t1, flush=True)
SPLIT = os.environ.get("SPLIT", "tile")            # tile: rank r's tiles, all launches; sample: whole frame, launches i = r mod n
all_seeds = seeds
for n in NS:
    ts = []
    for r in range(n):
        if SPLIT == "sample":
            seeds = all_seeds[r::n]; ctx.set_partition(0, 1)
        else:
            ctx.set_partition(r, n)
            if ctx2: ctx2.set_partition(r, n)
        ctx.load(hs)
        if ctx2: ctx2.load(hs)
        ts.append(timed())
    print("N=%d: max %.1f ms min %.1f ms -> efficiency %.1f %% (compute only)  per rank: %s" % (n, max(ts), min(ts), 100 * t1 / (n * max(ts)), " ".join("%.1f" % t for t in ts)), flush=True)
