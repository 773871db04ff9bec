"""One fuzz case's differing pixel, sample by sample, against the oracle.  Environment: SCENE, IARG, W, H, SPP, SEED0, RANK, RANKS, RULE,
PIXEL (flat index in the frame), A / B = option sets "k=v,k=v"."""
import os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from common import M, O, oracle_scene   # noqa: E402
E_ = os.environ
w, h, spp = int(E_["W"]), int(E_["H"]), int(E_["SPP"])
hs = M.HostScene(E_["SCENE"], w, h, iarg=int(E_.get("IARG", "0")))
seeds = M.launch_seeds(spp, int(E_["SEED0"]))
pix = int(E_["PIXEL"]); py, px = pix // w, pix % w
rule = int(E_.get("RULE", "1"))
ctx = M.Context(0)
ctx.set_partition(int(E_.get("RANK", "0")), int(E_.get("RANKS", "1")))
def run(optset, sd):
    for o in optset.split(","):
        k, v = o.split("="); ctx.set_option(k, int(v))
    ctx.set_option("shadow_rule", rule)
    ctx.load(hs); ctx.accum_clear(); ctx.render(sd)
    return ctx.accum_read().reshape(h, w, -1)[py, px].astype(np.float64)
A, B = E_["A"], E_["B"]
osc = oracle_scene(hs)
O.set_option("shadow_any_opaque_blocks", 1 if rule == 0 else 0)
for i, s in enumerate(seeds):
    a, b = run(A, [s]), run(B, [s])
    if not np.array_equal(a, b):
        acc, st = osc.render([s], region=(px, py, px + 1, py + 1))
        print("sample %d seed %d: A %s  B %s  oracle %s" % (i, s, a[:3], b[:3], acc[py, px]), flush=True)
O.set_option("shadow_any_opaque_blocks", 0)
print("done")
