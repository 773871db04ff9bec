"""Randomised consistency check on the GPU: every combination of scheduler options must give the bits of the
per-lane reference kernel (variant 0)."""
import os, sys, hashlib, random
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from common import M   # noqa: E402
ctx = M.Context(0)
rng = random.Random(int(os.environ.get("SEED", "1")))
cases = int(os.environ.get("CASES", "24"))
bad = 0
for case in range(cases):
    w, h = rng.choice([(64, 36), (101, 37), (200, 112), (320, 180), (33, 129), (8, 8), (640, 360)])
    spp = rng.choice([1, 2, 3, 5, 8, 17])
    scene = rng.choice(["file:coffee", "file:coffee", "dining_standin", "coffee_pot_standin"])
    kw = dict(iarg=2) if scene == "dining_standin" else {}
    hs = M.HostScene(scene, w, h, **kw)
    seeds = M.launch_seeds(spp, rng.randrange(1000))
    ranks = rng.choice([1, 1, 2, 3, 8]); rank = rng.randrange(ranks)
    ctx.set_partition(rank, ranks)
    ctx.set_option("kernel_variant", 0); ctx.set_option("leaf_size", 4); ctx.set_option("sample_buffer_mb", 8192)
    ctx.set_option("builder", 0); ctx.set_option("slots_in_use", -1)
    ctx.load(hs); ctx.accum_clear(); ctx.render(seeds); ref = ctx.accum_read()
    opts = dict(kernel_variant=rng.choice([3, 3, 4, 4]), leaf_size=rng.choice([1, 2, 4, 8]), tile_major=rng.choice([0, 1, 2, 3, 3]),
                swap_lanes=rng.choice([8, 24, 48]), starve_lanes=rng.choice([4, 16, 40]), blocks_per_cu=rng.choice([1, 2, 3]),
                sample_buffer_mb=rng.choice([1, 8192]), builder=rng.choice([0, 1, 1]), slots_in_use=rng.choice([-1, -1, 300, 64, 448, 509, 575, 576]),
                aux_depth=rng.choice([0, 1, 1, 2, 3, 16]), node_format=rng.choice([0, 64, 64, 128]))
    for k, v in opts.items():
        ctx.set_option(k, v)
    ctx.load(hs)
    ok = True
    for rep in range(2):                                  # second repetition uses the tile history
        ctx.accum_clear(); ctx.render(seeds)
        ok = ok and np.array_equal(ctx.accum_read(), ref)
    print("case %2d %-16s %dx%d spp %d rank %d/%d %s -> %s" % (case, scene, w, h, spp, rank, ranks, opts, "ok" if ok else "MISMATCH"), flush=True)
    bad += 0 if ok else 1
print("mismatches:", bad)
sys.exit(1 if bad else 0)
