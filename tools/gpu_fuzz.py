"""Randomised consistency check on the GPU: every combination of scheduler options must give the bits of the
per-lane reference kernel (variant 0)."""
import os, sys, hashlib, random
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from common import M, tree_containment_errors, hostsim_render, hostsim_lib   # noqa: E402
ctx = M.Context(0)
rng = random.Random(int(os.environ.get("SEED", "1")))
cases = int(os.environ.get("CASES", "24"))
bad = 0
tree_cases = 0
only = int(os.environ.get("ONLY", "-1"))                 # replay one case of a seed (the draws of the others are made, nothing is rendered)
for case in range(cases):
    w, h = rng.choice([(64, 36), (101, 37), (200, 112), (320, 180), (33, 129), (8, 8), (640, 360)])
    spp = rng.choice([1, 2, 3, 5, 8, 17])
    scene = rng.choice(["file:coffee", "file:coffee", "dining_standin", "coffee_pot_standin", "million_standin", "random_spheres", "cornell_quads", "spheres"])
    kw = dict(iarg=2) if scene == "dining_standin" else dict(iarg=rng.choice([3000, 40000])) if scene == "million_standin" else \
        dict(iarg=rng.choice([97, 497])) if scene == "random_spheres" else dict(farg=rng.choice([0.0, 0.1])) if scene == "spheres" else {}
    rule = rng.choice([1, 1, 0])                          # shadow rule (moptix.h D5'): reference and candidate use the same one
    seed0 = rng.randrange(1000)
    ranks = rng.choice([1, 1, 2, 3, 8]); rank = rng.randrange(ranks)
    opts = dict(kernel_variant=rng.choice([3, 3, 4, 4]), leaf_size=rng.choice([1, 2, 4, 8]), tile_major=rng.choice([0, 1, 2, 3, 3]),
                swap_lanes=rng.choice([8, 24, 48]), starve_lanes=rng.choice([4, 16, 40]), blocks_per_cu=rng.choice([1, 2, 3]),
                sample_buffer_mb=rng.choice([1, 8192]), builder=rng.choice([0, 1, 1]), slots_in_use=rng.choice([-1, -1, 300, 64, 448, 509, 575, 576]),
                aux_depth=rng.choice([0, 1, 1, 2, 3, 16]), node_format=rng.choice([0, 64, 64, 128]),
                drain_below=rng.choice([0, 1, 5, 16, 64, 64]))      # round 6: when the packet kernel's workgroups hand their last paths to the drain kernel
    if only >= 0 and case != only:
        continue
    for o in os.environ.get("OVERRIDE", "").split(","):   # replay with some options changed: which one does a mismatch need?
        if "=" in o:
            opts[o.split("=")[0]] = int(o.split("=")[1])
    hs = M.HostScene(scene, w, h, **kw)
    seeds = M.launch_seeds(spp, seed0)
    ctx.set_partition(rank, ranks)
    ctx.set_option("kernel_variant", 0); ctx.set_option("leaf_size", 4); ctx.set_option("sample_buffer_mb", 8192)
    ctx.set_option("builder", 0); ctx.set_option("slots_in_use", -1); ctx.set_option("shadow_rule", rule)
    ctx.load(hs); ctx.accum_clear(); ctx.render(seeds); ref = ctx.accum_read()
    for k, v in opts.items():
        ctx.set_option(k, v)
    ctx.load(hs)
    ok = True
    for rep in range(2 if only < 0 else 6):               # second repetition uses the tile history
        ctx.accum_clear(); ctx.render(seeds)
        got = ctx.accum_read()
        ok = ok and np.array_equal(got, ref)
        if only >= 0:
            d = np.abs(got.astype(np.float64) - ref.astype(np.float64)).reshape(-1, got.shape[-1]).max(axis=1)
            print("  rep %d: %d pixels differ, largest difference %.3g (of mean %.3g) at %s" % (rep, int((d > 0).sum()), d.max(), float(np.abs(ref).mean()),
                  np.flatnonzero(d > 0)[:8].tolist()), flush=True)
    used = (ctx.get_option("kernel_variant_used"), ctx.get_option("node_format_used"))
    verdict = "ok"
    if not ok:
        # Is it the scheduler or the tree?  The reference's float triangle test (pt_geom.h tri_test = Geometry.cu:121-160) can accept a
        # grazing hit on a needle triangle at a point OUTSIDE that triangle's bounding box; whether a traversal ever tests the triangle
        # then depends on the boxes around it ("the one exception to rule D5").  The proof is made on the
        # CANDIDATE's own tree: the per-lane kernel (variant 0: no scheduler, no 64-byte decode, the branched stack tail) with the
        # candidate's leaf_size and builder must give the candidate's bits -- then scheduler, Node64 decode and builder agree with
        # the simplest kernel on that tree and only the tree's shape separates it from the reference.  Anything else is a MISMATCH.
        ctx.set_option("kernel_variant", 0)
        ctx.load(hs); ctx.accum_clear(); ctx.render(seeds)
        own = ctx.accum_read()
        if np.array_equal(own, got):
            # ... and the candidate's tree itself must be VALID: every child box, 128-byte and decoded 64-byte form, contains the
            # triangles below it -- a builder bug would make every kernel miss the same triangle and pass the check above
            nodes_, tris_, _p = ctx.debug_read_accel()
            invalid = tree_containment_errors(nodes_, tris_, 0 if len(nodes_) else -1, ctx.debug_read_nodes64())
            verdict = ("TREE-DEPENDENT HIT (%d pixels; variant 0 on the candidate's tree gives the candidate's bits; every box of that tree contains its triangles)"
                       % int((ref != got).any(axis=-1).sum())) if invalid == 0 else "MISMATCH (INVALID TREE: %d child boxes do not contain their triangles)" % invalid
        elif used == (4, 64):
            # the per-lane kernel walks the 128-byte boxes of the same tree; the candidate walked their quantised (larger) form
            ctx.set_option("kernel_variant", 4); ctx.set_option("node_format", 128)
            ctx.load(hs); ctx.accum_clear(); ctx.render(seeds)
            if np.array_equal(ctx.accum_read(), own):
                # The 64-byte nodes of this tree give another image than its 128-byte nodes: a decode bug -- or a grazing hit that only the
                # quantised (larger) boxes admit.  Proof of the second: the CPU build of the kernel's own code (tests/hostsim, plain per-lane
                # walk) on the SAME 64-byte nodes gives the candidate's values at the differing pixels, on the 128-byte nodes the reference's,
                # the tree is valid, and the damage is a pixel or two.
                diff = (got != own).any(axis=-1)
                hostsim_lib().hostsim_set_builder(int(opts.get("builder", 1)))
                h64, _c = hostsim_render(hs, seeds, leaf_size=int(opts.get("leaf_size", 4)), node_format=64)
                hostsim_lib().hostsim_set_builder(int(opts.get("builder", 1)))
                h128, _c = hostsim_render(hs, seeds, leaf_size=int(opts.get("leaf_size", 4)), node_format=128)
                hostsim_lib().hostsim_set_builder(1)
                nodes_, tris_, _p = ctx.debug_read_accel()
                valid = tree_containment_errors(nodes_, tris_, 0 if len(nodes_) else -1, ctx.debug_read_nodes64()) == 0
                same64 = float(np.abs(h64[diff] - got[diff]).max()) <= 1e-5 * spp
                same128 = float(np.abs(h128[diff] - own[diff]).max()) <= 1e-5 * spp
                verdict = ("TREE-DEPENDENT HIT (%d pixels; the 64-byte boxes admit a grazing hit the 128-byte boxes of the same tree cull: the CPU build of the kernel code gives the "
                           "candidate's values on the 64-byte nodes and the reference's on the 128-byte nodes; the tree is valid)" % int(diff.sum())) \
                    if (same64 and same128 and valid and int(diff.sum()) <= 2) else \
                    "MISMATCH (the 64-byte nodes differ from the 128-byte nodes of the same tree and the CPU build does not reproduce it: same64 %s same128 %s valid %s pixels %d -- replay with ONLY=%d)" % (same64, same128, valid, int(diff.sum()), case)
            else:
                verdict = "MISMATCH"
        else:
            verdict = "MISMATCH"
        tree_cases += verdict.startswith("TREE"); ok = verdict.startswith("TREE")
    print("case %3d %-18s %s %dx%d spp %d seed0 %d rank %d/%d rule %d %s ran %s -> %s" % (case, scene, kw, w, h, spp, seed0, rank, ranks, rule, opts, used,
                                                                                        verdict), flush=True)
    bad += 0 if ok else 1
# a ceiling on the excuse: round 4 found 1 tree-dependent pixel-sample in 3,600 cases
ceiling = max(1, cases // 400)
print("mismatches:", bad, " tree-dependent grazing hits:", tree_cases, "(ceiling %d)" % ceiling)
sys.exit(1 if bad or tree_cases > ceiling else 0)
