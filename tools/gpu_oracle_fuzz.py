"""The product path against the oracle on random inputs: scene kind, frame size, sample count, seeds, shadow rule; default kernel
choice.  Per case: RMSE of the per-sample mean (bar: 2e-6, the arithmetic contract's; north star: 1e-3), and the counts of rays and
closest hits, which must be EQUAL -- one path decision taken differently anywhere in the frame shows there.  The one accepted
exception is a tree-dependent grazing hit, proved as such per case (below) and counted apart under a ceiling.
SEED, CASES as tools/gpu_fuzz.py."""
import os, sys, random, time
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from common import M, O, oracle_scene, rmse, hostsim_render , tree_containment_errors  # noqa: E402
ctx = M.Context(0)
rng = random.Random(int(os.environ.get("SEED", "1")))
cases = int(os.environ.get("CASES", "24"))
bad = 0; worst = 0.0; tree_cases = 0
default_variant = ctx.get_option("kernel_variant")
for case in range(cases):
    w, h = rng.choice([(64, 36), (101, 37), (200, 112), (160, 90), (33, 129), (8, 8), (240, 135)])
    spp = rng.choice([1, 2, 3, 5])
    scene = rng.choice(["file:coffee", "file:coffee", "dining_standin", "coffee_pot_standin", "million_standin", "random_spheres", "cornell_quads", "spheres"])
    kw = dict(iarg=2) if scene == "dining_standin" else dict(iarg=rng.choice([3000, 40000])) if scene == "million_standin" else \
        dict(iarg=rng.choice([97, 497])) if scene == "random_spheres" else dict(farg=rng.choice([0.0, 0.1, 0.5])) if scene == "spheres" else {}
    rule = rng.choice([1, 1, 0])
    seed0 = rng.randrange(100000)
    hs = M.HostScene(scene, w, h, **kw)
    seeds = M.launch_seeds(spp, seed0)
    ctx.set_option("shadow_rule", rule)
    ctx.set_option("kernel_variant", rng.choice([default_variant, 3, 4, 4]))      # the library's own choice, or one of the two schedulers
    ctx.load(hs); ctx.accum_clear(); st = ctx.render_counted(seeds)
    g = ctx.accum_read()[..., :3]
    O.set_option("shadow_any_opaque_blocks", 1 if rule == 0 else 0)
    t0 = time.time()
    o, ost = oracle_scene(hs).render(seeds)
    O.set_option("shadow_any_opaque_blocks", 0)
    e = rmse(g / spp, o / spp)
    ok = e <= 2e-6 and st.rays == ost.rays and st.closestHits == ost.closestHits
    verdict = "ok" if ok else "MISMATCH"
    if not ok and hs.sizes.nFaces > 0:
        # The one documented exception to "decisions are equal" (DESIGN.md section 2): the reference's float triangle test can accept a
        # grazing hit at a point OUTSIDE the triangle's own box, and whether a traversal ever tests that triangle depends on the boxes
        # round it; the oracle's tree (and its brute-force mode) is another tree.  Accepted as such only when BOTH hold:
        #   (1) the CPU build of the kernel's own code on the device's tree (tests/hostsim: plain per-lane walk, same node format)
        #       gives the GPU's counts and image -- the GPU executes the specified algorithm on that tree;
        #   (2) the GPU itself gives exactly the oracle's counts and image (RMSE <= 2e-6) on another tree of the same triangles
        #       (64-byte nodes, whose boxes are supersets; leaves of 8 or of 1).
        used = (ctx.get_option("kernel_variant_used"), ctx.get_option("node_format_used"))
        hi, hc = hostsim_render(hs, seeds, node_format=used[1] if used[0] == 4 else 128)
        same_on_host = rmse(hi / spp, g / spp) <= 2e-6 and hc["closestHits"] == st.closestHits and hc["primaryRays"] + hc["bounceRays"] + hc["shadowRays"] == st.rays
        other = None
        for alt in (dict(kernel_variant=4, node_format=64), dict(kernel_variant=4, node_format=128), dict(leaf_size=8), dict(leaf_size=1)):
            c2 = M.Context(0); c2.set_option("shadow_rule", rule)
            for k, v in alt.items():
                c2.set_option(k, v)
            c2.load(hs); c2.accum_clear(); st2 = c2.render_counted(seeds); g2 = c2.accum_read()[..., :3]; c2.close()
            if rmse(g2 / spp, o / spp) <= 2e-6 and st2.rays == ost.rays and st2.closestHits == ost.closestHits:
                other = alt; break
        nodes_, tris_, _p = ctx.debug_read_accel()      # (3) the device's tree is a valid one: every child box contains the triangles below it
        valid_tree = tree_containment_errors(nodes_, tris_, 0 if len(nodes_) else -1, ctx.debug_read_nodes64()) == 0
        if same_on_host and other is not None and valid_tree:
            verdict = "TREE-DEPENDENT HIT (the CPU build of the kernel code gives the GPU's result on this tree; the GPU gives the oracle's with %s)" % other
            tree_cases += 1; ok = True
    worst = max(worst, e if verdict == "ok" else 0.0)
    print("case %3d %-18s %s %dx%d spp %d seed0 %d rule %d ran (%d, %d): rmse %.2e rays %d / %d closest hits %d / %d oracle %.1fs -> %s" % (
        case, scene, kw, w, h, spp, seed0, rule, ctx.get_option("kernel_variant_used"), ctx.get_option("node_format_used"), e, st.rays, ost.rays,
        st.closestHits, ost.closestHits, time.time() - t0, verdict), flush=True)
    bad += 0 if ok else 1
ceiling = max(1, cases // 400)
print("cases %d mismatches %d tree-dependent grazing hits %d (ceiling %d) worst rmse of the identical cases %.2e" % (cases, bad, tree_cases, ceiling, worst))
sys.exit(1 if bad or tree_cases > ceiling else 0)
