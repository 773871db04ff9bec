"""Round-3 analysis of the oracle against the reference's demo/coffee.png.  Test infrastructure: uses oracle/.

  python tools/oracle_coffee_analysis.py [spp] [--shadows]     (about 2 minutes at 512 spp on 8 cores; --shadows: six more renders)

One oracle render of the 240x135 block frame per hypothesis, compared with tests/golden/coffee_8x.npy in the regions of
tests/test_oracle_kat.py.  Hypotheses (each a switch of the oracle or a change of the scene description, all off / unchanged
in the restated reference):
  stack     the 9608-byte OptiX stack overflows at recursion depth D and Exception.cu adds white instead of the sample
  order     unspecified C++ evaluation order of the rand() pairs (quad light, cosine_sample_hemisphere, camera jitter)
  specular  DisneyParams.specular 0.625 instead of initDisneyParams' 0.5
  cos2ulp   disneyPdf's cosTheta = |N.H| short by two ulps (1.2e-7): what a -use_fast_math normalize() that leaves N and H one ulp
            short each does to GTR2's 1 + (a^2 - 1) cos^2 at a^2 = 1e-6 (oracle switch cos_short_tenth_ulp = 20)
  pot       the lathe stand-in for the missing glass pot, shadow rays decided by their nearest any-hit surface (the default since
            round 3: OptiX's acceptance semantics, DESIGN.md 2 rule D5) -- and pot_oldrule: the same scene under rounds 1-2's rule
            (an opaque surface anywhere on the segment blocks)
  metal0    the two Metal meshes made black (how much light reaches the floor by way of the chrome parts)
  floor001  Floor roughness 0.001 (sampling alpha = evaluation alpha: no 10x mismatch)
and the statistics of the per-sample clamp (Camera.cu:39): share of samples above 1 and the mean before the clamp."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from common import M, O      # noqa: E402

REGIONS = {"back wall, left": (10, 50, 20, 80), "back wall, right": (10, 50, 160, 190), "machine body, centre": (20, 60, 113, 128),
           "black base": (70, 76, 105, 135), "floor, middle left": (95, 110, 30, 70), "floor, bottom right": (115, 133, 170, 215),
           "band left": (20, 60, 103, 112), "band right": (20, 60, 132, 139), "floor, bottom left": (115, 133, 20, 70),
           "floor beside the base, left": (108, 128, 50, 92), "floor beside the base, right": (108, 128, 150, 190)}


def material_class(m):
    if m["kind"] != 3:
        return "other"
    return "orange" if abs(m["color"][1] - 0.37) < 1e-3 else "floor" if abs(m["color"][1] - 0.578) < 1e-3 else "metal" if m["metallic"] > 0.5 else "black"


def variant_scene(hs, name):
    d = hs.to_dict()
    for m in d["materials"]:
        c = material_class(m)
        if name == "specular" and c != "other":
            m["specular"] = 0.625
        if name == "metal0" and c == "metal":
            m["color"] = [0.0, 0.0, 0.0]
        if name == "floor001" and c == "floor":
            m["roughness"] = 0.001
    return O.Scene(d)


def region_means(img, gold):
    return {n: (img[y0:y1, x0:x1] - gold[y0:y1, x0:x1]).mean(axis=(0, 1)) for n, (y0, y1, x0, x1) in REGIONS.items()}


def main():
    spp = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 512
    gold = np.load(os.path.join(REPO, "tests", "golden", "coffee_8x.npy")).astype(np.float64)
    seeds = M.launch_seeds(spp)
    hs = M.HostScene("file:coffee", 240, 135)
    sc = O.Scene(hs.to_dict())
    cs, cn = sc.render_by_depth(seeds, 64)
    cs, cn = cs[::-1].astype(np.float64), cn[::-1].astype(np.float64)
    base = np.clip(cs.sum(axis=2) / spp, 0, 1)
    print("== restated reference: oracle - PNG per region (R G B)")
    for n, d in region_means(base, gold).items():
        print("   %-30s %+.4f %+.4f %+.4f" % (n, *d))
    print("== stack overflow at depth D (white instead of the sample): share of samples that deep, gap left in G")
    for D in (32, 16, 8, 6):
        img = np.clip((cs[:, :, :D].sum(axis=2) + cn[:, :, D:].sum(axis=2)[..., None]) / spp, 0, 1)
        for n in ("band right", "floor, bottom left", "black base"):
            y0, y1, x0, x1 = REGIONS[n]
            print("   D=%2d %-22s deep share %.4f  oracle - PNG (G) %+.4f" % (D, n, cn[y0:y1, x0:x1, D:].sum() / cn[y0:y1, x0:x1].sum(),
                                                                          (img[y0:y1, x0:x1, 1] - gold[y0:y1, x0:x1, 1]).mean()))
    print("== per-sample clamp (Camera.cu:39): share of samples above 1, mean before the clamp (each sample capped at 10), mean after")
    for n in ("back wall, left", "floor, bottom right", "floor, bottom left", "floor beside the base, left", "band right"):
        y0, y1, x0, x1 = REGIONS[n]
        raw, ncl, cl = sc.render_clamp_stats(seeds[:min(spp, 256)], (x0, 135 - y1, x1, 135 - y0), 10.0)
        N = min(spp, 256) * (y1 - y0) * (x1 - x0)
        print("   %-30s above 1: %s  before: %s  after: %s" % (n, np.round(ncl.sum(axis=(0, 1)) / N, 3), np.round(raw.sum(axis=(0, 1)) / N, 3),
                                                               np.round(cl.sum(axis=(0, 1)) / N, 3)))
    print("== displacement of the PNG against the render: coverage of the edge block by the brighter side, PNG vs oracle (G channel)")
    def cover(img, row0, row1, col, dark_col, bright_col):
        v, d, b = img[row0:row1, col, 1].mean(), img[row0:row1, dark_col, 1].mean(), img[row0:row1, bright_col, 1].mean()
        return (v - d) / (b - d)
    for what, rows, col, dcol, bcol, sign in (("left light panel, right edge", (30, 70), 10, 12, 8, +1), ("right light panel, left edge", (30, 70), 199, 197, 201, -1),
                                              ("machine body, left edge", (25, 55), 98, 96, 100, -1)):
        cg, co = cover(gold, rows[0], rows[1], col, dcol, bcol), cover(base, rows[0], rows[1], col, dcol, bcol)
        print("   %-32s PNG %.2f oracle %.2f of the block -> the PNG's edge sits %.1f pixels to the left" % (what, cg, co, sign * (co - cg) * 8))
    vt = [(img[0:12, 118:122, 1].mean(axis=1)) for img in (gold, base)]
    print("   machine top (column 118:122, rows 0..7) PNG    %s" % np.round(vt[0][:8], 2))
    print("   machine top (column 118:122, rows 0..7) oracle %s  -> about 3 pixels lower in the PNG" % np.round(vt[1][:8], 2))
    if "--shadows" in sys.argv:
        print("== is the floor's gap the machine's shadows, lighter?  least squares of (PNG - oracle) against the three lights' shadow depths (G)")
        S = []
        for k in range(3):
            imgs = []
            for machine in (0, 1):
                d = hs.to_dict()
                if not machine:
                    keep = np.array([material_class(d["materials"][m]) == "floor" for m in d["faceMat"]])
                    for key in ("vIdx", "nIdx", "tIdx", "faceMat"):
                        d[key] = d[key][keep]
                for i, l in enumerate(d["lights"]):
                    if i != k:
                        l["emission"] = [0, 0, 0]
                nl = len(d["lights"])
                for i, m in enumerate(d["materials"]):
                    if m["kind"] == 4 and i - (len(d["materials"]) - nl) != k:
                        m["emission"] = [0, 0, 0]
                acc, _ = O.Scene(d).render(seeds[:min(spp, 256)])
                imgs.append(O.image_from_accum(acc, min(spp, 256)).astype(np.float64)[..., 1])
            S.append(imgs[0] - imgs[1])
        res = (gold - base)[..., 1]
        mask = np.zeros((135, 240), bool); mask[100:135, 12:190] = True; mask[100:130, 90:152] = False
        A = np.stack([s_[mask] for s_ in S], axis=1); b = res[mask]
        c, *_ = np.linalg.lstsq(A, b, rcond=None)
        print("   coefficients per light (left, right, top) %s; residual rms %.4f of the gap's rms %.4f" % (np.round(c, 3), np.sqrt(((A @ c - b) ** 2).mean()), np.sqrt((b ** 2).mean())))
        for n in ("floor, bottom left", "floor beside the base, left", "floor beside the base, right", "floor, bottom right", "floor, middle left"):
            y0, y1, x0, x1 = REGIONS[n]
            print("   %-30s gap %+.4f  fitted %+.4f  shadow depths %s" % (n, res[y0:y1, x0:x1].mean(), sum(ck * s_[y0:y1, x0:x1].mean() for ck, s_ in zip(c, S)),
                                                                        np.round([s_[y0:y1, x0:x1].mean() for s_ in S], 3)))
    # comparable blocks: everything but the pot itself (its shape is unknown) and a two-block margin round strong edges of the PNG
    # (the PNG's frame is displaced by 3-5 pixels against the checkout's camera)
    lum = gold.mean(axis=2)
    edge = np.zeros_like(lum, bool)
    edge[:, 1:] |= np.abs(np.diff(lum, axis=1)) > 0.06; edge[1:, :] |= np.abs(np.diff(lum, axis=0)) > 0.06
    grown = edge.copy()
    for dy in range(-2, 3):
        for dx in range(-2, 3):
            grown |= np.roll(np.roll(edge, dy, axis=0), dx, axis=1)
    comparable = ~grown
    comparable[74:120, 94:160] = False                     # the glass pot and its handle
    census = {}

    def take_census(tag, img):
        dd = np.abs(img - gold).max(axis=2)[comparable]
        census[tag] = (int(comparable.sum()), float((dd <= 0.005).mean()), float((dd <= 0.01).mean()), float((dd <= 0.02).mean()), float(dd.max()))
    take_census("shipped scene, restated reference", base)
    for name in ("order", "specular", "cos2ulp", "pot", "pot+cos2ulp", "pot_oldrule", "metal0", "floor001"):
        if name == "order":
            O.set_option("draw_order", 3)
        if name == "pot_oldrule":
            O.set_option("shadow_any_opaque_blocks", 1)
        if name in ("cos2ulp", "pot+cos2ulp"):
            O.set_option("cos_short_tenth_ulp", 20)
        try:
            s2 = O.Scene(M.HostScene("coffee_pot_standin", 240, 135).to_dict()) if name.startswith("pot") else sc if name in ("order", "cos2ulp") else variant_scene(hs, name)
            acc, _ = s2.render(seeds)
        finally:
            O.set_option("draw_order", 0); O.set_option("shadow_any_opaque_blocks", 0); O.set_option("cos_short_tenth_ulp", 0)
        img = O.image_from_accum(acc, spp).astype(np.float64)
        print("== variant %s: oracle - PNG per region (R G B)" % name)
        for n, d in region_means(img, gold).items():
            print("   %-30s %+.4f %+.4f %+.4f" % (n, *d))
        take_census(name, img)
    print("== census over the comparable 8x8-pixel blocks (max over R, G, B of |oracle - PNG| per block; %d spp leave ~0.004 of noise per block)" % spp)
    for tag, (n, a, b, c, mx) in census.items():
        print("   %-36s %5d blocks: %.3f within 0.005, %.3f within 0.01, %.3f within 0.02, worst %.3f" % (tag, n, a, b, c, mx))


if __name__ == "__main__":
    main()
