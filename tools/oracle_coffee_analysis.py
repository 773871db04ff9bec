"""Round-3 analysis of the oracle against the reference's demo/coffee.png.  Test infrastructure: uses oracle/.

  python tools/oracle_coffee_analysis.py [spp]          (about 2 minutes at 512 spp on 8 cores)

One oracle render of the 240x135 block frame per hypothesis, compared with tests/golden/coffee_8x.npy in the regions of
tests/test_oracle_kat.py.  Hypotheses (each a switch of the oracle or a change of the scene description, all off / unchanged
in the restated reference):
  stack     the 9608-byte OptiX stack overflows at recursion depth D and Exception.cu adds white instead of the sample
  order     unspecified C++ evaluation order of the rand() pairs (quad light, cosine_sample_hemisphere, camera jitter)
  specular  DisneyParams.specular 0.625 instead of initDisneyParams' 0.5
  pot       the lathe stand-in for the missing glass pot
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
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 512
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
    for name in ("order", "specular", "pot", "metal0", "floor001"):
        if name == "order":
            O.set_option("draw_order", 3)
        try:
            s2 = O.Scene(M.HostScene("coffee_pot_standin", 240, 135).to_dict()) if name == "pot" else sc if name == "order" else variant_scene(hs, name)
            acc, _ = s2.render(seeds)
        finally:
            O.set_option("draw_order", 0)
        img = O.image_from_accum(acc, spp).astype(np.float64)
        print("== variant %s: oracle - PNG per region (R G B)" % name)
        for n, d in region_means(img, gold).items():
            print("   %-30s %+.4f %+.4f %+.4f" % (n, *d))


if __name__ == "__main__":
    main()
