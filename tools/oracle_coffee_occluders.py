"""Which occluder casts the shadows that differ from demo/coffee.png?  (test infrastructure: uses oracle/.)

The oracle's analysis switch noshadow_first / noshadow_last lets shadow rays pass the materials (= meshes) in an index range.
Shipped coffee scene, oracle - PNG (G channel) on the floor regions, with the shadows of one mesh or one class switched off:
the gap responds to the parts that sit round the place of the (missing) glass pot -- its lid, the metal band, the base plate."""
import os
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from common import M, O, oracle_scene      # noqa: E402

gold = np.load(os.path.join(REPO, "tests", "golden", "coffee_8x.npy")).astype(np.float64)
hs = M.HostScene("file:coffee", 240, 135)
sc = oracle_scene(hs)
d = hs.to_dict()
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 384
R = {"floor, middle left": (95, 110, 30, 70), "floor, bottom right": (115, 133, 170, 215), "floor, bottom left": (115, 133, 20, 70),
     "beside the base, left": (108, 128, 50, 92), "beside the base, right": (108, 128, 150, 190)}
seeds = M.launch_seeds(spp)
pos, vi, fm = d["positions"], d["vIdx"], d["faceMat"]
print("# oracle - PNG (G), shipped coffee scene, %d spp; columns: %s" % (spp, " | ".join(R)))
cases = [(1, 0, "all shadows on")] + [(i, i, "mesh of material %d" % i) for i in range(18)] + [(0, 9, "all Plastic_Black"), (10, 15, "all Plastic_Orange"), (16, 17, "all Metal")]
try:
    for a, b, name in cases:
        O.set_option("noshadow_first", a); O.set_option("noshadow_last", b)
        out = []
        for n, (y0, y1, x0, x1) in R.items():
            acc = np.zeros((135, 240, 3), np.float32)
            sc.render(seeds, accum=acc, region=(x0, 135 - y1, x1, 135 - y0))
            out.append((O.image_from_accum(acc, spp)[y0:y1, x0:x1] - gold[y0:y1, x0:x1]).mean(axis=(0, 1))[1])
        box = ""
        if a == b:
            p = pos[vi[fm == a].reshape(-1)]
            box = "  y %.3f..%.3f, |x|,|z| <= %.3f" % (p[:, 1].min(), p[:, 1].max(), np.abs(p[:, [0, 2]]).max())
        print("%-22s %s%s" % (name, " ".join("%+.4f" % v for v in out), box), flush=True)
finally:
    O.set_option("noshadow_first", 1); O.set_option("noshadow_last", 0)
