"""Generates the committed golden fixtures from the reference checkout (run in the build
container only; /root/reference does not exist on the GPU box).

  spheres_lens_8x.npy / spheres_pinhole_8x.npy
      8x8 box-downsampled float32 copies (135x240x3, values in [0,1]) of the reference's
      demo/spheres_lens.png and demo/spheres_pinhole.png -- the only images in the reference
      whose scene is fully defined in code (MinimalOptiX.cpp:156-257).  They are *data derived
      from the reference's own outputs*, used as a statistical pin of the whole chain
      (camera, intersectors, lambertian/metal/glass, miss, clamp, normalise, flip).
  coffee_8x.npy
      same for demo/coffee.png (glass pot present there, missing in the shipped assets:
      only the regions away from the pot are compared).
  kat.json
      integer known-answer vectors for tea<16>/lcg/rand and the setCamParams vectors of
      SURVEY.md Appendix A3 (generated there from the reference's code).
"""
import json
import os

import numpy as np
from PIL import Image

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def down8(path):
    im = np.asarray(Image.open(path).convert("RGB"), dtype=np.float32) / 255.0
    h, w, _ = im.shape
    return im.reshape(h // 8, 8, w // 8, 8, 3).mean(axis=(1, 3)).astype(np.float32)


def main():
    for name in ("spheres_lens", "spheres_pinhole", "coffee"):
        np.save(os.path.join(HERE, name + "_8x.npy"), down8(os.path.join(REF, "demo", name + ".png")))
    kat = {
        "tea16": [[0, 0, 0x741c187d], [1, 0, 0x8da6b311], [0, 1, 0x70d3aef1], [12345, 42, 0xa353d458],
                  [1920 * 1080 - 1, 0x7fffffff, 0xfdeb947b], [5, (-7) & 0xffffffff, 0x6825414a]],
        "lcg_from_0xa353d458": [[0xaae3cbd7, 14928855, 0.889829099], [0x9536f74a, 3602250, 0.214710832],
                                [0xdcfafe21, 16449057, 0.980440199], [0x47a8010c, 11010316, 0.656265974]],
        "lcg_from_0": [[0x3c6ef35f, 7271263], [0x47502932, 5253426], [0xd1ccf6e9, 13432553]],
        "cam_spheres": {"from": [3, 3, 2], "at": [0, 0, -1], "fov": 20, "aspect": 1920 / 1080, "aperture": 0.5,
                        "focus": 27 ** 0.5, "origin": [3, 3, 2], "horizontal": [2.303526, 0, -2.303526],
                        "vertical": [-0.74809206, 1.4961841, -0.74809206],
                        "scrLowerLeftCorner": [-0.7777171, -0.7480924, 0.5258088],
                        "u": [0.70710677, 0, -0.70710677], "v": [-0.4082483, 0.8164966, -0.4082483], "lensRadius": 0.25},
        "cam_coffee": {"extent": [2, 0.811135, 2.09417], "fov": 45, "origin": [0, 0.17844969, 0.5235425],
                       "horizontal": [1.4727594, 0, 0], "vertical": [0, 0.82828164, -0.01553028],
                       "scrLowerLeftCorner": [-0.7363797, -0.25443783, -0.4685167], "u": [1, 0, 0],
                       "v": [0, 0.99982435, -0.0187467], "lensRadius": 0},
        "coffee_meshes": {"Mesh000": [8150, 10848], "Mesh001": [22809, 33694], "Mesh002": [1808, 3456], "Mesh003": [1378, 2624],
                          "Mesh004": [225, 384], "Mesh005": [8968, 3283], "Mesh006": [1186, 2304], "Mesh007": [728, 1452],
                          "Mesh008": [6656, 12288], "Mesh009": [12866, 25216], "Mesh011": [162, 320], "Mesh012": [17922, 35200],
                          "Mesh013": [7296, 14592], "Mesh014": [6018, 11776], "Mesh015": [2304, 4096], "Mesh016": [1152, 2304],
                          "Mesh017": [728, 1452], "Mesh018": [728, 1452], "Mesh019": [728, 1452]},
    }
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(kat, f, indent=1)


if __name__ == "__main__":
    main()
