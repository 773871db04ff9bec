"""Pins the CPU oracle: integer/camera known-answer vectors and the reference's
own demo renders of the hard-coded spheres scene (tests/golden/*_8x.npy)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from common import M, O, oracle_scene, rmse

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KAT = json.load(open(os.path.join(GOLD, "kat.json")))
L = O.lib()


def test_tea16_known_answers():
    for v0, v1, want in KAT["tea16"]:
        assert L.orc_tea16(v0, v1) == want


def test_lcg_and_rand_known_answers():
    s = C.c_int32(np.uint32(0xa353d458).astype(np.int32))
    for state, bits24, r in KAT["lcg_from_0xa353d458"]:
        got = L.orc_lcg(C.byref(s))
        assert (s.value & 0xffffffff) == state and got == bits24
        assert abs(got / 16777216.0 - r) < 1e-9
    s = C.c_int32(0)
    for state, bits24 in KAT["lcg_from_0"]:
        assert L.orc_lcg(C.byref(s)) == bits24 and (s.value & 0xffffffff) == state
    s = C.c_int32(np.uint32(0xa353d458).astype(np.int32))
    assert L.orc_rand(C.byref(s)) == np.float32(14928855) / np.float32(16777216)


def test_launch_seed_schedule_matches_host():
    assert list(O.launch_seeds(5, 7)) == list(M.launch_seeds(5, 7))
    assert int(O.launch_seeds(1)[0]) == np.uint32(0x741c187d).astype(np.int32)


def _cam(frm, at, fov, aspect, aperture, focus):
    cam = O.OrcCam()
    L.orc_set_cam_params(O.f3(*frm), O.f3(*at), O.f3(0, 1, 0), fov, aspect, aperture, focus, C.byref(cam))
    return cam


@pytest.mark.parametrize("key", ["cam_spheres", "cam_coffee"])
def test_set_cam_params_known_answers(key):
    k = KAT[key]
    if key == "cam_spheres":
        cam = _cam(k["from"], k["at"], k["fov"], k["aspect"], k["aperture"], k["focus"])
    else:
        ext = np.float32(k["extent"])
        frm = np.float32([0, np.float32(0.22 * float(ext[1])), np.float32(0.25 * float(ext[2]))])
        at = frm + np.float32([0, -0.01875, -1])
        cam = _cam(frm, at, k["fov"], 1920 / 1080, 0.0, 1.0)
    for f in ("origin", "horizontal", "vertical", "scrLowerLeftCorner", "u", "v"):
        assert np.allclose(list(getattr(cam, f)), k[f], atol=2e-6), f
    assert cam.lensRadius == k["lensRadius"]


def test_host_cam_and_quad_params_equal_oracle_bitwise():
    """The product's host helpers (utils_host.cpp:67-99 equivalents) and the oracle's agree bit for bit."""
    H = M._capi.host_lib()
    rng = np.random.default_rng(3)
    for _ in range(20):
        frm, at = rng.normal(size=3).astype(np.float32) * 5, rng.normal(size=3).astype(np.float32)
        fov, asp, ap, foc = float(rng.uniform(10, 80)), 16 / 9, float(rng.uniform(0, 1)), float(rng.uniform(0.5, 30))
        a = _cam(frm, at, fov, asp, ap, foc)
        b = M._capi.CamParams()
        f3 = (C.c_float * 3)
        H.mohost_set_cam_params(f3(*frm), f3(*at), f3(0, 1, 0), fov, asp, ap, foc, C.byref(b))
        for f in ("origin", "horizontal", "vertical", "scrLowerLeftCorner", "u", "v"):
            assert list(getattr(a, f)) == getattr(b, f).tolist(), f
        anchor, v1, v2 = (rng.normal(size=3).astype(np.float32) for _ in range(3))
        qa = O.OrcQuad(); L.orc_set_quad_params(O.f3(*anchor), O.f3(*v1), O.f3(*v2), C.byref(qa))
        qb = M._capi.QuadParams(); H.mohost_set_quad_params(f3(*anchor), f3(*v1), f3(*v2), C.byref(qb))
        assert list(qa.plane) == [qb.plane.x, qb.plane.y, qb.plane.z, qb.plane.w]
        assert list(qa.v1) == qb.v1.tolist() and list(qa.v2) == qb.v2.tolist() and list(qa.anchor) == qb.anchor.tolist()


@pytest.mark.parametrize("aperture,name", [(0.5, "spheres_lens"), (0.0, "spheres_pinhole")])
def test_oracle_reproduces_reference_demo_image(aperture, name):
    """demo/spheres_lens.png is MinimalOptiX.cpp:156-257 as committed (aperture 0.5); the pinhole
    variant is aperture 0.  8-bit PNG + unknown spp => statistical comparison."""
    gold = np.load(os.path.join(GOLD, name + "_8x.npy"))
    hs = M.HostScene("spheres", 480, 270, farg=aperture)
    spp = 48
    acc, st = oracle_scene(hs).render(M.launch_seeds(spp))
    img = O.image_from_accum(acc, spp).reshape(135, 2, 240, 2, 3).mean(axis=(1, 3))
    assert np.abs(img.mean(axis=(0, 1)) - gold.mean(axis=(0, 1))).max() < 2e-3      # frame mean colour
    assert np.abs(img - gold).mean() < 1.2e-2                                        # per-block mean abs diff
    blocks = lambda a: a.reshape(9, 15, 16, 15, 3).mean(axis=(1, 3))                 # 9x16 region means
    assert np.abs(blocks(img) - blocks(gold)).max() < 1.5e-2


# demo/coffee.png against the oracle, block by block.  One fixture block = 8x8 pixels of the 1920x1080 PNG; the oracle
# renders the same footprint as one jittered pixel of a 240x135 frame (Camera.cu:28-29 box-filters the pixel), 1024 spp.
# Regions are [y0:y1, x0:x1] in blocks, row 0 = top.  The glass pot (Mesh010.obj, absent from the checkout) and a
# two-block margin around silhouettes (the PNG sits about one pixel to the left of the render) are left out.
COFFEE_CLEAN = {          # what the reference image pins: (|signed mean| bound, mean |block diff| bound)
    "back wall, left": ((10, 50, 20, 80), 1.5e-3, 5e-3),       # Disney diffuse lobe + NEE over three quad lights + MIS + clamp
    "back wall, right": ((10, 50, 160, 190), 1.5e-3, 5e-3),
    "machine body, centre": ((20, 60, 113, 128), 6e-3, 1.2e-2),   # Plastic_Orange away from the lights' reflections (seams inside)
    "black base": ((70, 76, 105, 135), 3e-3, 1e-2),
    "floor, middle left": ((95, 110, 30, 70), 5e-3, 1.2e-2),
    "floor, bottom right": ((115, 133, 170, 215), 5e-3, 1.8e-2),
}
COFFEE_KNOWN = {          # where the PNG and the restated formulas differ, measured and kept visible (DESIGN.md "coffee.png pin")
    # reflection of the side lights in Plastic_Orange (roughness 0.001): the PNG is ~1.22x brighter in G/B (R is saturated)
    "light reflected in the body, left": ((20, 60, 101, 109), (-0.09, -0.02)),
    "light reflected in the body, right": ((20, 60, 131, 139), (-0.09, -0.02)),
    # floor in front of the machine: the PNG is brighter by 0.018, all channels alike -- light that reaches the floor THROUGH the glass
    # pot, which the checkout does not ship (Mesh010.obj); with a stand-in pot it closes: test_floor_round_the_machine_is_lit_through_the_glass_pot
    "floor, bottom left": ((115, 133, 20, 70), (-0.03, -0.008)),
}


def _coffee_region(sc, seeds, box):
    y0, y1, x0, x1 = box
    acc = np.zeros((135, 240, 3), np.float32)
    sc.render(seeds, accum=acc, region=(x0, 135 - y1, x1, 135 - y0))       # accuBuffer rows run bottom-up
    return O.image_from_accum(acc, len(seeds))[y0:y1, x0:x1]


def test_oracle_coffee_matches_reference_demo_blockwise():
    gold = np.load(os.path.join(GOLD, "coffee_8x.npy"))           # 135 x 240 blocks
    sc = oracle_scene(M.HostScene("file:coffee", 240, 135))
    seeds = M.launch_seeds(1024)
    for name, (box, signed_tol, abs_tol) in COFFEE_CLEAN.items():
        y0, y1, x0, x1 = box
        d = _coffee_region(sc, seeds, box) - gold[y0:y1, x0:x1]
        assert np.abs(d.mean(axis=(0, 1))).max() < signed_tol, (name, d.mean(axis=(0, 1)))
        assert np.abs(d).mean() < abs_tol, (name, np.abs(d).mean())
    for name, (box, (lo, hi)) in COFFEE_KNOWN.items():
        y0, y1, x0, x1 = box
        d = (_coffee_region(sc, seeds, box) - gold[y0:y1, x0:x1]).mean(axis=(0, 1))
        assert lo < d[1] < hi and lo < d[2] < hi, (name, d)
        if "body" in name:
            assert abs(d[0]) < 6e-3, (name, d)                                        # red is saturated on both sides
    # the two light panels themselves are saturated in both
    for xs in (slice(0, 4), slice(236, 240)):
        assert gold[20:100, xs].min() > 0.999


def test_floor_round_the_machine_is_lit_through_the_glass_pot():
    """Round 3's answer to the floor regions of COFFEE_KNOWN.  coffee.scene's glass pot (Mesh010.obj) is missing from the checkout;
    in the reference's image the floor round the machine is up to 0.03 brighter than in a render of the shipped meshes.  With
    a lathe stand-in for the pot AND OptiX's any-hit semantics (a shadow ray is decided by its nearest any-hit surface: the
    glass pot accepts the ray and the lid / body behind it never block, DESIGN.md 2 rule D5) the oracle meets the PNG there;
    with the stand-in but the old rule (an opaque surface anywhere blocks) nothing moves."""
    gold = np.load(os.path.join(GOLD, "coffee_8x.npy"))
    seeds = M.launch_seeds(384)
    regions = {"floor, bottom left": (115, 133, 20, 70), "beside the base, left": (108, 128, 50, 92), "beside the base, right": (108, 128, 150, 190)}
    shipped = oracle_scene(M.HostScene("file:coffee", 240, 135))
    potted = oracle_scene(M.HostScene("coffee_pot_standin", 240, 135))

    def gaps(sc):
        out = {}
        for n, box in regions.items():
            y0, y1, x0, x1 = box
            out[n] = float((_coffee_region(sc, seeds, box) - gold[y0:y1, x0:x1]).mean())
        return out
    g_shipped, g_potted = gaps(shipped), gaps(potted)
    try:
        O.set_option("shadow_any_opaque_blocks", 1)
        g_old_rule = gaps(potted)
    finally:
        O.set_option("shadow_any_opaque_blocks", 0)
    for n in regions:
        assert g_shipped[n] < -0.014, (n, g_shipped)                  # the gap of the shipped scene
        assert abs(g_potted[n]) < 0.008, (n, g_potted)                # closed (the stand-in is not the real pot's shape: a few 1e-3 remain)
        assert abs(g_old_rule[n] - g_shipped[n]) < 0.003, (n, g_old_rule)


def _coffee_comparable_blocks(gold):
    """Everything but the glass pot itself (its shape is unknown: Mesh010.obj is missing upstream) and a two-block margin round
    strong edges of the PNG, whose frame is displaced by 3-5 pixels against the checkout's camera."""
    lum = gold.mean(axis=2)
    edge = np.zeros_like(lum, bool)
    edge[:, 1:] |= np.abs(np.diff(lum, axis=1)) > 0.06
    edge[1:, :] |= np.abs(np.diff(lum, axis=0)) > 0.06
    grown = edge.copy()
    for dy in range(-2, 3):
        for dx in range(-2, 3):
            grown |= np.roll(np.roll(edge, dy, axis=0), dx, axis=1)
    comparable = ~grown
    comparable[74:120, 94:160] = False
    return comparable


def test_coffee_census_over_all_comparable_blocks_is_frozen():
    """The parity chapter as a regression test: of the 32,400 8x8-pixel blocks of demo/coffee.png 23,386 are
    comparable; per block, max over R, G, B of |oracle - PNG|.  profiles/r03_oracle_coffee_analysis.txt has the table at 1,024 spp
    (shipped scene 91.3 % within 0.02; pot stand-in + nearest-any-hit shadow rule + two-ulp cosine 98.0 %); here at 384 spp, where
    ~0.007 of sampling noise per block lowers both (88.2 % / 93.0 % within 0.02, 94.0 % / 98.2 % within 0.03).  An oracle edit
    that moves the image against the reference's PNG fails here."""
    gold = np.load(os.path.join(GOLD, "coffee_8x.npy")).astype(np.float64)
    comparable = _coffee_comparable_blocks(gold)
    assert int(comparable.sum()) == 23386
    spp = 384
    seeds = M.launch_seeds(spp)

    def census(kind):
        acc, _ = oracle_scene(M.HostScene(kind, 240, 135)).render(seeds)
        dd = np.abs(O.image_from_accum(acc, spp).astype(np.float64) - gold).max(axis=2)[comparable]
        return float((dd <= 0.02).mean()), float((dd <= 0.03).mean())
    shipped = census("file:coffee")
    try:
        O.set_option("cos_short_tenth_ulp", 20)
        fitted = census("coffee_pot_standin")
    finally:
        O.set_option("cos_short_tenth_ulp", 0)
    assert shipped[0] >= 0.875 and shipped[1] >= 0.935, shipped
    assert fitted[0] >= 0.925 and fitted[1] >= 0.977, fitted
    assert fitted[0] - shipped[0] >= 0.035, (shipped, fitted)          # what the pot + the shadow rule + the cosine explain


def test_stack_overflow_exception_does_not_explain_the_coffee_gap():
    """Hypothesis tested in round 3: the reference's 9608-byte OptiX stack (MinimalOptiX.cpp:134) overflows at some recursion
    depth D and Exception.cu adds white instead of the sample.  render_by_depth gives the image for every D from one render:
    in the right-hand reflection band fewer than 1 % of the samples are deeper than 8 bounces, so even white for all of them
    moves the band by < 0.01 of the 0.045 (G) / 0.07 (B) that separate the oracle from the PNG."""
    gold = np.load(os.path.join(GOLD, "coffee_8x.npy"))
    sc = oracle_scene(M.HostScene("file:coffee", 240, 135))
    spp = 192
    y0, y1, x0, x1 = 20, 60, 132, 139
    cs, cn = sc.render_by_depth(M.launch_seeds(spp), 16, region=(x0, 135 - y1, x1, 135 - y0))
    cs, cn = cs[::-1][y0:y1, x0:x1].astype(np.float64), cn[::-1][y0:y1, x0:x1].astype(np.float64)
    assert cn.sum() == spp * (y1 - y0) * (x1 - x0)
    plain = np.clip(cs.sum(axis=2) / spp, 0, 1)
    gap = (plain - gold[y0:y1, x0:x1]).mean(axis=(0, 1))
    assert gap[1] < -0.035 and gap[2] < -0.055
    deep = cn[:, :, 8:].sum() / cn.sum()
    assert deep < 0.01
    white8 = np.clip((cs[:, :, :8].sum(axis=2) + cn[:, :, 8:].sum(axis=2)[..., None]) / spp, 0, 1)
    assert np.abs((white8 - plain).mean(axis=(0, 1))).max() < 0.01


def test_floor_values_are_means_of_clamped_samples():
    """Camera.cu:39 clamps every sample before it is added.  On the lit floor a third of the samples are above 1 and the mean
    before the clamp is about twice the mean after it: what the image shows there is E[min(X, 1)], a functional of the whole
    sample distribution -- which is why the oracle reproduces the reference's draw order and estimator quirks."""
    sc = oracle_scene(M.HostScene("file:coffee", 240, 135))
    y0, y1, x0, x1 = 118, 130, 180, 200
    spp = 64
    raw, ncl, cl = sc.render_clamp_stats(M.launch_seeds(spp), (x0, 135 - y1, x1, 135 - y0), 10.0)
    n = spp * (y1 - y0) * (x1 - x0)
    share, before, after = ncl.sum() / (3 * n), raw.sum() / (3 * n), cl.sum() / (3 * n)
    assert 0.3 < share < 0.55 and before > 1.8 * after and 0.6 < after < 0.8


def test_specular_0625_closes_the_reflection_bands():
    """The one change found that reproduces the PNG in the reflections of the side lights in Plastic_Orange: DisneyParams.specular
    0.625 instead of initDisneyParams' 0.5 (utils_host.cpp:107) -- Cspec0 0.05 instead of 0.04.  It leaves the other regions where
    they were (tools/oracle_coffee_analysis.py, profiles/r03_oracle_coffee_analysis.txt); the checkout's value stays the oracle's."""
    gold = np.load(os.path.join(GOLD, "coffee_8x.npy"))
    hs = M.HostScene("file:coffee", 240, 135)
    d = hs.to_dict()
    for m in d["materials"]:
        if m["kind"] == O.ORC_DISNEY:
            m["specular"] = 0.625
    seeds = M.launch_seeds(256)
    for sc, lo, hi in ((O.Scene(d), -0.02, 0.006), (oracle_scene(hs), -0.08, -0.035)):
        for box in ((20, 60, 103, 112), (20, 60, 132, 139)):
            y0, y1, x0, x1 = box
            dd = (_coffee_region(sc, seeds, box) - gold[y0:y1, x0:x1]).mean(axis=(0, 1))
            assert lo < dd[1] < hi and lo < dd[2] < hi, (box, dd)


def test_two_ulp_cosine_shortfall_closes_the_reflection_bands_too():
    """The other candidate for the reflection bands, with a mechanism in the reference's build: -use_fast_math (utils_host.cpp:32)
    makes normalize() a product with an approximate reciprocal square root, so N and H are unit vectors only to ~1e-7, and
    disneyPdf's GTR2 forms 1 + (a^2 - 1) cos^2 with a^2 = 1e-6: one ulp of the cosine is 12 % of that sum at the lobe's peak.  A
    SYSTEMATIC shortfall of two ulps (|N||H| = 1 - 1.2e-7) puts both bands within 0.012 of the PNG and moves nothing else
    (analysis switch cos_short_tenth_ulp; the product path stays correctly rounded, which is the parity contract)."""
    gold = np.load(os.path.join(GOLD, "coffee_8x.npy"))
    sc = oracle_scene(M.HostScene("file:coffee", 240, 135))
    seeds = M.launch_seeds(256)
    try:
        O.set_option("cos_short_tenth_ulp", 20)
        for box in ((20, 60, 103, 112), (20, 60, 132, 139)):
            y0, y1, x0, x1 = box
            dd = (_coffee_region(sc, seeds, box) - gold[y0:y1, x0:x1]).mean(axis=(0, 1))
            assert -0.02 < dd[1] < 0.01 and -0.02 < dd[2] < 0.01, (box, dd)
        y0, y1, x0, x1 = 10, 50, 20, 80
        assert np.abs((_coffee_region(sc, seeds[:128], (y0, y1, x0, x1)) - gold[y0:y1, x0:x1]).mean(axis=(0, 1))).max() < 2e-3
    finally:
        O.set_option("cos_short_tenth_ulp", 0)


def test_light_reflection_in_roughness_0001_plastic_depends_on_rounding():
    """Plastic_Orange has roughness 0.001: GTR2's 1 + (a^2 - 1) cos^2 with a^2 = 1e-6 is formed from a cosine known to
    6e-8, so the weight of a specular bounce depends on how each binary32 operation before it rounded.  Evaluating
    the same formulas in binary64 (oracle analysis switch) moves the reflection of the side light by more than any
    other region moves: part (about a quarter) of the gap to the PNG, which a -use_fast_math build cannot be
    expected to hit, is rounding and not a missing BRDF term."""
    gold = np.load(os.path.join(GOLD, "coffee_8x.npy"))
    sc = oracle_scene(M.HostScene("file:coffee", 240, 135))
    seeds = M.launch_seeds(512)
    band, wall = (20, 60, 101, 109), (10, 50, 20, 50)
    try:
        f32 = {k: _coffee_region(sc, seeds, b).mean(axis=(0, 1)) for k, b in (("band", band), ("wall", wall))}
        O.set_option("disney_binary64", 1)
        f64 = {k: _coffee_region(sc, seeds, b).mean(axis=(0, 1)) for k, b in (("band", band), ("wall", wall))}
    finally:
        O.set_option("disney_binary64", 0)
    assert f64["band"][2] - f32["band"][2] > 4e-3                 # blue = specular only (Cdlin.b = 7e-5)
    assert np.abs(f64["wall"] - f32["wall"]).max() < 1e-3         # a rough surface does not care
    y0, y1, x0, x1 = band
    assert f64["band"][2] < gold[y0:y1, x0:x1, 2].mean()          # ... and binary64 does not reach the PNG either


def test_oracle_bvh_equals_brute_force():
    hs = M.HostScene("file:coffee", 64, 36)
    d = hs.to_dict()
    keep = 3000                                                     # brute force over 3k faces is affordable
    d["vIdx"], d["nIdx"], d["faceMat"] = d["vIdx"][:keep], d["nIdx"][:keep], d["faceMat"][:keep]
    seeds = M.launch_seeds(2)
    a, sa = O.Scene(d, brute_force_tris=False).render(seeds)
    b, sb = O.Scene(d, brute_force_tris=True).render(seeds)
    assert np.array_equal(a, b) and sa.rays == sb.rays


def test_refract_and_offset_unit_cases():
    r = O.f3()
    assert L.orc_refract(r, O.f3(0, -1, 0), O.f3(0, 1, 0), 1.5) == 1 and np.allclose(list(r), [0, -1, 0])
    s = np.float32(np.sin(np.radians(60))); c = np.float32(np.cos(np.radians(60)))
    assert L.orc_refract(r, O.f3(s, c, 0), O.f3(0, 1, 0), 1.5) == 0 and list(r) == [0, 0, 0]   # TIR when leaving
    out = O.f3()
    L.orc_offset(O.f3(1.0, 1e-5, -2.0), O.f3(0, 1, 0), out)
    assert out[0] == 1.0 and out[2] == -2.0 and np.isclose(out[1], 1e-5 + 1e-4)
    L.orc_offset(O.f3(1.0, 1.0, 1.0), O.f3(1, 0, 0), out)
    assert out[0] == np.float32(1.0).view(np.int32).__add__(8192).astype(np.int32).view(np.float32)


@pytest.mark.parametrize("glass_below", [True, False])
def test_shadow_ray_is_decided_by_its_nearest_any_hit_surface(tmp_path, glass_below):
    """DESIGN.md 2, rule D5 (round 3): disneyAnyHit's GLASS branch accepts the hit (Material.cu:226-227 neither ignores it nor
    terminates the ray), which in OptiX ends the shadow ray's interval there.  Floor under a glass pane under an opaque pane
    under the light: the floor is lit through the glass, tinted by its colour; with the panes swapped it is dark.  The old
    order-independent rule (switch shadow_any_opaque_blocks) calls both dark.  The CPU build of the kernels' per-lane code
    (tests/hostsim) takes the same decisions as the oracle."""
    from common import write_glass_over_opaque_scene, hostsim_render
    hs = M.HostScene("file:cornell", 64, 48, base_folder=write_glass_over_opaque_scene(tmp_path, glass_below))
    seeds = M.launch_seeds(8)
    sc = oracle_scene(hs)
    img, st = sc.render(seeds)
    try:
        O.set_option("shadow_any_opaque_blocks", 1)
        old, _ = sc.render(seeds)
    finally:
        O.set_option("shadow_any_opaque_blocks", 0)
    gain = (img.astype(np.float64) - old).reshape(-1, 3).sum(axis=0) / len(seeds)       # light the new rule lets through, per channel
    if glass_below:
        lit = np.abs(img - old).sum(axis=2) > 0
        assert lit.mean() > 0.02                                                        # the floor strip of the frame
        assert gain[0] > 1.0 and gain[0] > 1.2 * gain[1] > 1.2 * 1.2 * gain[2] > 0       # tinted by the pane's colour (0.9, 0.6, 0.3)
        assert (img - old).min() >= 0.0
    else:
        assert np.array_equal(img, old)                                                 # opaque pane first: dark either way
    h, c = hostsim_render(hs, seeds)
    assert rmse(h / len(seeds), img / len(seeds)) <= 1e-6
    assert c["shadowRays"] == st.shadowRays and c["bounceRays"] == st.bounceRays
