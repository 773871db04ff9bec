"""Host-side ingest kept from the reference: .obj loader, .scene parser, scene builders."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from common import M

K = M._capi
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
KAT = json.load(open(os.path.join(GOLD, "kat.json")))


def _obj_stats(path):
    nv, nn, nt, ns = (C.c_int32() for _ in range(4))
    faces = K.host_lib().mohost_obj_stats(path.encode(), C.byref(nv), C.byref(nn), C.byref(nt), C.byref(ns))
    return faces, nv.value, nn.value, nt.value, ns.value


def test_obj_loader_matches_reference_mesh_inventory():
    """SURVEY A3 inventory, measured there with the reference's vendored tiny_obj_loader."""
    total_f = total_v = 0
    for name, (verts, faces) in KAT["coffee_meshes"].items():
        f, nv, nn, nt, ns = _obj_stats(os.path.join(M.scenes_dir(), "coffee", name + ".obj"))
        assert (f, nv, ns) == (faces, verts, 1) and nn == verts and nt == verts, name
        total_f += f; total_v += nv
    assert (total_f, total_v) == (168193, 101812)


def test_obj_loader_syntax(tmp_path):
    p = tmp_path / "t.obj"
    p.write_text("# comment\nmtllib x.mtl\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvn 0 0 1\nvt 0.5 0.5\n"
                 "g quad\nf 1/1/1 2/1/1 3/1/1 4/1/1\n"          # quad -> fan of 2 triangles
                 "o tri\nusemtl m\nf -4//1 -3//1 -2//1\n"       # negative indices, v//vn form
                 "f 1 2 3\n")                                   # bare vertex form, same shape
    f, nv, nn, nt, ns = _obj_stats(str(p))
    assert (f, nv, nn, nt, ns) == (4, 4, 1, 1, 2)
    bad = tmp_path / "bad.obj"
    bad.write_text("v 0 0 0\nf 1 2 3\n")                         # indices out of range are kept as-is by tinyobj too
    assert _obj_stats(str(tmp_path / "missing.obj"))[0] == -1
    assert "Cannot open file" in K.host_lib().mohost_last_error().decode()


def test_coffee_scene_file():
    hs = M.HostScene("file:coffee", 1920, 1080)
    s = hs.sizes
    assert (s.nFaces, s.nVerts, s.nMeshes, s.nLights, s.nQuads, s.nSpheres) == (168193, 101812, 19, 3, 3, 0)
    assert hs.accel == "Trbvh" and len(hs.warnings) == 1 and "Mesh010.obj" in hs.warnings[0]
    assert np.allclose(hs.aabb_min, [-1, 0, -1.094168], atol=1e-6) and np.allclose(hs.aabb_max, [1, 0.811135, 1], atol=1e-6)
    d = hs.to_dict()
    k = KAT["cam_coffee"]
    for f in ("origin", "horizontal", "vertical", "scrLowerLeftCorner", "u", "v"):
        assert np.allclose(d["cam"][f], k[f], atol=2e-6), f
    assert d["bgColor"] == [0, 0, 0] and d["rayMaxDepth"] == 256 and abs(d["rayEpsilonT"] - 1e-3) < 1e-9
    mats = d["materials"]
    floor = [m for m in mats if m["kind"] == K.MAT_DISNEY and np.allclose(m["color"], [0.578] * 3)]
    assert floor and abs(floor[0]["roughness"] - 0.01) < 1e-7 and floor[0]["specular"] == 0.5 and floor[0]["clearcoatGloss"] == 1.0
    assert sum(1 for m in mats if m["kind"] == K.MAT_LIGHT) == 3
    assert all(np.allclose(l["emission"], [4, 4, 4]) and l["shape"] == K.LIGHT_QUAD for l in d["lights"])
    # scene.cpp:78-83: u=v1-pos, v=v2-pos, area=|u x v|, normal=normalize(u x v)
    l0 = d["lights"][0]
    u, v = np.float32(l0["u"]), np.float32(l0["v"])
    assert np.isclose(l0["area"], np.linalg.norm(np.cross(u, v)), rtol=1e-5)
    # light quad geometry uses setQuadParams(position,u,v): its plane normal is MINUS light.normal
    assert np.allclose(d["quads"][0][:3], -np.float32(l0["normal"]), atol=1e-6)


def test_missing_mesh_is_an_error_when_strict():
    with pytest.raises(M.MoptixError):
        M.HostScene("file:coffee", 64, 36, skip_missing=False)
    with pytest.raises(M.MoptixError):
        M.HostScene("file:nonexistent", 64, 36)


def test_scene_parser_quirks(tmp_path):
    """scene.cpp:5-124: '#' comments, per-line sscanf of every key, brdf enum, quad light derivation,
    width/height parsed but unused, unknown material name reported and skipped."""
    d = tmp_path / "scenes" / "mini"
    d.mkdir(parents=True)
    (d / "a.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//1\n")
    (d / "mini.scene").write_text(
        "# a comment\nproperties\n{\n\twidth 800\n\theight 1000\n}\n"
        "material Shiny\n{\n\tcolor 0.25 0.5 0.75\n\troughness 0.125\n\tmetallic 1.0\n\tbrdf 1\n\tclearcoat 0.5\n}\n"
        "mesh\n{\n\tfile a.obj\n\tmaterial Shiny\n}\n"
        "light\n{\n\ttype Quad\n\tposition 0 2 0\n\tv1 1 2 0\n\tv2 0 2 1\n\temission 3 2 1\n}")
    # "cornell" selects the camera/background of MinimalOptiX.cpp:323-335; the folder holds our file
    os.rename(d, tmp_path / "scenes" / "cornell")
    os.rename(tmp_path / "scenes" / "cornell" / "mini.scene", tmp_path / "scenes" / "cornell" / "cornell.scene")
    hs = M.HostScene("file:cornell", 128, 72, base_folder=str(tmp_path / "scenes") + "/")
    dd = hs.to_dict()
    m = dd["materials"][0]
    assert m["kind"] == K.MAT_DISNEY and np.allclose(m["color"], [0.25, 0.5, 0.75]) and m["brdfType"] == K.BRDF_GLASS
    assert m["roughness"] == 0.125 and m["metallic"] == 1.0 and m["clearcoat"] == 0.5 and m["sheenTint"] == 0.5
    l = dd["lights"][0]
    assert np.allclose(l["u"], [1, 0, 0]) and np.allclose(l["v"], [0, 0, 1]) and np.isclose(l["area"], 1.0)
    assert np.allclose(l["normal"], [0, -1, 0]) and np.allclose(l["emission"], [3, 2, 1])
    assert dd["bgColor"] == [0.5, 0.5, 0.5] and (dd["width"], dd["height"]) == (128, 72)   # .scene width/height ignored
    assert hs.sizes.nFaces == 1


def test_builtin_scenes():
    s = M.HostScene("spheres", 1920, 1080, farg=0.5)
    assert (s.sizes.nSpheres, s.sizes.nQuads, s.accel) == (3, 2, "NoAccel")
    kinds = [m["kind"] for m in s.to_dict()["materials"]]
    assert kinds == [K.MAT_LAMBERTIAN, K.MAT_METAL, K.MAT_GLASS, K.MAT_LAMBERTIAN, K.MAT_LIGHT]
    r = M.HostScene("random_spheres", 1280, 720, iarg=497)
    assert (r.sizes.nSpheres, r.sizes.nQuads) == (500, 33)                    # 3 + 497 spheres, floor + 32 light quads
    d = r.to_dict()
    sp = d["spheres"]
    assert np.allclose(sp[:3, 3], 3.0) and (sp[3:, 3] >= 0.0099).all() and (sp[3:, 3] <= 0.8 + 1e-6).all()
    assert np.allclose(sp[3:, 1], np.sqrt(sp[3:, 0] ** 2 + sp[3:, 2] ** 2), atol=1e-5)   # y = |(x,z)| (MinimalOptiX.cpp:643-645)
    r2 = M.HostScene("random_spheres", 1280, 720, iarg=497)
    assert np.array_equal(r2.to_dict()["spheres"], sp)                                  # deterministic layout (mt19937(42))
    c = M.HostScene("cornell_quads", 256, 256)
    assert (c.sizes.nQuads, c.sizes.nSpheres, c.sizes.nFaces) == (16, 0, 0)


def test_animation_matches_oracle_restatement():
    """MinimalOptiX::move/animate (MinimalOptiX.cpp:562-592): the host's physics step against the
    oracle's plain-C restatement, bit for bit over 300 frames (free fall, bounces, coming to rest)."""
    from common import O
    hs = M.HostScene("random_spheres", 320, 180, iarg=40)
    f = hs.flat()
    n = hs.sizes.nSpheres
    sph = (K.SphereParams * n)()
    for i in range(n):
        sph[i] = f["spheres"][i]
    ref = [([sph[i].center.x, sph[i].center.y, sph[i].center.z], sph[i].radius, [0.0, 0.0, 0.0]) for i in range(n)]
    angle = C.c_float(0.0)
    L = O.lib()
    moved = 0
    for frame in range(300):
        K.host_lib().mohost_animate_spheres(sph, n, 0.002, C.byref(angle))
        for i in range(n):
            c, r, v = O.f3(*ref[i][0]), ref[i][1], O.f3(*ref[i][2])
            L.orc_move_sphere(c, r, v, 0.002)
            ref[i] = (list(c), r, list(v))
            assert [sph[i].center.x, sph[i].center.y, sph[i].center.z] == list(c), (frame, i)
            assert [sph[i].velocity.x, sph[i].velocity.y, sph[i].velocity.z] == list(v), (frame, i)
        moved += 1
    assert abs(angle.value - 300 * 0.002 * 5) < 1e-4
    # everything has fallen onto the plane y = -0.5 or is still bouncing above it
    assert all(sph[i].center.y >= -0.5 + sph[i].radius - 1e-4 for i in range(n))
    # the three r=3 spheres start below the plane: divergence D7 snaps them onto it at rest
    assert all(sph[i].center.y == 2.5 and sph[i].velocity.y == 0.0 for i in range(3))
    assert any(sph[i].velocity.y < 0 for i in range(3, n))            # somebody is on the way up after a bounce
    cam = K.CamParams()
    K.host_lib().mohost_video_camera(angle.value, 16 / 9, C.byref(cam))
    assert abs(cam.origin.y - min(12.0, angle.value / 10 + 8.0)) < 1e-5 and abs(cam.lensRadius - 0.1) < 1e-7


def test_sphere_light_scene_matches_oracle(tmp_path):
    """`.scene` sphere lights (`radius`, `normal`), `name` lines and `properties`, rendered by
    the device code path compiled for the host and by the oracle (NEE on a sphere light uses the rejection sampler)."""
    from common import O, hostsim_render, oracle_scene, rmse
    d = tmp_path / "hyperion"
    d.mkdir()
    (d / "floor.obj").write_text("v -2 0 -2\nv -2 0 2\nv 2 0 2\nv 2 0 -2\nvn 0 1 0\nf 1//1 2//1 3//1 4//1\n")
    (d / "blade.obj").write_text("v -0.5 0.1 -0.3\nv 0.6 0.1 -0.2\nv 0.0 0.9 0.4\nf 1 2 3\n")
    (d / "hyperion.scene").write_text(
        "properties\n{\n\twidth 640\n\theight 480\n}\n"
        "material Ground\n{\n\tname Ground\n\tcolor 0.7 0.6 0.5\n\troughness 0.3\n}\n"
        "material Blade\n{\n\tcolor 0.2 0.4 0.9\n\tmetallic 0.8\n\troughness 0.2\n}\n"
        "mesh\n{\n\tfile floor.obj\n\tmaterial Ground\n}\n"
        "mesh\n{\n\tfile blade.obj\n\tmaterial Blade\n}\n"
        "light\n{\n\ttype Sphere\n\tposition 0.3 1.6 -0.4\n\tradius 0.25\n\tnormal 0 -2 0\n\temission 20 18 16\n}\n")
    hs = M.HostScene("file:hyperion", 72, 54, base_folder=str(tmp_path) + "/")
    dd = hs.to_dict()
    l = dd["lights"][0]
    assert l["shape"] == K.LIGHT_SPHERE and np.isclose(l["area"], 4 * np.pi * 0.25 ** 2, rtol=1e-6)
    assert np.allclose(l["normal"], [0, -1, 0])                              # scene.cpp:85 normalises it
    assert hs.sizes.nSpheres == 1 and np.isclose(dd["spheres"][0][3], 0.25)  # the light's geometry (MinimalOptiX.cpp:497-503)
    assert (dd["width"], dd["height"]) == (72, 54)
    seeds = M.launch_seeds(6, 3)
    ref, st = oracle_scene(hs).render(seeds)
    got, cnt = hostsim_render(hs, seeds)
    assert (cnt["primaryRays"], cnt["bounceRays"], cnt["shadowRays"]) == (st.primaryRays, st.bounceRays, st.shadowRays)
    assert st.shadowRays > 0 and rmse(got, ref) / len(seeds) < 1e-5
    assert ref.max() > 0.5                                                   # the light reaches the floor


def test_face_index_past_the_last_vertex_is_an_error(tmp_path):
    """A face that references a vertex the file never defines: the reference reads attrib.vertices out of bounds
    (MinimalOptiX.cpp:430-433); here scene ingest reports it."""
    d = tmp_path / "cornell"
    d.mkdir()
    (d / "a.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 7\n")
    (d / "cornell.scene").write_text("material M\n{\n\tcolor 1 1 1\n}\nmesh\n{\n\tfile a.obj\n\tmaterial M\n}\n")
    with pytest.raises(M.MoptixError) as e:
        M.HostScene("file:cornell", 32, 32, base_folder=str(tmp_path) + "/")
    assert "references vertex 7 of 3" in str(e.value)
