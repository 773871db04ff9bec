"""Shared helpers for the test-suite (oracle + hostsim bindings, metrics)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import minimaloptix_amd as M                     # noqa: E402
from minimaloptix_amd import _capi as K          # noqa: E402
from oracle import oracle as O                   # noqa: E402

_HOSTSIM_DIR = os.path.join(REPO, "tests", "hostsim")
_hostsim = None


class HostsimScene(C.Structure):
    _fields_ = [("params", K.Params),
                ("nMaterials", C.c_int32), ("materials", C.POINTER(K.Material)),
                ("nSpheres", C.c_int32), ("spheres", C.POINTER(K.SphereParams)), ("sphereMat", C.POINTER(C.c_int32)),
                ("nQuads", C.c_int32), ("quads", C.POINTER(K.QuadParams)), ("quadMat", C.POINTER(C.c_int32)),
                ("nLights", C.c_int32), ("lights", C.POINTER(K.LightParams)),
                ("nFaces", C.c_int32), ("facePos", C.POINTER(C.c_float)), ("faceNrm", C.POINTER(C.c_float)),
                ("faceHasNrm", C.POINTER(C.c_int32)), ("faceMat", C.POINTER(C.c_int32)),
                ("faceUV", C.POINTER(C.c_float)), ("faceHasUV", C.POINTER(C.c_int32)),
                ("nTextures", C.c_int32), ("texSize", C.POINTER(C.c_int32)), ("texels", C.POINTER(C.POINTER(C.c_float)))]


class HostsimBvhOut(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("tris", C.c_void_p), ("triPrim", C.POINTER(C.c_int32)),
                ("nNodes", C.c_int32), ("rootRef", C.c_int32), ("depth", C.c_int32), ("nodes64", C.c_void_p)]


def hostsim_lib():
    global _hostsim
    if _hostsim is None:
        path = os.environ.get("HOSTSIM_LIB", os.path.join(_HOSTSIM_DIR, "libhostsim.so"))
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", _HOSTSIM_DIR, "-s"])
        L = C.CDLL(path)
        L.hostsim_render.argtypes = [C.POINTER(HostsimScene), C.c_int, C.POINTER(C.c_int32), C.c_int,
                                     C.POINTER(C.c_float), C.POINTER(C.c_uint64)]
        L.hostsim_render_timed.argtypes = [C.POINTER(HostsimScene), C.c_int, C.POINTER(C.c_int32), C.c_int,
                                           C.POINTER(C.c_float), C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
        L.hostsim_build_bvh.argtypes = [C.POINTER(HostsimScene), C.c_int, C.POINTER(HostsimBvhOut)]
        _hostsim = L
    return _hostsim


def _hostsim_scene(hs):
    """HostScene -> (HostsimScene, keepalive)"""
    f = hs.flat()
    fp, fn, has, fm = hs.face_arrays()
    fm = np.ascontiguousarray(fm, np.int32); has = np.ascontiguousarray(has, np.int32)
    smat = np.ascontiguousarray(f["sphereMat"], np.int32); qmat = np.ascontiguousarray(f["quadMat"], np.int32)
    s = HostsimScene()
    s.params = hs.params
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
    fpp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    s.nMaterials, s.materials = hs.sizes.nMaterials, C.cast(f["materials"], C.POINTER(K.Material))
    s.nSpheres, s.spheres, s.sphereMat = hs.sizes.nSpheres, C.cast(f["spheres"], C.POINTER(K.SphereParams)), ip(smat)
    s.nQuads, s.quads, s.quadMat = hs.sizes.nQuads, C.cast(f["quads"], C.POINTER(K.QuadParams)), ip(qmat)
    s.nLights, s.lights = hs.sizes.nLights, C.cast(f["lights"], C.POINTER(K.LightParams))
    s.nFaces, s.facePos, s.faceNrm, s.faceHasNrm, s.faceMat = len(fm), fpp(fp), fpp(fn), ip(has), ip(fm)
    uv, has_uv = hs.face_uvs()
    uv = np.ascontiguousarray(uv, np.float32); has_uv = np.ascontiguousarray(has_uv, np.int32)
    s.faceUV, s.faceHasUV = fpp(uv), ip(has_uv)
    tex = [np.ascontiguousarray(t, np.float32) for t in f["textures"]]
    tsz = np.ascontiguousarray([[t.shape[1], t.shape[0]] for t in tex], np.int32).reshape(-1, 2)
    tptr = (C.POINTER(C.c_float) * max(1, len(tex)))(*[fpp(t) for t in tex])
    s.nTextures, s.texSize, s.texels = len(tex), ip(tsz), tptr
    return s, (f, fp, fn, has, fm, smat, qmat, uv, has_uv, tex, tsz, tptr)


HOSTSIM_COUNTERS = ("samples", "primaryRays", "bounceRays", "shadowRays", "nodeFetches", "triTests", "closestHits",
                    "lightLoads", "analyticTests")


def hostsim_render(hs, seeds, leaf_size=4, accum=None, node_format=128):
    """node_format 64: the CPU build walks the 64-byte nodes (pt_types.h Node64), as the packet kernel does by default."""
    hostsim_lib().hostsim_set_node_format(int(node_format))
    s, keep = _hostsim_scene(hs)
    seeds = np.ascontiguousarray(np.asarray(seeds, np.int32))
    if accum is None:
        accum = np.zeros((hs.height, hs.width, 3), np.float32)
    cnt = (C.c_uint64 * 9)()
    timing = (C.c_double * 3)()
    rc = hostsim_lib().hostsim_render_timed(C.byref(s), leaf_size, seeds.ctypes.data_as(C.POINTER(C.c_int32)), len(seeds),
                                            accum.ctypes.data_as(C.POINTER(C.c_float)), cnt, timing)
    hostsim_lib().hostsim_set_node_format(128)
    assert rc == 0
    out = dict(zip(HOSTSIM_COUNTERS, [int(x) for x in cnt]))
    out.update(build_s=timing[0], render_s=timing[1], threads=int(timing[2]))
    return accum, out


def node64_boxes(n64):
    """Decodes [n, 16]-word Node64 records as the kernels do (plane = fma(q, step, corner), pt_lbvh.h node64_plane): returns
    (boxes [n, 6, 4] float32 in the order lox loy loz hix hiy hiz x child, refs [n, 4] int32, steps [n, 3] float32)."""
    n64 = np.ascontiguousarray(n64, np.uint32)
    corner = n64[:, 0:3].view(np.float32).astype(np.float64)
    step = n64[:, 3:6].view(np.float32).astype(np.float64)
    q = ((n64[:, 6:12, None] >> (8 * np.arange(4, dtype=np.uint32))) & 0xff).astype(np.float64)      # [n, 6, 4]
    ax = np.array([0, 1, 2, 0, 1, 2])
    boxes = (q * step[:, ax, None] + corner[:, ax, None]).astype(np.float32)      # exact product, one rounding: as the fma
    return boxes, n64[:, 12:16].view(np.int32), n64[:, 3:6].view(np.float32)


def hostsim_bvh(hs, leaf_size=4, builder=1, want_nodes64=False):
    """builder: 0 = Morton radix tree, 1 = binned SAH (the device default; moptix option "builder")."""
    hostsim_lib().hostsim_set_builder(int(builder))
    s, keep = _hostsim_scene(hs)
    nf = max(1, s.nFaces)
    nodes = np.zeros((nf, 32), np.uint32); tris = np.zeros((nf, 12), np.uint32); prim = np.zeros(nf, np.int32)
    out = HostsimBvhOut()
    out.nodes, out.tris, out.triPrim = nodes.ctypes.data, tris.ctypes.data, prim.ctypes.data_as(C.POINTER(C.c_int32))
    n64 = np.zeros((nf, 16), np.uint32)
    out.nodes64 = n64.ctypes.data if want_nodes64 else None
    rc = hostsim_lib().hostsim_build_bvh(C.byref(s), leaf_size, C.byref(out))
    hostsim_lib().hostsim_set_builder(1)
    assert rc == 0
    if want_nodes64:
        return nodes[:out.nNodes], tris[:s.nFaces], prim[:s.nFaces], out.rootRef, out.depth, n64[:out.nNodes]
    return nodes[:out.nNodes], tris[:s.nFaces], prim[:s.nFaces], out.rootRef, out.depth


def oracle_scene(hs, brute_force_tris=False):
    return O.Scene(hs.to_dict(), brute_force_tris=brute_force_tris)


def rmse(a, b):
    d = np.asarray(a, np.float64) - np.asarray(b, np.float64)
    return float(np.sqrt(np.mean(d * d)))


def have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


# ---------------------------------------------------------------------------------------------
# textured test scene: written to disk and loaded through the .scene/.obj/image ingest
# ---------------------------------------------------------------------------------------------
def write_png(path, rgb, filter_type=0):
    """Minimal PNG writer (8-bit RGB, zlib level 9 -> dynamic Huffman blocks); filter 0 (None) or 1 (Sub)."""
    import struct, zlib
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w, _ = rgb.shape
    rows = []
    for y in range(h):
        row = rgb[y].reshape(-1).astype(np.int32)
        if filter_type == 1:
            prev = np.concatenate([np.zeros(3, np.int32), row[:-3]])
            row = (row - prev) & 0xff
        rows.append(bytes([filter_type]) + row.astype(np.uint8).tobytes())
    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    data = (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(b"".join(rows), 9)) + chunk(b"IEND", b""))
    with open(path, "wb") as f:
        f.write(data)


def test_textures():
    """Two small deterministic images: a noisy checker (PNG) and a colour ramp (PPM)."""
    rng = np.random.RandomState(7)
    yy, xx = np.mgrid[0:24, 0:32]
    checker = np.where(((xx // 4 + yy // 4) % 2)[..., None] == 0, np.uint8([230, 60, 40]), np.uint8([40, 90, 220]))
    checker = np.clip(checker.astype(np.int32) + rng.randint(-20, 21, checker.shape), 0, 255).astype(np.uint8)
    ramp = np.zeros((16, 16, 3), np.uint8)
    ramp[..., 0] = (np.arange(16) * 17)[None, :]; ramp[..., 1] = (np.arange(16) * 17)[:, None]; ramp[..., 2] = 128
    return checker, ramp


test_textures.__test__ = False


def write_textured_scene(root):
    """root/cornell/{cornell.scene, *.obj, checker.png, ramp.ppm}: a box seen by the 'cornell' camera with
    textured Disney walls (repeat-wrapped and negative texcoords), a textured glass pane (BRDF_GLASS: tint from
    the texture, shadow any-hit from the constant colour), a textured material on a mesh WITHOUT texcoords
    (texcoord (0,0)), an untextured mesh and one quad light."""
    d = os.path.join(str(root), "cornell")
    os.makedirs(d, exist_ok=True)
    checker, ramp = test_textures()
    write_png(os.path.join(d, "checker.png"), checker, filter_type=1)
    with open(os.path.join(d, "ramp.ppm"), "wb") as f:
        f.write(b"P6\n# ramp\n16 16\n255\n" + ramp.tobytes())

    def quad_obj(name, p, uv=None, n=None):
        lines = ["v %g %g %g" % tuple(q) for q in p]
        if uv is not None:
            lines += ["vt %g %g" % tuple(t) for t in uv]
        if n is not None:
            lines += ["vn %g %g %g" % tuple(n)]
        def idx(i):
            if n is not None:
                return "%d/%s/1" % (i, i if uv is not None else "")
            return "%d/%d" % (i, i) if uv is not None else "%d" % i
        lines.append("f " + " ".join(idx(i) for i in (1, 2, 3, 4)))
        with open(os.path.join(d, name), "w") as f:
            f.write("\n".join(lines) + "\n")

    quad_obj("back.obj", [(-1, -1, 1), (-1, 1, 1), (1, 1, 1), (1, -1, 1)], uv=[(-0.7, -0.4), (-0.7, 1.9), (2.2, 1.9), (2.2, -0.4)], n=(0, 0, -1))
    quad_obj("floor.obj", [(-1, -1, -1), (-1, -1, 1), (1, -1, 1), (1, -1, -1)], uv=[(0, 0), (0, 1), (1, 1), (1, 0)], n=(0, 1, 0))
    quad_obj("left.obj", [(1, -1, -1), (1, -1, 1), (1, 1, 1), (1, 1, -1)])                       # no texcoords, textured material
    quad_obj("right.obj", [(-1, -1, -1), (-1, 1, -1), (-1, 1, 1), (-1, -1, 1)], n=(1, 0, 0))    # untextured
    quad_obj("pane.obj", [(-0.6, -1, 0.2), (-0.6, 0.3, 0.2), (0.5, 0.3, 0.0), (0.5, -1, 0.0)], uv=[(0, 0), (0, 1.5), (1.5, 1.5), (1.5, 0)])
    scene = """
material Back
{
    color 0.9 0.9 0.9
    albedoTex checker.png
    roughness 0.4
    specular 0.7
    specularTint 0.5
    sheen 0.3
}
material Floor
{
    color 0.5 0.5 0.5
    albedoTex ramp.ppm
    roughness 0.15
    metallic 0.6
    clearcoat 0.4
}
material Left
{
    color 0.2 0.8 0.2
    albedoTex checker.png
}
material Right
{
    color 0.8 0.3 0.2
    roughness 0.6
}
material Pane
{
    color 0.9 0.8 0.7
    albedoTex ramp.ppm
    brdf 1
}
mesh
{
    file back.obj
    material Back
}
mesh
{
    file floor.obj
    material Floor
}
mesh
{
    file left.obj
    material Left
}
mesh
{
    file right.obj
    material Right
}
mesh
{
    file pane.obj
    material Pane
}
light
{
    type Quad
    position -0.4 0.98 -0.4
    v1 0.4 0.98 -0.4
    v2 -0.4 0.98 0.4
    emission 12 12 12
}
"""
    with open(os.path.join(d, "cornell.scene"), "w") as f:
        f.write(scene)
    return str(root) + "/"


def textured_scene(root, width=96, height=72):
    base = write_textured_scene(root)
    return M.HostScene("file:cornell", width, height, base_folder=base)


def write_glass_over_opaque_scene(root, glass_below, scale=1.0):
    """A floor (y = 0), a quad light above it (y = 3) and two large horizontal panes between them (y = 1.5 and 2), one Disney GLASS (tinted) and
    one opaque Disney, the glass one nearer to the floor when `glass_below`: every shadow ray from the floor to the light meets
    both.  disneyAnyHit accepts the glass hit without terminating the ray, so in OptiX the ray's interval ends there and the
    opaque pane behind it never blocks (rule D5): floor lit (tinted) with the glass below, dark with it above."""
    d = os.path.join(str(root), "cornell")
    os.makedirs(d, exist_ok=True)

    def quad(name, y, hx, hz):
        with open(os.path.join(d, name), "w") as f:
            f.write("v %g %g %g\nv %g %g %g\nv %g %g %g\nv %g %g %g\nvn 0 1 0\nf 1//1 3//1 2//1\nf 1//1 4//1 3//1\n" % (
                -hx * scale, y * scale, -hz * scale, hx * scale, y * scale, -hz * scale, hx * scale, y * scale, hz * scale,
                -hx * scale, y * scale, hz * scale))      # wound so that the geometric normal is +y
    # scene box 0..2 in y: the cornell camera (MinimalOptiX.cpp:323-335) sits at its centre height 1, below both panes, and sees the floor
    quad("floor.obj", 0.0, 3.0, 1.5)
    quad("glass.obj", 1.5 if glass_below else 2.0, 3.0, 1.5)
    quad("opaque.obj", 2.0 if glass_below else 1.5, 3.0, 1.5)
    scene = """material Floor
{
    color 0.8 0.8 0.8
    roughness 0.9
}
material Pane
{
    color 0.9 0.6 0.3
    brdf 1
}
material Lid
{
    color 0.2 0.2 0.2
    roughness 0.6
}
mesh
{
    file floor.obj
    material Floor
}
mesh
{
    file glass.obj
    material Pane
}
mesh
{
    file opaque.obj
    material Lid
}
light
{
    type Quad
    position %g %g %g
    v1 %g %g %g
    v2 %g %g %g
    emission 20 20 20
}
""" % tuple(scale * c for c in (-1, 3, -0.5, 1, 3, -0.5, -1, 3, 0.5))
    with open(os.path.join(d, "cornell.scene"), "w") as f:
        f.write(scene)
    return str(root) + "/"


def tree_containment_errors(nodes, tris, root, nodes64=None):
    """is this tree VALID?  Every child box of every node (the 128-byte form, and the decoded 64-byte form when given) must
    contain the bounds of all triangles below that child; a builder bug -- a child box that fails to enclose its triangles under some
    leaf size / builder / Node64 rounding -- would make every kernel miss the same triangle, and a comparison of kernels on that tree
    could not see it.  nodes: [n, 32] words (pt_types.h Node128), tris: [m, 12] words (Tri48: p0 mat e0 prim e1 shadow), root: node
    index or leaf reference.  Returns the number of (node, child) pairs whose box does not contain its triangles."""
    nodes = np.ascontiguousarray(nodes, np.uint32); tris = np.ascontiguousarray(tris, np.uint32)
    f = tris.view(np.float32)
    p0 = f[:, 0:3].astype(np.float64); p1 = p0 + f[:, 4:7]; p2 = p0 - f[:, 8:11]       # e0 = p1 - p0, e1 = p0 - p2 (rounded once: compare with slack)
    tlo = np.minimum(np.minimum(p0, p1), p2); thi = np.maximum(np.maximum(p0, p1), p2)
    nf = nodes.view(np.float32)
    box128 = nf[:, 0:24].reshape(-1, 6, 4)                                              # lox loy loz hix hiy hiz x child
    refs = nodes[:, 24:28].view(np.int32); count = nodes[:, 28].view(np.int32)
    box64 = node64_boxes(nodes64)[0] if nodes64 is not None else None
    bad = 0
    if root < 0:
        return 0
    # post-order over the tree: exact bounds of every subtree
    lo = {}; hi = {}
    stack = [(int(root), False)]
    while stack:
        n, done = stack.pop()
        if not done:
            stack.append((n, True))
            for k in range(int(count[n])):
                r = int(refs[n, k])
                if r >= 0:
                    stack.append((r, False))
            continue
        nlo = np.full(3, np.inf); nhi = np.full(3, -np.inf)
        for k in range(int(count[n])):
            r = int(refs[n, k])
            if r >= 0:
                clo, chi = lo[r], hi[r]
            else:
                first, cnt = (~r) >> 3, ((~r) & 7) + 1
                clo, chi = tlo[first:first + cnt].min(axis=0), thi[first:first + cnt].max(axis=0)
            eps = 1e-6 * np.maximum(1.0, np.abs(chi) + np.abs(clo))                     # p1 / p2 are reconstructed from rounded edges
            for boxes in (box128, box64):
                if boxes is None:
                    continue
                b = boxes[n, :, k].astype(np.float64)
                if (b[0:3] > clo + eps).any() or (b[3:6] < chi - eps).any():
                    bad += 1
            nlo = np.minimum(nlo, clo); nhi = np.maximum(nhi, chi)
        lo[n], hi[n] = nlo, nhi
    return bad
