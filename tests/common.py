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
                ("faceHasNrm", C.POINTER(C.c_int32)), ("faceMat", C.POINTER(C.c_int32))]


class HostsimBvhOut(C.Structure):
    _fields_ = [("nodes", C.c_void_p), ("tris", C.c_void_p), ("triPrim", C.POINTER(C.c_int32)),
                ("nNodes", C.c_int32), ("rootRef", C.c_int32), ("depth", C.c_int32)]


def hostsim_lib():
    global _hostsim
    if _hostsim is None:
        path = os.path.join(_HOSTSIM_DIR, "libhostsim.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", _HOSTSIM_DIR, "-s"])
        L = C.CDLL(path)
        L.hostsim_render.argtypes = [C.POINTER(HostsimScene), C.c_int, C.POINTER(C.c_int32), C.c_int,
                                     C.POINTER(C.c_float), C.POINTER(C.c_uint64)]
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
    return s, (f, fp, fn, has, fm, smat, qmat)


HOSTSIM_COUNTERS = ("samples", "primaryRays", "bounceRays", "shadowRays", "nodeFetches", "triTests", "closestHits",
                    "lightLoads", "analyticTests")


def hostsim_render(hs, seeds, leaf_size=4, accum=None):
    s, keep = _hostsim_scene(hs)
    seeds = np.ascontiguousarray(np.asarray(seeds, np.int32))
    if accum is None:
        accum = np.zeros((hs.height, hs.width, 3), np.float32)
    cnt = (C.c_uint64 * 9)()
    rc = hostsim_lib().hostsim_render(C.byref(s), leaf_size, seeds.ctypes.data_as(C.POINTER(C.c_int32)), len(seeds),
                                      accum.ctypes.data_as(C.POINTER(C.c_float)), cnt)
    assert rc == 0
    return accum, dict(zip(HOSTSIM_COUNTERS, [int(x) for x in cnt]))


def hostsim_bvh(hs, leaf_size=4):
    s, keep = _hostsim_scene(hs)
    nf = max(1, s.nFaces)
    nodes = np.zeros((nf, 16), np.uint32); tris = np.zeros((nf, 12), np.uint32); prim = np.zeros(nf, np.int32)
    out = HostsimBvhOut()
    out.nodes, out.tris, out.triPrim = nodes.ctypes.data, tris.ctypes.data, prim.ctypes.data_as(C.POINTER(C.c_int32))
    rc = hostsim_lib().hostsim_build_bvh(C.byref(s), leaf_size, C.byref(out))
    assert rc == 0
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
