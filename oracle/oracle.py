"""ctypes binding for the CPU oracle (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY.  Importable from tests/, bench.py's cpu_baseline leg
and __graft_entry__.smoke(); never from minimaloptix_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

ORC_LAMBERTIAN, ORC_METAL, ORC_GLASS, ORC_DISNEY, ORC_LIGHT = range(5)
BRDF_NORMAL, BRDF_GLASS = 0, 1
LIGHT_SPHERE, LIGHT_QUAD = 0, 1

f3 = C.c_float * 3
f4 = C.c_float * 4


class OrcMaterial(C.Structure):
    _fields_ = [("kind", C.c_int32), ("albedo", f3), ("fuzz", C.c_float), ("refIdx", C.c_float),
                ("emission", f3), ("color", f3),
                ("metallic", C.c_float), ("subsurface", C.c_float), ("specular", C.c_float),
                ("roughness", C.c_float), ("specularTint", C.c_float), ("anisotropic", C.c_float),
                ("sheen", C.c_float), ("sheenTint", C.c_float), ("clearcoat", C.c_float),
                ("clearcoatGloss", C.c_float), ("brdfType", C.c_int32), ("albedoID", C.c_int32)]


class OrcSphere(C.Structure):
    _fields_ = [("center", f3), ("radius", C.c_float), ("mat", C.c_int32)]


class OrcQuad(C.Structure):
    _fields_ = [("plane", f4), ("v1", f3), ("v2", f3), ("anchor", f3), ("mat", C.c_int32)]


class OrcLight(C.Structure):
    _fields_ = [("position", f3), ("normal", f3), ("emission", f3), ("u", f3), ("v", f3),
                ("area", C.c_float), ("radius", C.c_float), ("shape", C.c_int32)]


class OrcCam(C.Structure):
    _fields_ = [("origin", f3), ("horizontal", f3), ("vertical", f3), ("scrLowerLeftCorner", f3),
                ("u", f3), ("v", f3), ("lensRadius", C.c_float)]


class OrcTexture(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("rgba", C.POINTER(C.c_float))]


class OrcScene(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("cam", OrcCam), ("bgColor", f3),
                ("rayMaxDepth", C.c_int32), ("rayMinIntensity", C.c_float), ("rayEpsilonT", C.c_float),
                ("nMaterials", C.c_int32), ("materials", C.POINTER(OrcMaterial)),
                ("nSpheres", C.c_int32), ("spheres", C.POINTER(OrcSphere)),
                ("nQuads", C.c_int32), ("quads", C.POINTER(OrcQuad)),
                ("nLights", C.c_int32), ("lights", C.POINTER(OrcLight)),
                ("nVerts", C.c_int32), ("positions", C.POINTER(C.c_float)),
                ("nNorms", C.c_int32), ("normals", C.POINTER(C.c_float)),
                ("nUVs", C.c_int32), ("texcoords", C.POINTER(C.c_float)),
                ("nFaces", C.c_int32),
                ("vIdx", C.POINTER(C.c_int32)), ("nIdx", C.POINTER(C.c_int32)),
                ("tIdx", C.POINTER(C.c_int32)), ("faceMat", C.POINTER(C.c_int32)),
                ("bruteForceTris", C.c_int32), ("nTextures", C.c_int32), ("textures", C.POINTER(OrcTexture))]


class OrcStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("primaryRays", "bounceRays", "shadowRays", "samples", "closestHits", "misses", "depthCapped")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}

    @property
    def rays(self):
        return int(self.primaryRays + self.bounceRays + self.shadowRays)


_lib = None


def build(force=False):
    """Compile oracle/liboracle.so with the committed Makefile."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("pt_oracle.c", "pt_oracle.h")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.orc_render.restype = C.c_int
        L.orc_render.argtypes = [C.POINTER(OrcScene), C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_float),
                                 C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(OrcStats)]
        L.orc_tea16.restype = C.c_uint32
        L.orc_tea16.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_lcg.restype = C.c_uint32
        L.orc_lcg.argtypes = [C.POINTER(C.c_int32)]
        L.orc_rand.restype = C.c_float
        L.orc_rand.argtypes = [C.POINTER(C.c_int32)]
        L.orc_launch_seed.restype = C.c_int32
        L.orc_launch_seed.argtypes = [C.c_uint32, C.c_uint32]
        L.orc_set_cam_params.restype = None
        L.orc_set_cam_params.argtypes = [f3, f3, f3, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(OrcCam)]
        L.orc_set_quad_params.restype = None
        L.orc_set_quad_params.argtypes = [f3, f3, f3, C.POINTER(OrcQuad)]
        L.orc_init_disney.restype = None
        L.orc_init_disney.argtypes = [C.POINTER(OrcMaterial)]
        L.orc_refract.restype = C.c_int
        L.orc_refract.argtypes = [f3, f3, f3, C.c_float]
        L.orc_offset.restype = None
        L.orc_offset.argtypes = [f3, f3, f3]
        L.orc_disney_pdf.restype = C.c_float
        L.orc_disney_pdf.argtypes = [C.POINTER(OrcMaterial), f3, f3, f3, f3]
        L.orc_disney_eval.restype = None
        L.orc_disney_eval.argtypes = [C.POINTER(OrcMaterial), f3, f3, f3, f3, f3, f3]
        L.orc_disney_sample.restype = None
        L.orc_disney_sample.argtypes = [C.POINTER(C.c_int32), C.POINTER(OrcMaterial), f3, f3, f3, f3]
        L.orc_intersect_triangle.restype = C.c_int
        L.orc_intersect_triangle.argtypes = [f3, f3, C.c_float, C.c_float, f3, f3, f3, f3,
                                             C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_trace_one.restype = None
        L.orc_trace_one.argtypes = [C.POINTER(OrcScene), f3, f3, C.c_int32, f3]
        L.orc_closest_hit.restype = C.c_int
        L.orc_closest_hit.argtypes = [C.POINTER(OrcScene), f3, f3, C.c_float, C.c_float, C.POINTER(C.c_float)]
        L.orc_closest_hit_batch.restype = C.c_int
        L.orc_closest_hit_batch.argtypes = [C.POINTER(OrcScene), C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_float)]
        L.orc_set_option.restype = C.c_int
        L.orc_set_option.argtypes = [C.c_char_p, C.c_int]
        L.orc_move_sphere.restype = None
        L.orc_move_sphere.argtypes = [f3, C.c_float, f3, C.c_float]
        L.orc_tex2d.argtypes = [C.POINTER(OrcTexture), C.c_float, C.c_float, C.c_float * 4]
        L.orc_tex2d.restype = None
        L.orc_num_threads.restype = C.c_int
        L.orc_render_clamp_stats.restype = C.c_int
        L.orc_render_clamp_stats.argtypes = [C.POINTER(OrcScene), C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_render_by_depth.restype = C.c_int
        L.orc_render_by_depth.argtypes = [C.POINTER(OrcScene), C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        _lib = L
    return _lib


def set_option(name, value):
    """Analysis switches of the oracle ("disney_binary64": evaluate the Disney BRDF in binary64)."""
    if lib().orc_set_option(name.encode(), int(value)) != 0:
        raise ValueError(name)


def launch_seeds(n, base_seed=0, first=0):
    """Seed schedule of SURVEY 8(d): launchSeed(i) = (int)tea<16>(i, baseSeed)."""
    L = lib()
    return np.array([L.orc_launch_seed(first + i, base_seed) for i in range(n)], dtype=np.int32)


def _ptr(arr, ctype):
    return arr.ctypes.data_as(C.POINTER(ctype)) if arr is not None and arr.size else C.POINTER(ctype)()


class Scene:
    """Owns the numpy arrays behind an OrcScene.

    `desc` is the plain dict produced by minimaloptix_amd.host.SceneDesc.to_dict()
    (or hand-written in a test): width,height,cam(dict of 3-vectors+lensRadius),
    bgColor, rayMaxDepth, rayMinIntensity, rayEpsilonT, materials (list of dicts),
    spheres (n,4)+sphereMat, quads (n,13)+quadMat, lights (list of dicts),
    positions (nv,3), normals, texcoords, vIdx,nIdx,tIdx (nf,3), faceMat (nf,).
    """

    def __init__(self, desc, brute_force_tris=False):
        s = OrcScene()
        s.width, s.height = int(desc["width"]), int(desc["height"])
        cam = desc["cam"]
        for k in ("origin", "horizontal", "vertical", "scrLowerLeftCorner", "u", "v"):
            setattr(s.cam, k, f3(*[float(x) for x in cam[k]]))
        s.cam.lensRadius = float(cam["lensRadius"])
        s.bgColor = f3(*[float(x) for x in desc["bgColor"]])
        s.rayMaxDepth = int(desc.get("rayMaxDepth", 256))
        s.rayMinIntensity = float(desc.get("rayMinIntensity", 1e-3))
        s.rayEpsilonT = float(desc.get("rayEpsilonT", 1e-3))

        mats = desc["materials"]
        self._mats = (OrcMaterial * max(1, len(mats)))()
        for i, m in enumerate(mats):
            om = self._mats[i]
            om.kind = int(m["kind"])
            om.albedo = f3(*m.get("albedo", (0, 0, 0)))
            om.fuzz = float(m.get("fuzz", 0.0))
            om.refIdx = float(m.get("refIdx", 1.0))
            om.emission = f3(*m.get("emission", (0, 0, 0)))
            om.color = f3(*m.get("color", (1, 1, 1)))
            for k, dv in (("metallic", 0.0), ("subsurface", 0.0), ("specular", 0.5), ("roughness", 0.5),
                          ("specularTint", 0.0), ("anisotropic", 0.0), ("sheen", 0.0), ("sheenTint", 0.5),
                          ("clearcoat", 0.0), ("clearcoatGloss", 1.0)):
                setattr(om, k, float(m.get(k, dv)))
            om.brdfType = int(m.get("brdfType", 0))
            om.albedoID = int(m.get("albedoID", 0))
        s.nMaterials, s.materials = len(mats), C.cast(self._mats, C.POINTER(OrcMaterial))

        sph = np.asarray(desc.get("spheres", np.zeros((0, 4))), dtype=np.float32).reshape(-1, 4)
        smat = np.asarray(desc.get("sphereMat", []), dtype=np.int32)
        self._spheres = (OrcSphere * max(1, len(sph)))()
        for i in range(len(sph)):
            self._spheres[i].center = f3(*sph[i, :3]); self._spheres[i].radius = float(sph[i, 3]); self._spheres[i].mat = int(smat[i])
        s.nSpheres, s.spheres = len(sph), C.cast(self._spheres, C.POINTER(OrcSphere))

        q = np.asarray(desc.get("quads", np.zeros((0, 13))), dtype=np.float32).reshape(-1, 13)
        qmat = np.asarray(desc.get("quadMat", []), dtype=np.int32)
        self._quads = (OrcQuad * max(1, len(q)))()
        for i in range(len(q)):
            self._quads[i].plane = f4(*q[i, 0:4]); self._quads[i].v1 = f3(*q[i, 4:7])
            self._quads[i].v2 = f3(*q[i, 7:10]); self._quads[i].anchor = f3(*q[i, 10:13]); self._quads[i].mat = int(qmat[i])
        s.nQuads, s.quads = len(q), C.cast(self._quads, C.POINTER(OrcQuad))

        lights = desc.get("lights", [])
        self._lights = (OrcLight * max(1, len(lights)))()
        for i, l in enumerate(lights):
            ol = self._lights[i]
            for k in ("position", "normal", "emission", "u", "v"):
                setattr(ol, k, f3(*[float(x) for x in l.get(k, (0, 0, 0))]))
            ol.area = float(l.get("area", 0.0)); ol.radius = float(l.get("radius", 0.0)); ol.shape = int(l["shape"])
        s.nLights, s.lights = len(lights), C.cast(self._lights, C.POINTER(OrcLight))

        def arr(name, dt, cols):
            a = desc.get(name)
            if a is None:
                return np.zeros((0, cols), dtype=dt)
            return np.ascontiguousarray(np.asarray(a, dtype=dt).reshape(-1, cols))
        self._pos = arr("positions", np.float32, 3); self._nrm = arr("normals", np.float32, 3)
        self._uv = arr("texcoords", np.float32, 2)
        self._vi = arr("vIdx", np.int32, 3); self._ni = arr("nIdx", np.int32, 3); self._ti = arr("tIdx", np.int32, 3)
        nf = len(self._vi)
        if len(self._ni) != nf:
            self._ni = np.full((nf, 3), -1, dtype=np.int32)
        if len(self._ti) != nf:
            self._ti = np.full((nf, 3), -1, dtype=np.int32)
        self._fm = np.ascontiguousarray(np.asarray(desc.get("faceMat", np.zeros(nf)), dtype=np.int32).reshape(-1))
        s.nVerts, s.positions = len(self._pos), _ptr(self._pos, C.c_float)
        s.nNorms, s.normals = len(self._nrm), _ptr(self._nrm, C.c_float)
        s.nUVs, s.texcoords = len(self._uv), _ptr(self._uv, C.c_float)
        s.nFaces = nf
        s.vIdx, s.nIdx, s.tIdx, s.faceMat = (_ptr(self._vi, C.c_int32), _ptr(self._ni, C.c_int32),
                                            _ptr(self._ti, C.c_int32), _ptr(self._fm, C.c_int32))
        s.bruteForceTris = 1 if brute_force_tris else 0
        # textures: list of (H, W, 4) float32 arrays, row 0 = v 0; material albedoID = index + 1
        self._texpx = [np.ascontiguousarray(np.asarray(t, np.float32)) for t in desc.get("textures", [])]
        self._textures = (OrcTexture * max(1, len(self._texpx)))()
        for i, t in enumerate(self._texpx):
            assert t.ndim == 3 and t.shape[2] == 4
            self._textures[i].height, self._textures[i].width = t.shape[0], t.shape[1]
            self._textures[i].rgba = _ptr(t, C.c_float)
        s.nTextures, s.textures = len(self._texpx), C.cast(self._textures, C.POINTER(OrcTexture))
        self.c = s
        self.width, self.height = s.width, s.height

    def render(self, seeds, accum=None, region=None, threads=0):
        """accum += one clamped sample per seed (in order). Returns (accum[H,W,3], OrcStats)."""
        seeds = np.ascontiguousarray(np.asarray(seeds, dtype=np.int32))
        if accum is None:
            accum = np.zeros((self.height, self.width, 3), dtype=np.float32)
        assert accum.dtype == np.float32 and accum.flags["C_CONTIGUOUS"]
        x0, y0, x1, y1 = region if region is not None else (0, 0, self.width, self.height)
        st = OrcStats()
        rc = lib().orc_render(C.byref(self.c), _ptr(seeds, C.c_int32), len(seeds), _ptr(accum, C.c_float),
                              x0, y0, x1, y1, threads, C.byref(st))
        if rc != 0:
            raise RuntimeError("orc_render failed rc=%d" % rc)
        return accum, st

    def render_by_depth(self, seeds, n_buckets=64, region=None):
        """Analysis: (colourSum[H,W,K,3], count[H,W,K]) of the clamped samples by the deepest radiance rtTrace of the sample
        (pt_oracle.c orc_render_by_depth): the image under a stack overflow at depth D is a sum over these buckets."""
        seeds = np.ascontiguousarray(np.asarray(seeds, dtype=np.int32))
        cs = np.zeros((self.height, self.width, n_buckets, 3), np.float32)
        cn = np.zeros((self.height, self.width, n_buckets), np.float32)
        x0, y0, x1, y1 = region if region is not None else (0, 0, self.width, self.height)
        rc = lib().orc_render_by_depth(C.byref(self.c), _ptr(seeds, C.c_int32), len(seeds), x0, y0, x1, y1, n_buckets,
                                       _ptr(cs, C.c_float), _ptr(cn, C.c_float))
        if rc != 0:
            raise RuntimeError("orc_render_by_depth failed rc=%d" % rc)
        return cs, cn

    def render_clamp_stats(self, seeds, region, cap=1e3):
        """Analysis: (rawSum, nClamped, clampedSum), each [H,W,3], of the samples in `region` before / after Camera.cu:39's clamp."""
        seeds = np.ascontiguousarray(np.asarray(seeds, dtype=np.int32))
        out = [np.zeros((self.height, self.width, 3), np.float32) for _ in range(3)]
        x0, y0, x1, y1 = region
        rc = lib().orc_render_clamp_stats(C.byref(self.c), _ptr(seeds, C.c_int32), len(seeds), x0, y0, x1, y1, float(cap),
                                          *[_ptr(a, C.c_float) for a in out])
        if rc != 0:
            raise RuntimeError("orc_render_clamp_stats failed rc=%d" % rc)
        return out

    def closest_hit(self, org, dirn, tmin=1e-3, tmax=1e27):
        t = C.c_float(0)
        prim = lib().orc_closest_hit(C.byref(self.c), f3(*org), f3(*dirn), tmin, tmax, C.byref(t))
        return prim, t.value

    def closest_hits(self, rays):
        """rays: (n, 8) = o, d, tmin, tmax (the layout of moptix_debug_trace) -> (prim[n] or -1, t[n])."""
        rays = np.ascontiguousarray(np.asarray(rays, np.float32).reshape(-1, 8))
        prim = np.zeros(len(rays), np.int32); t = np.zeros(len(rays), np.float32)
        lib().orc_closest_hit_batch(C.byref(self.c), _ptr(rays, C.c_float), len(rays), _ptr(prim, C.c_int32), _ptr(t, C.c_float))
        return prim, t


def image_from_accum(accum, spp):
    """MinimalOptiX.cpp:43-66 updateContent: clamp(accu/spp,0,1), flip rows (row 0 = bottom)."""
    img = np.clip(accum / np.float32(spp), 0.0, 1.0)
    return img[::-1].copy()


def rgb8_from_accum(accum, n_accumulation):
    """MinimalOptiX.cpp:43-66 updateContent down to the bytes of the RGB888 canvas: QColor::setRedF stores
    qRound(v * 65535) in 16 bits, QImage::Format_RGB888 keeps the high byte; rows flipped (row 0 = top)."""
    v = np.clip(np.asarray(accum, np.float32) / np.float32(n_accumulation), np.float32(0.0), np.float32(1.0))
    q = (v * np.float32(65535.0) + np.float32(0.5)).astype(np.uint32) >> 8
    return q.astype(np.uint8)[::-1].copy()
