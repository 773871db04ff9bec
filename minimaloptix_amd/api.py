"""Thin Python wrappers over the C ABI (include/moptix.h, include/moptix_host.h)."""
import ctypes as C
import os

import numpy as np

from . import _capi as K
from ._capi import MoptixError


def scenes_dir():
    """Folder holding `<name>/<name>.scene` (the reference's baseSceneFolder "scenes/")."""
    return os.path.join(K.REPO_ROOT, "scenes") + "/"


def _tea16(v0, v1):
    """utils_device.h:8-22 (integer; used only for the seed schedule on the host)."""
    v0 &= 0xffffffff; v1 &= 0xffffffff; s0 = 0
    for _ in range(16):
        s0 = (s0 + 0x9e3779b9) & 0xffffffff
        v0 = (v0 + ((((v1 << 4) + 0xa341316c) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4)) & 0xffffffff)) & 0xffffffff
        v1 = (v1 + ((((v0 << 4) + 0xad90777d) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761e)) & 0xffffffff)) & 0xffffffff
    return v0


def launch_seeds(n, base_seed=0, first=0):
    """Seed schedule: launchSeed(i) = (int)tea<16>(i, baseSeed)."""
    return np.array([_tea16(first + i, base_seed) for i in range(n)], dtype=np.uint32).view(np.int32)


def _f3(v):
    return [float(v.x), float(v.y), float(v.z)]


class HostScene:
    """A scene built by the C++ host library (scene.cpp/tinyobj/setupScene equivalents)."""

    def __init__(self, kind, width, height, iarg=0, farg=0.0, base_folder=None, skip_missing=True):
        L = K.host_lib()
        self._h = C.c_void_p()
        base = (base_folder or scenes_dir()).encode()
        rc = L.mohost_scene_build(kind.encode(), base, width, height, int(iarg), float(farg), 1 if skip_missing else 0,
                                  C.byref(self._h))
        if rc != K.MOPTIX_OK:
            raise MoptixError(rc, L.mohost_last_error().decode())
        self.kind = kind
        self.sizes = K.SceneSizes()
        L.mohost_scene_get_sizes(self._h, C.byref(self.sizes))
        self.params = K.Params()
        amin, amax, accel = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_char * 16)()
        L.mohost_scene_get_params(self._h, C.byref(self.params), amin, amax, accel)
        self.aabb_min, self.aabb_max = np.array(list(amin), np.float32), np.array(list(amax), np.float32)
        self.accel = accel.value.decode()
        self.warnings = [L.mohost_scene_warning(self._h, i).decode() for i in range(self.sizes.nWarnings)]
        self._flat = None

    def __del__(self):
        try:
            if self._h:
                K.host_lib().mohost_scene_free(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def width(self):
        return int(self.params.width)

    @property
    def height(self):
        return int(self.params.height)

    def upload(self, ctx):
        rc = K.host_lib().mohost_scene_upload(self._h, ctx._h)
        if rc != K.MOPTIX_OK:
            raise MoptixError(rc, ctx.last_error())

    def flat(self):
        """Flattened arrays (ctypes records + numpy) of the whole scene."""
        if self._flat is None:
            s = self.sizes
            mats = (K.Material * max(1, s.nMaterials))()
            sph = (K.SphereParams * max(1, s.nSpheres))(); smat = np.zeros(max(1, s.nSpheres), np.int32)
            quads = (K.QuadParams * max(1, s.nQuads))(); qmat = np.zeros(max(1, s.nQuads), np.int32)
            lights = (K.LightParams * max(1, s.nLights))()
            pos = np.zeros((max(1, s.nVerts), 3), np.float32); nrm = np.zeros((max(1, s.nNormals), 3), np.float32)
            vi = np.zeros((max(1, s.nFaces), 3), np.int32); ni = np.zeros((max(1, s.nFaces), 3), np.int32)
            fm = np.zeros(max(1, s.nFaces), np.int32)
            ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
            fp = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
            K.host_lib().mohost_scene_copy(self._h, mats, sph, ip(smat), quads, ip(qmat), lights, fp(pos), fp(nrm),
                                           ip(vi), ip(ni), ip(fm))
            uv = np.zeros((max(1, s.nTexcoords), 2), np.float32); ti = np.full((max(1, s.nFaces), 3), -1, np.int32)
            K.host_lib().mohost_scene_copy_texcoords(self._h, fp(uv), ip(ti))
            textures = []
            for t in range(s.nTextures):
                w, h = C.c_int32(), C.c_int32()
                K.host_lib().mohost_scene_texture(self._h, t, C.byref(w), C.byref(h), None)
                px = np.zeros((h.value, w.value, 4), np.float32)
                K.host_lib().mohost_scene_texture(self._h, t, None, None, fp(px))
                textures.append(px)
            self._flat = dict(materials=mats, spheres=sph, sphereMat=smat[:s.nSpheres], quads=quads, quadMat=qmat[:s.nQuads],
                              lights=lights, positions=pos[:s.nVerts], normals=nrm[:s.nNormals], vIdx=vi[:s.nFaces],
                              nIdx=ni[:s.nFaces], faceMat=fm[:s.nFaces], texcoords=uv[:s.nTexcoords], tIdx=ti[:s.nFaces],
                              textures=textures)
        return self._flat

    def to_dict(self):
        """Plain-python description (what oracle.oracle.Scene consumes)."""
        f, s, p = self.flat(), self.sizes, self.params
        mats = []
        for i in range(s.nMaterials):
            m = f["materials"][i]; d = m.disney
            e = _f3(d.emission) if m.kind == K.MAT_DISNEY else _f3(m.emission)
            mats.append(dict(kind=int(m.kind), albedo=_f3(m.albedo), fuzz=float(m.fuzz), refIdx=float(m.refIdx), emission=e,
                             color=_f3(d.color), metallic=d.metallic, subsurface=d.subsurface, specular=d.specular,
                             roughness=d.roughness, specularTint=d.specularTint, anisotropic=d.anisotropic, sheen=d.sheen,
                             sheenTint=d.sheenTint, clearcoat=d.clearcoat, clearcoatGloss=d.clearcoatGloss,
                             brdfType=int(d.brdfType), albedoID=int(d.albedoID)))
        spheres = np.array([[*_f3(f["spheres"][i].center), f["spheres"][i].radius] for i in range(s.nSpheres)], np.float32).reshape(-1, 4)
        quads = np.array([[q.plane.x, q.plane.y, q.plane.z, q.plane.w, *_f3(q.v1), *_f3(q.v2), *_f3(q.anchor)]
                          for q in (f["quads"][i] for i in range(s.nQuads))], np.float32).reshape(-1, 13)
        lights = [dict(position=_f3(l.position), normal=_f3(l.normal), emission=_f3(l.emission), u=_f3(l.u), v=_f3(l.v),
                       area=float(l.area), radius=float(l.radius), shape=int(l.shape))
                  for l in (f["lights"][i] for i in range(s.nLights))]
        cam = p.cam
        return dict(width=int(p.width), height=int(p.height),
                    cam=dict(origin=_f3(cam.origin), horizontal=_f3(cam.horizontal), vertical=_f3(cam.vertical),
                             scrLowerLeftCorner=_f3(cam.scrLowerLeftCorner), u=_f3(cam.u), v=_f3(cam.v),
                             lensRadius=float(cam.lensRadius)),
                    bgColor=_f3(p.bgColor), rayMaxDepth=int(p.rayMaxDepth), rayMinIntensity=float(p.rayMinIntensity),
                    rayEpsilonT=float(p.rayEpsilonT), materials=mats, spheres=spheres, sphereMat=f["sphereMat"],
                    quads=quads, quadMat=f["quadMat"], lights=lights, positions=f["positions"], normals=f["normals"],
                    vIdx=f["vIdx"], nIdx=f["nIdx"], faceMat=f["faceMat"], texcoords=f["texcoords"], tIdx=f["tIdx"],
                    textures=f["textures"])

    def face_arrays(self):
        """Per-face positions / normals (9 floats each) + flags, as moptix_add_mesh flattens them."""
        f = self.flat()
        vi, ni = f["vIdx"], f["nIdx"]
        nf = len(vi)
        face_pos = np.ascontiguousarray(f["positions"][vi.reshape(-1)].reshape(nf, 9)) if nf else np.zeros((0, 9), np.float32)
        has = (ni >= 0).all(axis=1).astype(np.int32) if nf else np.zeros(0, np.int32)
        face_nrm = np.zeros((nf, 9), np.float32)
        if nf and has.any():
            idx = np.where(has.astype(bool))[0]
            face_nrm[idx] = f["normals"][ni[idx].reshape(-1)].reshape(-1, 9)
        return face_pos, face_nrm, has, f["faceMat"]

    def face_uvs(self):
        """Per-face texcoords (6 floats: u0 v0 u1 v1 u2 v2) + flags, as moptix_add_mesh flattens them."""
        f = self.flat()
        ti = f["tIdx"]
        nf = len(ti)
        has = (ti >= 0).all(axis=1).astype(np.int32) if nf else np.zeros(0, np.int32)
        uv = np.zeros((nf, 6), np.float32)
        if nf and has.any():
            idx = np.where(has.astype(bool))[0]
            uv[idx] = f["texcoords"][ti[idx].reshape(-1)].reshape(-1, 6)
        return uv, has


class Context:
    """One moptix_context (one GPU)."""

    def __init__(self, device=0):
        self._L = K.device_lib()
        self._h = C.c_void_p()
        rc = self._L.moptix_create(C.byref(self._h), device)
        if rc != K.MOPTIX_OK:
            raise MoptixError(rc, self._L.moptix_last_error(None).decode())
        self.width = self.height = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.moptix_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_error(self):
        return self._L.moptix_last_error(self._h).decode()

    def _chk(self, rc):
        if rc != K.MOPTIX_OK:
            raise MoptixError(rc, self.last_error())

    def load(self, scene):
        """Upload a HostScene (set_params + add_* + build_accel) and validate."""
        scene.upload(self)
        self.width, self.height = scene.width, scene.height
        self._chk(self._L.moptix_validate(self._h))

    def set_params(self, params):
        self._chk(self._L.moptix_set_params(self._h, C.byref(params)))
        self.width, self.height = int(params.width), int(params.height)

    def set_option(self, name, value):
        self._chk(self._L.moptix_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name):
        v = C.c_int32()
        self._chk(self._L.moptix_get_option(self._h, name.encode(), C.byref(v)))
        return v.value

    def set_partition(self, rank, nranks):
        self._chk(self._L.moptix_set_partition(self._h, rank, nranks))

    def update_spheres(self, first, spheres, n):
        """spheres: ctypes array of SphereParams (updateVideo, MinimalOptiX.cpp:763-764)."""
        self._chk(self._L.moptix_update_spheres(self._h, int(first), spheres, int(n)))

    def build_accel(self, kind):
        self._chk(self._L.moptix_build_accel(self._h, kind.encode()))

    def accel_info(self):
        a = K.AccelInfo()
        self._chk(self._L.moptix_get_accel_info(self._h, C.byref(a)))
        return a

    @staticmethod
    def _seeds(seeds):
        s = np.ascontiguousarray(np.asarray(seeds, dtype=np.int32))
        return s, s.ctypes.data_as(C.POINTER(C.c_int32))

    def launch(self, seed):
        self._chk(self._L.moptix_launch(self._h, int(seed)))

    def render(self, seeds):
        s, p = self._seeds(seeds)
        self._chk(self._L.moptix_render(self._h, p, len(s)))

    def render_async(self, seeds):
        s, p = self._seeds(seeds)
        self._chk(self._L.moptix_render_async(self._h, p, len(s)))

    def sync(self):
        self._chk(self._L.moptix_sync(self._h))

    def render_counted(self, seeds):
        s, p = self._seeds(seeds)
        st = K.Stats()
        self._chk(self._L.moptix_render_counted(self._h, p, len(s), C.byref(st)))
        return st

    def accum_read(self):
        out = np.empty((self.height, self.width, 3), np.float32)
        self._chk(self._L.moptix_accum_read(self._h, out.ctypes.data_as(C.POINTER(C.c_float))))
        return out

    def accum_clear(self):
        self._chk(self._L.moptix_accum_clear(self._h))

    def accum_device_ptr(self):
        p = C.c_void_p()
        self._chk(self._L.moptix_accum_device_ptr(self._h, C.byref(p)))
        return p.value

    def accum_bind(self, dev_ptr):
        self._chk(self._L.moptix_accum_bind(self._h, C.c_void_p(dev_ptr)))

    def set_stream(self, hip_stream):
        self._chk(self._L.moptix_set_stream(self._h, C.c_void_p(hip_stream)))

    def resolve_rgb8(self, n_accumulation, clear=False):
        out = np.empty((self.height, self.width, 3), np.uint8)
        self._chk(self._L.moptix_resolve_rgb8(self._h, float(n_accumulation), 1 if clear else 0,
                                              out.ctypes.data_as(C.POINTER(C.c_uint8))))
        return out

    def kernel_time(self, reset=False):
        ms, n = C.c_double(), C.c_uint64()
        self._chk(self._L.moptix_kernel_time(self._h, C.byref(ms), C.byref(n), 1 if reset else 0))
        return ms.value, n.value

    def reduce_time(self):
        ms = C.c_double()
        self._chk(self._L.moptix_reduce_time(self._h, C.byref(ms)))
        return ms.value

    # ---- multi-GPU collectives on RCCL, behind the C ABI (include/moptix.h "Multi-GPU collectives") ----
    @staticmethod
    def comm_unique_id():
        """ncclGetUniqueId as 128 bytes: made by one rank, handed to the others by the host (bench.py: torch.distributed)."""
        buf = (C.c_uint8 * 128)()
        rc = K.device_lib().moptix_comm_unique_id(buf)
        if rc != K.MOPTIX_OK:
            raise MoptixError(rc, K.device_lib().moptix_last_error(None).decode())
        return bytes(buf)

    def comm_init(self, unique_id, rank, nranks):
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(unique_id))
        self._chk(self._L.moptix_comm_init(self._h, buf, int(rank), int(nranks)))

    def comm_destroy(self):
        self._chk(self._L.moptix_comm_destroy(self._h))

    def gather_tiles(self, dst=0):
        """Tile split: every rank's tiles into rank dst's accuBuffer (grouped ncclSend / ncclRecv + unpack on the device)."""
        self._chk(self._L.moptix_gather_tiles(self._h, int(dst)))

    def reduce_frame(self, dst=0):
        """Sample split: ncclReduce(sum) of the accuBuffers into rank dst's."""
        self._chk(self._L.moptix_reduce_frame(self._h, int(dst)))

    def packed_tile_floats(self, nranks):
        n = C.c_uint64()
        self._chk(self._L.moptix_packed_tile_floats(self._h, int(nranks), C.byref(n)))
        return int(n.value)

    def pack_tiles(self, rank, nranks, dst_device_ptr):
        self._chk(self._L.moptix_pack_tiles(self._h, int(rank), int(nranks), C.c_void_p(dst_device_ptr)))

    def unpack_tiles(self, rank, nranks, src_device_ptr):
        self._chk(self._L.moptix_unpack_tiles(self._h, int(rank), int(nranks), C.c_void_p(src_device_ptr)))

    def debug_read_accel(self):
        a = self.accel_info()
        nodes = np.zeros((max(1, a.nNodes), 32), np.uint32)     # Node128 = 32 words
        tris = np.zeros((max(1, a.nTriangles), 12), np.uint32)
        prim = np.zeros(max(1, a.nTriangles), np.int32)
        self._chk(self._L.moptix_debug_read_accel(self._h, nodes.ctypes.data, tris.ctypes.data,
                                                  prim.ctypes.data_as(C.POINTER(C.c_int32))))
        return nodes[:a.nNodes], tris[:a.nTriangles], prim[:a.nTriangles]

    def debug_read_nodes64(self):
        """The nodes as the trace kernels fetch them: [nNodes, 16] words (Node64: corner xyz, step xyz, 6 plane words, 4 refs)."""
        a = self.accel_info()
        nodes = np.zeros((max(1, a.nNodes), 16), np.uint32)
        self._chk(self._L.moptix_debug_read_nodes64(self._h, nodes.ctypes.data))
        return nodes[:a.nNodes]

    def debug_trace(self, rays):
        rays = np.ascontiguousarray(np.asarray(rays, np.float32).reshape(-1, 8))
        n = len(rays)
        t = np.zeros(n, np.float32); prim = np.zeros(n, np.int32)
        self._chk(self._L.moptix_debug_trace(self._h, rays.ctypes.data_as(C.POINTER(C.c_float)), n,
                                             t.ctypes.data_as(C.POINTER(C.c_float)), prim.ctypes.data_as(C.POINTER(C.c_int32))))
        return t, prim
