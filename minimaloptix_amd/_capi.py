"""ctypes mirror of include/moptix.h and include/moptix_host.h.

The product's compute lives in minimaloptix_amd/lib/libmoptix.so (HIP, gfx950).  This
module only loads it; if the library is missing or there is no MI355X, calls fail loudly
(MoptixError) -- there is no Python/CPU fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
REPO_ROOT = os.path.dirname(_HERE)

MOPTIX_OK = 0
MOPTIX_ERR_INVALID, MOPTIX_ERR_NO_DEVICE, MOPTIX_ERR_HIP, MOPTIX_ERR_STATE, MOPTIX_ERR_LIMIT, MOPTIX_ERR_COMM = -1, -2, -3, -4, -5, -6
ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_STATE, ERR_LIMIT, ERR_COMM = -1, -2, -3, -4, -5, -6
MAT_LAMBERTIAN, MAT_METAL, MAT_GLASS, MAT_DISNEY, MAT_LIGHT = range(5)
BRDF_NORMAL, BRDF_GLASS = 0, 1
LIGHT_SPHERE, LIGHT_QUAD = 0, 1


class MoptixError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("moptix error %d: %s" % (code, msg))
        self.code = code


class Float3(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float)]

    def tolist(self):
        return [self.x, self.y, self.z]


class Float4(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("w", C.c_float)]


class CamParams(C.Structure):
    _fields_ = [("origin", Float3), ("horizontal", Float3), ("vertical", Float3),
                ("scrLowerLeftCorner", Float3), ("u", Float3), ("v", Float3), ("lensRadius", C.c_float)]


class SphereParams(C.Structure):
    _fields_ = [("radius", C.c_float), ("center", Float3), ("velocity", Float3)]


class QuadParams(C.Structure):
    _fields_ = [("plane", Float4), ("v1", Float3), ("v2", Float3), ("anchor", Float3)]


class DisneyParams(C.Structure):
    _fields_ = [("albedoID", C.c_int32), ("color", Float3), ("emission", Float3),
                ("metallic", C.c_float), ("subsurface", C.c_float), ("specular", C.c_float),
                ("roughness", C.c_float), ("specularTint", C.c_float), ("anisotropic", C.c_float),
                ("sheen", C.c_float), ("sheenTint", C.c_float), ("clearcoat", C.c_float),
                ("clearcoatGloss", C.c_float), ("brdfType", C.c_int32)]


class LightParams(C.Structure):
    _fields_ = [("position", Float3), ("normal", Float3), ("emission", Float3), ("u", Float3), ("v", Float3),
                ("area", C.c_float), ("radius", C.c_float), ("shape", C.c_int32)]


class Material(C.Structure):
    _fields_ = [("kind", C.c_int32), ("albedo", Float3), ("fuzz", C.c_float), ("refIdx", C.c_float),
                ("emission", Float3), ("disney", DisneyParams)]


class Params(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("rayMaxDepth", C.c_uint32),
                ("rayMinIntensity", C.c_float), ("rayEpsilonT", C.c_float), ("bgColor", Float3), ("cam", CamParams)]


class Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in
                ("samples", "primaryRays", "bounceRays", "shadowRays", "nodeFetches", "triTests", "closestHits",
                 "lightLoads", "analyticTests", "traversalSteps", "activeLaneSteps", "shadeBatches", "shadeBatchLanes")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}

    @property
    def rays(self):
        return int(self.primaryRays + self.bounceRays + self.shadowRays)


class AccelInfo(C.Structure):
    _fields_ = [("nTriangles", C.c_uint32), ("nNodes", C.c_uint32), ("maxLeafSize", C.c_uint32),
                ("treeDepth", C.c_uint32), ("buildMs", C.c_float), ("nodeBytes", C.c_uint64), ("triBytes", C.c_uint64)]


class SceneSizes(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("nMaterials", "nSpheres", "nQuads", "nLights", "nVerts", "nNormals",
                                          "nTexcoords", "nFaces", "nMeshes", "nWarnings", "nTextures")]


class RenderResult(C.Structure):
    _fields_ = [("renderMs", C.c_double), ("bvhBuildMs", C.c_float), ("nVertices", C.c_uint64),
                ("nFaces", C.c_uint64), ("nNodes", C.c_uint32), ("treeDepth", C.c_uint32)]


# every symbol include/moptix.h declares (tests check that the library exports all of them)
DEVICE_SYMBOLS = [
    "moptix_create", "moptix_destroy", "moptix_last_error", "moptix_version", "moptix_set_stream",
    "moptix_set_params", "moptix_clear_scene", "moptix_add_texture", "moptix_add_material", "moptix_add_spheres", "moptix_add_quads",
    "moptix_add_mesh", "moptix_set_lights", "moptix_update_spheres", "moptix_build_accel", "moptix_get_accel_info",
    "moptix_validate", "moptix_launch", "moptix_render", "moptix_render_async", "moptix_sync",
    "moptix_render_counted", "moptix_set_partition", "moptix_set_option", "moptix_get_option",
    "moptix_accum_read", "moptix_accum_clear", "moptix_accum_device_ptr", "moptix_accum_bind",
    "moptix_resolve_rgb8", "moptix_kernel_time", "moptix_reduce_time", "moptix_debug_read_accel", "moptix_debug_read_nodes64", "moptix_debug_trace",
    "moptix_comm_unique_id", "moptix_comm_init", "moptix_comm_destroy", "moptix_gather_tiles", "moptix_reduce_frame",
    "moptix_packed_tile_floats", "moptix_pack_tiles", "moptix_unpack_tiles",
]
HOST_SYMBOLS = [
    "mohost_last_error", "mohost_scene_build", "mohost_scene_free", "mohost_scene_get_sizes",
    "mohost_scene_get_params", "mohost_scene_warning", "mohost_scene_copy", "mohost_scene_upload",
    "mohost_set_quad_params", "mohost_set_cam_params", "mohost_obj_stats", "mohost_render_scene",
    "mohost_animate_spheres", "mohost_video_camera", "mohost_scene_copy_texcoords", "mohost_scene_texture",
    "mohost_read_image",
]

_dev = None
_host = None


def _load(name):
    path = os.path.join(LIB_DIR, name)
    if not os.path.exists(path):
        raise MoptixError(ERR_NO_DEVICE, "%s not found -- run `make` (or __graft_entry__.build()); "
                                          "there is no fallback implementation" % path)
    return C.CDLL(path, mode=C.RTLD_GLOBAL)


def device_lib():
    global _dev
    if _dev is None:
        L = _load(os.environ.get("MOPTIX_DEVICE_LIB", "libmoptix.so"))   # alternate builds for A/B experiments
        vp, i32, f32p, i32p = C.c_void_p, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_int32)
        L.moptix_create.argtypes = [C.POINTER(vp), C.c_int]
        L.moptix_destroy.argtypes = [vp]
        L.moptix_last_error.argtypes = [vp]; L.moptix_last_error.restype = C.c_char_p
        L.moptix_version.restype = C.c_char_p
        L.moptix_set_stream.argtypes = [vp, vp]
        L.moptix_set_params.argtypes = [vp, C.POINTER(Params)]
        L.moptix_clear_scene.argtypes = [vp]
        L.moptix_add_material.argtypes = [vp, C.POINTER(Material), i32p]
        L.moptix_add_texture.argtypes = [vp, f32p, i32, i32, i32p]
        L.moptix_add_spheres.argtypes = [vp, C.POINTER(SphereParams), i32p, i32]
        L.moptix_add_quads.argtypes = [vp, C.POINTER(QuadParams), i32p, i32]
        L.moptix_add_mesh.argtypes = [vp, f32p, i32, f32p, i32, f32p, i32, i32p, i32p, i32p, i32, i32]
        L.moptix_set_lights.argtypes = [vp, C.POINTER(LightParams), i32]
        L.moptix_update_spheres.argtypes = [vp, i32, C.POINTER(SphereParams), i32]
        L.moptix_build_accel.argtypes = [vp, C.c_char_p]
        L.moptix_get_accel_info.argtypes = [vp, C.POINTER(AccelInfo)]
        L.moptix_validate.argtypes = [vp]
        L.moptix_launch.argtypes = [vp, i32]
        L.moptix_render.argtypes = [vp, i32p, i32]
        L.moptix_render_async.argtypes = [vp, i32p, i32]
        L.moptix_sync.argtypes = [vp]
        L.moptix_render_counted.argtypes = [vp, i32p, i32, C.POINTER(Stats)]
        L.moptix_set_partition.argtypes = [vp, i32, i32]
        L.moptix_set_option.argtypes = [vp, C.c_char_p, i32]
        L.moptix_get_option.argtypes = [vp, C.c_char_p, i32p]
        L.moptix_accum_read.argtypes = [vp, f32p]
        L.moptix_accum_clear.argtypes = [vp]
        L.moptix_accum_device_ptr.argtypes = [vp, C.POINTER(vp)]
        L.moptix_accum_bind.argtypes = [vp, vp]
        L.moptix_resolve_rgb8.argtypes = [vp, C.c_float, C.c_int, C.POINTER(C.c_uint8)]
        L.moptix_kernel_time.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.c_int]
        L.moptix_reduce_time.argtypes = [vp, C.POINTER(C.c_double)]
        L.moptix_debug_read_accel.argtypes = [vp, vp, vp, i32p]
        L.moptix_debug_read_nodes64.argtypes = [vp, vp]
        L.moptix_debug_trace.argtypes = [vp, f32p, i32, f32p, i32p]
        u8p = C.POINTER(C.c_uint8)
        L.moptix_comm_unique_id.argtypes = [u8p]
        L.moptix_comm_init.argtypes = [vp, u8p, i32, i32]
        L.moptix_comm_destroy.argtypes = [vp]
        L.moptix_gather_tiles.argtypes = [vp, i32]
        L.moptix_reduce_frame.argtypes = [vp, i32]
        L.moptix_packed_tile_floats.argtypes = [vp, i32, C.POINTER(C.c_uint64)]
        L.moptix_pack_tiles.argtypes = [vp, i32, i32, vp]
        L.moptix_unpack_tiles.argtypes = [vp, i32, i32, vp]
        _dev = L
    return _dev


def host_lib():
    global _host
    if _host is None:
        device_lib()
        L = _load("libmoptix_host.so")
        vp, i32, f32p, i32p = C.c_void_p, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_int32)
        L.mohost_last_error.restype = C.c_char_p
        L.mohost_scene_build.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.c_uint32, i32, C.c_float, C.c_int, C.POINTER(vp)]
        L.mohost_scene_free.argtypes = [vp]; L.mohost_scene_free.restype = None
        L.mohost_scene_get_sizes.argtypes = [vp, C.POINTER(SceneSizes)]
        L.mohost_scene_get_params.argtypes = [vp, C.POINTER(Params), C.c_float * 3, C.c_float * 3, C.c_char * 16]
        L.mohost_scene_warning.argtypes = [vp, i32]; L.mohost_scene_warning.restype = C.c_char_p
        L.mohost_scene_copy.argtypes = [vp, C.POINTER(Material), C.POINTER(SphereParams), i32p, C.POINTER(QuadParams), i32p,
                                        C.POINTER(LightParams), f32p, f32p, i32p, i32p, i32p]
        L.mohost_scene_upload.argtypes = [vp, vp]
        L.mohost_scene_copy_texcoords.argtypes = [vp, f32p, i32p]
        L.mohost_scene_texture.argtypes = [vp, i32, i32p, i32p, f32p]
        L.mohost_read_image.argtypes = [C.c_char_p, i32p, i32p, C.POINTER(C.c_uint8), C.c_uint64]
        L.mohost_set_quad_params.argtypes = [C.c_float * 3, C.c_float * 3, C.c_float * 3, C.POINTER(QuadParams)]
        L.mohost_set_quad_params.restype = None
        L.mohost_set_cam_params.argtypes = [C.c_float * 3, C.c_float * 3, C.c_float * 3, C.c_float, C.c_float, C.c_float,
                                            C.c_float, C.POINTER(CamParams)]
        L.mohost_set_cam_params.restype = None
        L.mohost_animate_spheres.argtypes = [C.POINTER(SphereParams), i32, C.c_float, C.POINTER(C.c_float)]
        L.mohost_animate_spheres.restype = None
        L.mohost_video_camera.argtypes = [C.c_float, C.c_float, C.POINTER(CamParams)]
        L.mohost_video_camera.restype = None
        L.mohost_obj_stats.argtypes = [C.c_char_p, i32p, i32p, i32p, i32p]
        L.mohost_render_scene.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                          C.c_int, C.c_char_p, C.c_char_p, C.POINTER(C.c_uint8), C.POINTER(RenderResult)]
        _host = L
    return _host
