#include "../../include/moptix_host.h"

#include <cstring>
#include <exception>
#include <string>

#include "image_io.h"
#include "minimal_optix.h"
#include "obj_loader.h"
#include "scene_desc.h"

using namespace moptix;

struct mohost_scene_t { SceneDesc desc; };

static thread_local std::string g_err;
const char* mohost_last_error(void) { return g_err.c_str(); }

int mohost_scene_build(const char* kind, const char* baseFolder, uint32_t width, uint32_t height,
                       int32_t iarg, float farg, int skipMissing, mohost_scene* out) {
  if (!kind || !out) { g_err = "null argument"; return MOPTIX_ERR_INVALID; }
  try {
    mohost_scene s = new mohost_scene_t();
    const std::string k(kind), base(baseFolder ? baseFolder : "scenes/");
    if (k == "spheres") buildSpheresScene(s->desc, width, height, farg);
    else if (k.rfind("file:", 0) == 0) buildFileScene(s->desc, base, k.substr(5), width, height, skipMissing != 0);
    else if (k == "random_spheres") buildRandomSpheresScene(s->desc, iarg, width, height);
    else if (k == "cornell_quads") buildCornellQuadsScene(s->desc, width, height);
    else if (k == "dining_standin") buildDiningStandInScene(s->desc, base, iarg, width, height);
    else if (k == "coffee_pot_standin") buildCoffeePotStandInScene(s->desc, base, width, height);
    else if (k == "million_standin") buildProceduralMillionScene(s->desc, iarg, width, height);
    else { delete s; g_err = "unknown scene kind: " + k; return MOPTIX_ERR_INVALID; }
    *out = s;
    return MOPTIX_OK;
  } catch (const std::exception& e) { g_err = e.what(); return MOPTIX_ERR_INVALID; }
}

void mohost_scene_free(mohost_scene s) { delete s; }

static void counts(const SceneDesc& d, mohost_scene_sizes* o) {
  memset(o, 0, sizeof(*o));
  o->nMaterials = (int32_t)d.materials.size(); o->nSpheres = (int32_t)d.spheres.size(); o->nQuads = (int32_t)d.quads.size();
  o->nLights = (int32_t)d.lights.size(); o->nMeshes = (int32_t)d.meshes.size(); o->nWarnings = (int32_t)d.warnings.size();
  o->nTextures = (int32_t)d.textures.size();
  for (const MeshDesc& m : d.meshes) {
    o->nVerts += (int32_t)(m.positions.size() / 3); o->nNormals += (int32_t)(m.normals.size() / 3);
    o->nTexcoords += (int32_t)(m.texcoords.size() / 2); o->nFaces += (int32_t)(m.vIdx.size() / 3);
  }
}

int mohost_scene_get_sizes(mohost_scene s, mohost_scene_sizes* out) {
  if (!s || !out) return MOPTIX_ERR_INVALID;
  counts(s->desc, out);
  return MOPTIX_OK;
}

int mohost_scene_get_params(mohost_scene s, moptix_params* out, float aabbMin[3], float aabbMax[3], char accel[16]) {
  if (!s) return MOPTIX_ERR_INVALID;
  if (out) *out = s->desc.params;
  if (aabbMin) { aabbMin[0] = s->desc.aabb.m_min.x; aabbMin[1] = s->desc.aabb.m_min.y; aabbMin[2] = s->desc.aabb.m_min.z; }
  if (aabbMax) { aabbMax[0] = s->desc.aabb.m_max.x; aabbMax[1] = s->desc.aabb.m_max.y; aabbMax[2] = s->desc.aabb.m_max.z; }
  if (accel) { strncpy(accel, s->desc.accel.c_str(), 15); accel[15] = 0; }
  return MOPTIX_OK;
}

const char* mohost_scene_warning(mohost_scene s, int32_t i) {
  if (!s || i < 0 || i >= (int32_t)s->desc.warnings.size()) return "";
  return s->desc.warnings[i].c_str();
}

int mohost_scene_copy(mohost_scene s, moptix_material* materials, moptix_sphere_params* spheres, int32_t* sphereMat,
                      moptix_quad_params* quads, int32_t* quadMat, moptix_light_params* lights,
                      float* positions, float* normals, int32_t* vIdx, int32_t* nIdx, int32_t* faceMat) {
  if (!s) return MOPTIX_ERR_INVALID;
  const SceneDesc& d = s->desc;
  if (materials) memcpy(materials, d.materials.data(), d.materials.size() * sizeof(moptix_material));
  if (spheres) memcpy(spheres, d.spheres.data(), d.spheres.size() * sizeof(moptix_sphere_params));
  if (sphereMat) memcpy(sphereMat, d.sphereMat.data(), d.sphereMat.size() * sizeof(int32_t));
  if (quads) memcpy(quads, d.quads.data(), d.quads.size() * sizeof(moptix_quad_params));
  if (quadMat) memcpy(quadMat, d.quadMat.data(), d.quadMat.size() * sizeof(int32_t));
  if (lights) memcpy(lights, d.lights.data(), d.lights.size() * sizeof(moptix_light_params));
  size_t vOff = 0, nOff = 0, fOff = 0;
  for (const MeshDesc& m : d.meshes) {
    const size_t nv = m.positions.size() / 3, nn = m.normals.size() / 3, nf = m.vIdx.size() / 3;
    if (positions) memcpy(positions + 3 * vOff, m.positions.data(), m.positions.size() * sizeof(float));
    if (normals && nn) memcpy(normals + 3 * nOff, m.normals.data(), m.normals.size() * sizeof(float));
    for (size_t f = 0; f < nf; f++) {
      const bool hasN = nn > 0 && m.nIdx[3 * f] >= 0 && m.nIdx[3 * f + 1] >= 0 && m.nIdx[3 * f + 2] >= 0;
      for (int k = 0; k < 3; k++) {
        if (vIdx) vIdx[3 * (fOff + f) + k] = m.vIdx[3 * f + k] + (int32_t)vOff;
        if (nIdx) nIdx[3 * (fOff + f) + k] = hasN ? m.nIdx[3 * f + k] + (int32_t)nOff : -1;
      }
      if (faceMat) faceMat[fOff + f] = m.matId;
    }
    vOff += nv; nOff += nn; fOff += nf;
  }
  return MOPTIX_OK;
}

int mohost_scene_copy_texcoords(mohost_scene s, float* texcoords, int32_t* tIdx) {
  if (!s) return MOPTIX_ERR_INVALID;
  size_t tOff = 0, fOff = 0;
  for (const MeshDesc& m : s->desc.meshes) {
    const size_t nt = m.texcoords.size() / 2, nf = m.vIdx.size() / 3;
    if (texcoords && nt) memcpy(texcoords + 2 * tOff, m.texcoords.data(), m.texcoords.size() * sizeof(float));
    for (size_t f = 0; f < nf; f++) {
      const bool hasT = nt > 0 && m.tIdx[3 * f] >= 0 && m.tIdx[3 * f + 1] >= 0 && m.tIdx[3 * f + 2] >= 0;
      for (int k = 0; k < 3; k++) if (tIdx) tIdx[3 * (fOff + f) + k] = hasT ? m.tIdx[3 * f + k] + (int32_t)tOff : -1;
    }
    tOff += nt; fOff += nf;
  }
  return MOPTIX_OK;
}

int mohost_scene_texture(mohost_scene s, int32_t i, int32_t* width, int32_t* height, float* rgba) {
  if (!s || i < 0 || i >= (int32_t)s->desc.textures.size()) { g_err = "no such texture"; return MOPTIX_ERR_INVALID; }
  const TextureDesc& t = s->desc.textures[i];
  if (width) *width = t.width;
  if (height) *height = t.height;
  if (rgba) memcpy(rgba, t.rgba.data(), t.rgba.size() * sizeof(float));
  return MOPTIX_OK;
}

int mohost_read_image(const char* path, int32_t* width, int32_t* height, uint8_t* rgb, uint64_t rgbCapacity) {
  int w = 0, h = 0; std::vector<uint8_t> px; std::string err;
  if (!path || !readImage(path, w, h, px, err)) { g_err = path ? err : "null path"; return MOPTIX_ERR_INVALID; }
  if (width) *width = w;
  if (height) *height = h;
  if (rgb) {
    if (rgbCapacity < px.size()) { g_err = "image buffer too small"; return MOPTIX_ERR_INVALID; }
    memcpy(rgb, px.data(), px.size());
  }
  return MOPTIX_OK;
}

int mohost_scene_upload(mohost_scene s, moptix_context ctx) {
  if (!s || !ctx) return MOPTIX_ERR_INVALID;
  return upload(s->desc, ctx);
}

void mohost_set_quad_params(const float a[3], const float v1[3], const float v2[3], moptix_quad_params* out) {
  setQuadParams(pt::mk3(a[0], a[1], a[2]), pt::mk3(v1[0], v1[1], v1[2]), pt::mk3(v2[0], v2[1], v2[2]), *out);
}
void mohost_set_cam_params(const float f[3], const float a[3], const float u[3], float vFoV, float aspect,
                           float aperture, float focus, moptix_cam_params* out) {
  setCamParams(pt::mk3(f[0], f[1], f[2]), pt::mk3(a[0], a[1], a[2]), pt::mk3(u[0], u[1], u[2]), vFoV, aspect, aperture, focus, *out);
}

void mohost_animate_spheres(moptix_sphere_params* spheres, int32_t n, float time, float* angle) {
  VideoParams vp;
  vp.angle = angle ? *angle : 0.f;
  vp.spheresParams.assign(spheres, spheres + n);
  animateSpheres(vp, time);
  for (int32_t i = 0; i < n; i++) spheres[i] = vp.spheresParams[i];
  if (angle) *angle = vp.angle;
}
void mohost_video_camera(float angle, float aspect, moptix_cam_params* out) {
  VideoParams vp; vp.angle = angle;
  videoCamera(vp, aspect, *out);
}

int mohost_obj_stats(const char* path, int32_t* nVerts, int32_t* nNormals, int32_t* nTexcoords, int32_t* nShapes) {
  mobj::attrib_t attrib; std::vector<mobj::shape_t> shapes; std::vector<mobj::material_t> mats; std::string warn, err;
  if (!mobj::LoadObj(&attrib, &shapes, &mats, &warn, &err, path)) { g_err = err; return -1; }
  if (nVerts) *nVerts = (int32_t)(attrib.vertices.size() / 3);
  if (nNormals) *nNormals = (int32_t)(attrib.normals.size() / 3);
  if (nTexcoords) *nTexcoords = (int32_t)(attrib.texcoords.size() / 2);
  if (nShapes) *nShapes = (int32_t)shapes.size();
  int faces = 0;
  for (auto& sh : shapes) faces += (int)sh.mesh.num_face_vertices.size();
  return faces;
}

int mohost_render_scene(int device, int sceneId, const char* baseFolder, uint32_t width, uint32_t height,
                        uint32_t nSuperSampling, uint32_t baseSeed, int autoSave, const char* fileNamePrefix,
                        const char* outputDir, uint8_t* canvasRGB8, mohost_render_result* result) {
  try {
    MinimalOptiX app(device);
    app.verbose = false;
    app.fixedWidth = width; app.fixedHeight = height; app.nSuperSampling = nSuperSampling;
    app.baseSeed = baseSeed; app.sceneId = (MinimalOptiX::SceneId)sceneId;
    if (baseFolder) app.baseSceneFolder = baseFolder;
    if (outputDir) app.outputDir = outputDir;
    app.setupContext();
    app.renderScene(autoSave != 0, fileNamePrefix ? fileNamePrefix : "");
    if (canvasRGB8) memcpy(canvasRGB8, app.canvas.data(), app.canvas.size());
    if (result) {
      result->renderMs = app.lastRenderMs; result->bvhBuildMs = app.lastAccel.buildMs;
      result->nVertices = app.nVertices; result->nFaces = app.nFaces;
      result->nNodes = app.lastAccel.nNodes; result->treeDepth = app.lastAccel.treeDepth;
    }
    return MOPTIX_OK;
  } catch (const std::exception& e) { g_err = e.what(); return MOPTIX_ERR_INVALID; }
}
