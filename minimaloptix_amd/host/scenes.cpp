// scenes.cpp -- scene construction: the host half of the render path that the reference keeps
// in MinimalOptiX::setupScene()/setupScene(name)/setUpVideo() + utils_host.cpp helpers.
#include "scene_desc.h"

#include <cmath>
#include <cstring>
#include <iostream>
#include <random>
#include <stdexcept>

#include "image_io.h"
#include "obj_loader.h"
#include "scene_file.h"

using pt::v3;
using pt::mk3;

namespace moptix {

namespace {
inline moptix_float3 f3(const v3& a) { return { a.x, a.y, a.z }; }
inline v3 tv(const moptix_float3& a) { return mk3(a.x, a.y, a.z); }
}

// utils_host.cpp:67-75
void setQuadParams(const v3& anchor, const v3& v1, const v3& v2, moptix_quad_params& q) {
  const v3 normal = pt::normalize(pt::cross(v2, v1));
  const float d = pt::dot(normal, anchor);
  q.plane = { normal.x, normal.y, normal.z, d };
  q.v1 = f3(v1 / pt::dot(v1, v1));
  q.v2 = f3(v2 / pt::dot(v2, v2));
  q.anchor = f3(anchor);
}

// utils_host.cpp:77-99
void setCamParams(const v3& lookFrom, const v3& lookAt, const v3& up,
                  float vFoV, float aspect, float aperture, float focus, moptix_cam_params& cam) {
  static const float pi = 3.141592653589793238462643383279502884f;
  const float theta = vFoV * pi / 180;
  const float halfHeight = tanf(theta / 2);
  const float halfWidth = aspect * halfHeight;
  const v3 w = pt::normalize(lookFrom - lookAt);
  const v3 u = pt::normalize(pt::cross(up, w));
  const v3 v = pt::cross(w, u);
  const v3 scrLowerLeftCorner = ((lookFrom - u * (focus * halfWidth)) - v * (focus * halfHeight)) - w * focus;
  const v3 horizontal = u * (2 * focus * halfWidth);
  const v3 vertical = v * (2 * focus * halfHeight);
  cam.origin = f3(lookFrom); cam.horizontal = f3(horizontal); cam.vertical = f3(vertical);
  cam.scrLowerLeftCorner = f3(scrLowerLeftCorner); cam.u = f3(u); cam.v = f3(v);
  cam.lensRadius = aperture / 2;
}

void defaultParams(moptix_params& p, uint32_t width, uint32_t height) {
  memset(&p, 0, sizeof(p));
  p.width = width; p.height = height;
  p.rayMaxDepth = 256u;          // MinimalOptiX.h:85
  p.rayMinIntensity = 0.001f;    // MinimalOptiX.h:88
  p.rayEpsilonT = 0.001f;        // MinimalOptiX.h:89
}

moptix_material lambertianMaterial(float r, float g, float b) {
  moptix_material m; memset(&m, 0, sizeof(m)); m.kind = MOPTIX_MAT_LAMBERTIAN; m.albedo = { r, g, b }; return m;
}
moptix_material metalMaterial(float r, float g, float b, float fuzz) {
  moptix_material m; memset(&m, 0, sizeof(m)); m.kind = MOPTIX_MAT_METAL; m.albedo = { r, g, b }; m.fuzz = fuzz; return m;
}
moptix_material glassMaterial(float r, float g, float b, float refIdx) {
  moptix_material m; memset(&m, 0, sizeof(m)); m.kind = MOPTIX_MAT_GLASS; m.albedo = { r, g, b }; m.refIdx = refIdx; return m;
}
moptix_material lightMaterial(float r, float g, float b) {
  moptix_material m; memset(&m, 0, sizeof(m)); m.kind = MOPTIX_MAT_LIGHT; m.emission = { r, g, b }; return m;
}
moptix_material disneyMaterial(const moptix_disney_params& p) {
  moptix_material m; memset(&m, 0, sizeof(m)); m.kind = MOPTIX_MAT_DISNEY; m.disney = p; return m;
}

static void addQuad(SceneDesc& s, const v3& anchor, const v3& v1, const v3& v2, int mat) {
  moptix_quad_params q;
  setQuadParams(anchor, v1, v2, q);
  s.quads.push_back(q); s.quadMat.push_back(mat);
}
static void addSphere(SceneDesc& s, float radius, const v3& c, int mat) {
  moptix_sphere_params sp; memset(&sp, 0, sizeof(sp));
  sp.radius = radius; sp.center = f3(c);
  s.spheres.push_back(sp); s.sphereMat.push_back(mat);
}

// ---------------------------------------------------------------- SCENE_SPHERES -----------
void buildSpheresScene(SceneDesc& s, uint32_t width, uint32_t height, float aperture) {
  s = SceneDesc();
  s.name = "spheres";
  defaultParams(s.params, width, height);
  s.params.bgColor = { 0.5f, 0.5f, 0.5f };                                   // :165
  s.accel = "NoAccel";                                                       // :248
  addSphere(s, 0.5f, mk3(0.f, 0.f, -1.f), s.addMaterial(lambertianMaterial(0.1f, 0.2f, 0.5f)));   // :157-186
  addSphere(s, 0.5f, mk3(1.f, 0.f, -1.f), s.addMaterial(metalMaterial(0.8f, 0.6f, 0.2f, 0.f)));    // :188-197
  addSphere(s, 0.5f, mk3(-1.f, 0.f, -1.f), s.addMaterial(glassMaterial(1.f, 1.f, 1.f, 1.5f)));     // :199-208
  addQuad(s, mk3(-1000.f, -0.5f, -1000.f), mk3(2000.f, 0.f, 0.f), mk3(0.f, 0.f, 2000.f),
          s.addMaterial(lambertianMaterial(0.8f, 0.8f, 0.f)));                                    // :210-224
  addQuad(s, mk3(-5.f, 5.f, 5.f), mk3(0.f, 0.f, -10.f), mk3(10.f, 0.f, 0.f),
          s.addMaterial(lightMaterial(1.f, 1.f, 1.f)));                                           // :226-240
  const v3 lookFrom = mk3(3.f, 3.f, 2.f), lookAt = mk3(0.f, 0.f, -1.f), up = mk3(0.f, 1.f, 0.f);  // :250-254
  setCamParams(lookFrom, lookAt, up, 20, (float)width / (float)height, aperture, pt::length(lookFrom - lookAt), s.params.cam);
}

// ---------------------------------------------------------------- file scenes -------------
static void loadFileScene(SceneDesc& s, const std::string& baseSceneFolder, const std::string& sceneName, bool skipMissing) {
  s.nVertices = 0; s.nFaces = 0;
  s.accel = "Trbvh";                                                         // :378,494,534
  const std::string sceneFolder = baseSceneFolder + sceneName + "/";
  Scene scene((sceneFolder + sceneName + ".scene").c_str());                  // :373-374

  size_t matCursor = 0;
  for (size_t i = 0; i < scene.meshNames.size(); ++i) {                       // :379-490
    mobj::attrib_t attrib;
    std::vector<mobj::shape_t> shapes;
    std::vector<mobj::material_t> materials;
    std::string warn, err;
    const bool ret = mobj::LoadObj(&attrib, &shapes, &materials, &warn, &err, (sceneFolder + scene.meshNames[i]).c_str());
    const size_t matIndex = matCursor++;                                      // materials[i] pairs with meshNames[i]
    if (!err.empty() || !ret) {
      if (skipMissing) { s.warnings.push_back("skipped mesh " + scene.meshNames[i] + ": " + err); continue; }
      std::cerr << err << std::endl;
      throw std::logic_error("Cannot load mesh file.");                       // :386-389
    }
    if (matIndex >= scene.materials.size()) throw std::logic_error("mesh entry without material: " + scene.meshNames[i]);
    if (!scene.textures[matIndex].empty()) {                                  // :445-479 one sampler per file name
      const std::string& texName = scene.textures[matIndex];
      int texId = 0;
      for (size_t t = 0; t < s.textures.size(); t++) if (s.textures[t].name == texName) texId = (int)t + 1;
      if (texId == 0) {
        int tw = 0, th = 0; std::vector<uint8_t> rgb; std::string ierr;
        if (readImage(sceneFolder + texName, tw, th, rgb, ierr)) {
          TextureDesc td; td.name = texName; td.width = tw; td.height = th;
          imageToTextureRGBA(rgb, tw, th, td.rgba);
          s.textures.push_back(std::move(td));
          texId = (int)s.textures.size();
        } else {
          // QImage yields a null image here and the reference goes on to create a 0x0 buffer; we keep the
          // material's constant colour instead and say so
          s.warnings.push_back("albedoTex ignored: " + ierr);
        }
      }
      scene.materials[matIndex].albedoID = texId;
    }
    const int matId = s.addMaterial(disneyMaterial(scene.materials[matIndex]));   // :482-485
    for (size_t sh = 0; sh < shapes.size(); sh++) {                           // :390-442 one Geometry per shape
      MeshDesc m;
      m.source = scene.meshNames[i];
      m.matId = matId;
      m.positions = attrib.vertices; m.normals = attrib.normals; m.texcoords = attrib.texcoords;
      s.nVertices += attrib.vertices.size() / 3;
      const size_t nf = shapes[sh].mesh.num_face_vertices.size();
      m.vIdx.resize(3 * nf); m.nIdx.resize(3 * nf); m.tIdx.resize(3 * nf);
      for (size_t f = 0; f < nf; f++) {
        for (int fv = 0; fv < 3; ++fv) {
          const mobj::index_t& idx = shapes[sh].mesh.indices[f * 3 + fv];
          // the reference reads attrib.vertices[3 * index] unchecked (:430-433); an index past the file's last "v" is an error here
          if (idx.vertex_index < 0 || (size_t)idx.vertex_index >= attrib.vertices.size() / 3)
            throw std::logic_error("face references vertex " + std::to_string(idx.vertex_index + 1) + " of " +
                                   std::to_string(attrib.vertices.size() / 3) + " in " + scene.meshNames[i]);
          m.vIdx[f * 3 + fv] = idx.vertex_index;
          m.tIdx[f * 3 + fv] = idx.texcoord_index;
          m.nIdx[f * 3 + fv] = idx.normal_index;
          s.aabb.include(mk3(attrib.vertices[3 * idx.vertex_index + 0], attrib.vertices[3 * idx.vertex_index + 1],
                             attrib.vertices[3 * idx.vertex_index + 2]));       // :430-433
        }
      }
      s.nFaces += nf;
      s.meshes.push_back(std::move(m));
    }
  }

  for (const moptix_light_params& light : scene.lights) {                     // :493-521
    if (light.shape == MOPTIX_LIGHT_SPHERE) {
      // NOTE: the reference's sphereBBox hands OptiX an inverted box, so under
      // Trbvh a sphere light is probably never hit by radiance rays; here it is hit correctly.
      addSphere(s, light.radius, tv(light.position), s.addMaterial(lightMaterial(light.emission.x, light.emission.y, light.emission.z)));
    } else if (light.shape == MOPTIX_LIGHT_QUAD) {
      addQuad(s, tv(light.position), tv(light.u), tv(light.v),
              s.addMaterial(lightMaterial(light.emission.x, light.emission.y, light.emission.z)));
    } else {
      throw std::logic_error("No shape for light.");                          // :512
    }
    s.lights.push_back(light);                                                // :523-531
  }
}

void buildFileScene(SceneDesc& s, const std::string& baseSceneFolder, const std::string& sceneName,
                    uint32_t width, uint32_t height, bool skipMissing) {
  s = SceneDesc();
  s.name = sceneName;
  defaultParams(s.params, width, height);
  const float aspect = (float)width / (float)height;
  const v3 up = mk3(0.f, 1.f, 0.f);
  v3 lookFrom, lookAt; float fov = 45.f;
  // the asset folder of both SCENE_HYPERION and SCENE_DRAGON is "hyperion" (:340)
  const std::string folder = (sceneName == "dragon") ? "hyperion" : sceneName;
  loadFileScene(s, baseSceneFolder, folder, skipMissing);
  const Aabb& aabb = s.aabb;
  if (sceneName == "coffee") {                                                // :258-270
    s.params.bgColor = { 0.f, 0.f, 0.f };
    lookFrom = mk3(0.f, (float)(0.22 * aabb.extent(1)), (float)(0.25 * aabb.extent(2)));
    lookAt = lookFrom + mk3(0.f, -0.01875f, -1.f);
  } else if (sceneName == "bedroom") {                                        // :271-283
    s.params.bgColor = { 0.f, 0.f, 0.f };
    lookFrom = aabb.center() + mk3(0.3f, 0.1f, 0.45f) * aabb.extent();
    lookAt = aabb.center() + mk3(0.05f, -0.1f, 0.f) * aabb.extent();
  } else if (sceneName == "diningroom") {                                     // :284-296
    s.params.bgColor = { 0.f, 0.f, 0.f };
    lookFrom = aabb.center() + mk3(-0.7f, 0.f, 0.f) * aabb.extent();
    lookAt = aabb.center() + mk3(0.f, 0.f, 0.f) * aabb.extent();
  } else if (sceneName == "stormtrooper") {                                   // :297-309
    s.params.bgColor = { 0.5f, 0.5f, 0.5f };
    lookFrom = aabb.center() + mk3(0.25f, 0.1f, 0.395f) * aabb.extent();
    lookAt = aabb.center() + mk3(0.25f, 0.1f, 0.f) * aabb.extent();
    fov = 30.f;
  } else if (sceneName == "spaceship") {                                      // :310-322
    s.params.bgColor = { 0.5f, 0.5f, 0.5f };
    lookFrom = aabb.center() + mk3(-0.03f, 0.03f, -0.03f) * aabb.extent();
    lookAt = aabb.center() + mk3(0.f, 0.f, 0.f) * aabb.extent();
  } else if (sceneName == "cornell") {                                        // :323-335
    s.params.bgColor = { 0.5f, 0.5f, 0.5f };
    lookFrom = aabb.center() + mk3(0.f, 0.f, -2.f) * aabb.extent();
    lookAt = aabb.center() + mk3(0.f, 0.f, 0.f) * aabb.extent();
    fov = (float)39.3077;
  } else if (sceneName == "hyperion" || sceneName == "dragon") {              // :336-353
    s.params.bgColor = { 0.5f, 0.5f, 0.5f };
    lookFrom = (sceneName == "hyperion") ? aabb.center() + mk3(-0.08f, 2.f, 0.f) * aabb.extent()
                                         : aabb.center() + mk3(0.05f, 0.3f, -0.005f) * aabb.extent();
    lookAt = aabb.center() + mk3(0.f, 0.f, 0.f) * aabb.extent();
    fov = 30.f;
  } else {
    throw std::logic_error("unknown file scene: " + sceneName);
  }
  setCamParams(lookFrom, lookAt, up, fov, aspect, 0.f, 1.f, s.params.cam);
}

// ---------------------------------------------------------------- SCENE_SPHERES_VIDEO -----
// The reference draws the layout from std::mt19937(42) through std::normal/uniform
// distributions, whose output is implementation-defined.  The engine is kept
// (its sequence is fixed by the standard); the distributions are defined here:
//   uniform()      = (mt() >> 8) * 2^-24                       in [0,1)
//   uniform_int()  = mt() % 3                                  in {0,1,2}
//   normal(0,.1)   = .1 * sqrt(-2 ln u1) * cos(2 pi u2),  u1 = ((mt()>>8)+1) * 2^-24, u2 = uniform()
namespace {
struct LayoutRng {
  std::mt19937 mt{ 42 };
  float uniform() { return (float)(mt() >> 8) * (1.0f / 16777216.0f); }
  int uniform_int() { return (int)(mt() % 3u); }
  float normal01() {
    const float u1 = (float)((mt() >> 8) + 1u) * (1.0f / 16777216.0f);
    const float u2 = uniform();
    return 0.1f * sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
  }
};
}

void buildRandomSpheresScene(SceneDesc& s, int nSpheres, uint32_t width, uint32_t height) {
  s = SceneDesc();
  s.name = "random_spheres";
  defaultParams(s.params, width, height);
  s.params.bgColor = { 0.2f, 0.2f, 0.2f };                                    // :611
  s.accel = "NoAccel";                                                        // :748
  LayoutRng rng;
  std::vector<moptix_sphere_params> sp;
  auto push = [&](float r, float x, float y, float z) {
    moptix_sphere_params p; memset(&p, 0, sizeof(p)); p.radius = r; p.center = { x, y, z }; sp.push_back(p);
  };
  for (int i = 0; i < 3; ++i) push(3.0f, -10.f + 10.f * i, 2.0f, 0.f);        // :629-631
  for (int i = 0; i < nSpheres; ++i) {                                        // :632-646
    float x, z, radius;
    do {
      x = rng.uniform() * 30.f - 15.f;
      z = rng.uniform() * 30.f - 15.f;
      radius = 1.0f;
      for (auto& param : sp)
        radius = std::min(radius, sqrtf((x - param.center.x) * (x - param.center.x) + (z - param.center.z) * (z - param.center.z)) - param.radius);
      radius = (float)(radius * 0.8);
    } while (radius < .01f);
    const float h = sqrtf(x * x + z * z);
    radius = std::min(h + .5f, radius);
    push(radius, x, h, z);
  }
  // materials of the three big spheres (:647-671, useDisney == false)
  s.spheres = sp;
  s.sphereMat.resize(sp.size());
  s.sphereMat[0] = s.addMaterial(lambertianMaterial(0.5f, 0.8f, 0.8f));
  s.sphereMat[1] = s.addMaterial(glassMaterial(1.f, 1.f, 1.f, 1.5f));
  {
    float tmp = (float)(rng.normal01() + 0.5);
    tmp = std::min(0.9f, std::max(0.1f, tmp));
    s.sphereMat[2] = s.addMaterial(metalMaterial(0.9f, 0.7f, 0.7f, tmp));
  }
  for (size_t i = 3; i < sp.size(); ++i) {                                    // :672-699
    const float cr = 0.2f + 0.8f * rng.uniform(), cg = 0.2f + 0.8f * rng.uniform(), cb = 0.2f + 0.8f * rng.uniform();
    const int type = rng.uniform_int();
    if (type == 0) {
      s.sphereMat[i] = s.addMaterial(lambertianMaterial(cr, cg, cb));
    } else if (type == 1) {
      float tmp = (float)(rng.normal01() + 0.5);
      tmp = std::min(0.9f, std::max(0.1f, tmp));
      s.sphereMat[i] = s.addMaterial(metalMaterial(cr, cg, cb, tmp));
    } else {
      float tmp = (float)(rng.normal01() + 2.0);
      tmp = std::min(3.0f, std::max(1.5f, tmp));
      s.sphereMat[i] = s.addMaterial(glassMaterial(1.f, 1.f, 1.f, tmp));
    }
  }
  addQuad(s, mk3(-100.f, -0.5f, 100.f), mk3(0.f, 0.f, -200.f), mk3(200.f, 0.f, 0.f),
          s.addMaterial(lambertianMaterial(0.7f, 0.9f, 0.9f)));               // :701-715
  const int lightMat = s.addMaterial(lightMaterial(1.f, 1.f, 1.f));           // buildLight :780-794
  for (int i = 0; i < 4; ++i)                                                 // :718-722
    for (int j = 0; j < 4; ++j)
      addQuad(s, mk3(-24.f + 10.f * i, 15.f, -24.f + 10.f * j), mk3(0.f, 0.f, -8.f), mk3(8.f, 0.f, 0.f), lightMat);
  constexpr int nLight = 16;                                                  // :723-728
  constexpr float angle = (float)(3.1415926 * 2 / nLight);
  for (int i = 0; i < nLight; ++i)
    addQuad(s, mk3(40.f * sinf(i * angle), 1.f, 40.f * cosf(i * angle)), mk3(0.f, 4.f, 0.f),
            mk3(10.f * sinf(i * angle + angle) - 10.f * sinf(i * angle), 0.f, 10.f * cosf(i * angle + angle) - 10.f * cosf(i * angle)), lightMat);
  // the `lights` buffer the reference fills here holds garbage and is never read; omitted.
  setCamParams(mk3(0.f, 8.0f, 20.f), mk3(0.f, 0.f, 0.f), mk3(0.f, 1.f, 0.f), 45, (float)width / (float)height, .2f, 20.f, s.params.cam);  // :751-754
}

// ---------------------------------------------------------------- Cornell (authored) ------
void buildCornellQuadsScene(SceneDesc& s, uint32_t width, uint32_t height) {
  s = SceneDesc();
  s.name = "cornell_quads";
  defaultParams(s.params, width, height);
  s.params.bgColor = { 0.5f, 0.5f, 0.5f };                                    // :326
  s.accel = "NoAccel";
  const int white = s.addMaterial(lambertianMaterial(.73f, .73f, .73f));
  const int red = s.addMaterial(lambertianMaterial(.65f, .05f, .05f));
  const int green = s.addMaterial(lambertianMaterial(.12f, .45f, .15f));
  const int light = s.addMaterial(lightMaterial(15.f, 15.f, 15.f));
  Aabb box;
  auto quad = [&](v3 a, v3 b, v3 d, int mat) {      // parallelogram a, b, (b+d-a), d
    addQuad(s, a, b - a, d - a, mat);
    box.include(a); box.include(b); box.include(d); box.include(b + (d - a));
  };
  quad(mk3(0, 0, 0), mk3(555, 0, 0), mk3(0, 0, 555), white);        // floor
  quad(mk3(0, 555, 0), mk3(555, 555, 0), mk3(0, 555, 555), white);  // ceiling
  quad(mk3(0, 0, 555), mk3(555, 0, 555), mk3(0, 555, 555), white);  // back wall
  quad(mk3(555, 0, 0), mk3(555, 0, 555), mk3(555, 555, 0), red);    // left (image space)
  quad(mk3(0, 0, 0), mk3(0, 0, 555), mk3(0, 555, 0), green);        // right
  quad(mk3(213, 554, 227), mk3(343, 554, 227), mk3(213, 554, 332), light);
  auto block = [&](const v3 top[4], float h) {
    quad(top[0], top[1], top[3], white);
    for (int k = 0; k < 4; k++) {
      const v3 a = top[k], b = top[(k + 1) & 3];
      quad(mk3(a.x, 0, a.z), mk3(b.x, 0, b.z), mk3(a.x, h, a.z), white);
    }
  };
  const v3 shortTop[4] = { mk3(130, 165, 65), mk3(82, 165, 225), mk3(240, 165, 272), mk3(290, 165, 114) };
  const v3 tallTop[4] = { mk3(423, 330, 247), mk3(265, 330, 296), mk3(314, 330, 456), mk3(472, 330, 406) };
  block(shortTop, 165.f);
  block(tallTop, 330.f);
  s.aabb = box;
  const v3 lookFrom = box.center() + mk3(0.f, 0.f, -2.f) * box.extent();     // :328-332
  const v3 lookAt = box.center();
  setCamParams(lookFrom, lookAt, mk3(0.f, 1.f, 0.f), (float)39.3077, (float)width / (float)height, 0.f, 1.f, s.params.cam);
}

// ---------------------------------------------------------------- animation ---------------
// MinimalOptiX.cpp:562-585
void moveSphere(const VideoParams& vp, moptix_sphere_params& param, float time) {
  float distance = param.velocity.y * time + time * time * vp.gravity / 2.0f;
  if (distance < param.center.y - param.radius + 0.5f) {   // -0.5f is the plane
    param.center.x += param.velocity.x * time;
    param.center.z += param.velocity.z * time;
    param.center.y -= distance;
    param.velocity.y += vp.gravity * time;
  } else {
    // Divergence D7: for a sphere that already pokes through the plane (the three r=3 spheres at y=2 do)
    // the radicand is negative; the reference then recurses forever on NaN.  Clamped to 0 here, which
    // snaps such a sphere onto the plane at rest.
    float vend = sqrtf(fmaxf(0.0f, param.velocity.y * param.velocity.y + (2.0f * vp.gravity * (param.center.y - param.radius + 0.5f))));
    float t = (vend - param.velocity.y) / vp.gravity;
    if (t < 1e-6) {
      param.velocity.y = 0.f;
      param.center.y = -0.5f + param.radius;
      return;
    }
    param.center.x += param.velocity.x * t;
    param.center.z += param.velocity.z * t;
    param.center.y = -0.5f + param.radius;
    param.velocity.x *= vp.attenuationCoef;
    param.velocity.y *= vp.attenuationCoef;
    param.velocity.y = -vend * vp.attenuationCoef;
    moveSphere(vp, param, time - t);
  }
}
// MinimalOptiX.cpp:587-592
void animateSpheres(VideoParams& vp, float time) {
  vp.angle += time * 5;
  for (size_t i = 0; i < vp.spheresParams.size(); ++i) moveSphere(vp, vp.spheresParams[i], time);
}
// MinimalOptiX.cpp:766-767
void videoCamera(const VideoParams& vp, float aspect, moptix_cam_params& cam) {
  const v3 lookFrom = mk3(20 * sinf(vp.angle), (float)std::min(12.0, vp.angle / 10 + 8.0), 20.f * cosf(vp.angle));
  setCamParams(lookFrom, vp.lookAt, vp.up, 45, aspect, .2f, 20.f, cam);
}

// ---------------------------------------------------------------- upload ------------------
int upload(const SceneDesc& s, moptix_context ctx) {
  int rc;
  if ((rc = moptix_clear_scene(ctx)) != MOPTIX_OK) return rc;
  if ((rc = moptix_set_params(ctx, &s.params)) != MOPTIX_OK) return rc;
  for (const TextureDesc& t : s.textures) {
    int32_t id = 0;
    if ((rc = moptix_add_texture(ctx, t.rgba.data(), t.width, t.height, &id)) != MOPTIX_OK) return rc;
  }
  for (const moptix_material& m : s.materials) {
    int32_t id = -1;
    if ((rc = moptix_add_material(ctx, &m, &id)) != MOPTIX_OK) return rc;
  }
  if (!s.spheres.empty() && (rc = moptix_add_spheres(ctx, s.spheres.data(), s.sphereMat.data(), (int32_t)s.spheres.size())) != MOPTIX_OK) return rc;
  if (!s.quads.empty() && (rc = moptix_add_quads(ctx, s.quads.data(), s.quadMat.data(), (int32_t)s.quads.size())) != MOPTIX_OK) return rc;
  for (const MeshDesc& m : s.meshes) {
    rc = moptix_add_mesh(ctx, m.positions.data(), (int32_t)(m.positions.size() / 3),
                         m.normals.empty() ? nullptr : m.normals.data(), (int32_t)(m.normals.size() / 3),
                         m.texcoords.empty() ? nullptr : m.texcoords.data(), (int32_t)(m.texcoords.size() / 2),
                         m.vIdx.data(), m.nIdx.data(), m.tIdx.data(), (int32_t)(m.vIdx.size() / 3), m.matId);
    if (rc != MOPTIX_OK) return rc;
  }
  if ((rc = moptix_set_lights(ctx, s.lights.empty() ? nullptr : s.lights.data(), (int32_t)s.lights.size())) != MOPTIX_OK) return rc;
  return moptix_build_accel(ctx, s.accel.c_str());
}

}  // namespace moptix
