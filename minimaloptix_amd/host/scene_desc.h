// scene_desc.h -- host-side description of one renderable scene: what the reference's
// setupScene()/setupScene(name)/setUpVideo() (MinimalOptiX.cpp:154-538, 607-759) hand to
// the OptiX context, flattened into plain arrays that are then pushed through the C ABI
// of include/moptix.h (upload()).
#pragma once
#include <string>
#include <vector>
#include "../../include/moptix.h"
#include "../csrc/pt_math.h"

namespace moptix {

// optix::Aabb
struct Aabb {
  pt::v3 m_min, m_max;
  Aabb() { invalidate(); }
  void invalidate() { m_min = pt::mk3(1e37f, 1e37f, 1e37f); m_max = pt::mk3(-1e37f, -1e37f, -1e37f); }
  void include(pt::v3 p) {
    m_min = pt::mk3(fminf(m_min.x, p.x), fminf(m_min.y, p.y), fminf(m_min.z, p.z));
    m_max = pt::mk3(fmaxf(m_max.x, p.x), fmaxf(m_max.y, p.y), fmaxf(m_max.z, p.z));
  }
  pt::v3 extent() const { return m_max - m_min; }
  float extent(int d) const { return d == 0 ? m_max.x - m_min.x : d == 1 ? m_max.y - m_min.y : m_max.z - m_min.z; }
  pt::v3 center() const { return (m_min + m_max) * 0.5f; }
};

struct MeshDesc {                 // one tinyobj shape (MinimalOptiX.cpp:392-441)
  std::vector<float> positions, normals, texcoords;
  std::vector<int32_t> vIdx, nIdx, tIdx;   // 3 per face
  int32_t matId = 0;
  std::string source;
};

struct TextureDesc {              // one TextureSampler + its float4 buffer (MinimalOptiX.cpp:444-479)
  std::string name;               // file name relative to the scene folder (texNameSamplerMap key)
  int32_t width = 0, height = 0;
  std::vector<float> rgba;        // 4*width*height, row 0 = bottom image row
};

struct SceneDesc {
  std::string name;
  moptix_params params{};         // W,H, depth, eps, bgColor, camParams
  std::string accel = "NoAccel";  // "NoAccel" | "Trbvh"
  std::vector<moptix_material> materials;
  std::vector<moptix_sphere_params> spheres; std::vector<int32_t> sphereMat;
  std::vector<moptix_quad_params> quads;     std::vector<int32_t> quadMat;
  std::vector<moptix_light_params> lights;   // NEE light list (context["lights"])
  std::vector<MeshDesc> meshes;
  std::vector<TextureDesc> textures;         // DisneyParams.albedoID = index + 1
  Aabb aabb;
  size_t nVertices = 0, nFaces = 0;          // MinimalOptiX.h:86-87
  std::vector<std::string> warnings;

  int addMaterial(const moptix_material& m) { materials.push_back(m); return (int)materials.size() - 1; }
};

// utils_host.cpp:67-75 / :77-99
void setQuadParams(const pt::v3& anchor, const pt::v3& v1, const pt::v3& v2, moptix_quad_params& quadParams);
void setCamParams(const pt::v3& lookFrom, const pt::v3& lookAt, const pt::v3& up,
                  float vFoV, float aspect, float aperture, float focus, moptix_cam_params& camParams);
// context defaults (MinimalOptiX.h:82-89, MinimalOptiX.cpp:136-151)
void defaultParams(moptix_params& p, uint32_t width, uint32_t height);

moptix_material lambertianMaterial(float r, float g, float b);
moptix_material metalMaterial(float r, float g, float b, float fuzz);
moptix_material glassMaterial(float r, float g, float b, float refIdx);
moptix_material lightMaterial(float r, float g, float b);
moptix_material disneyMaterial(const moptix_disney_params& p);

// ---- scene builders.  `width`/`height` replace the reference's fixedWidth/fixedHeight. ----
// SCENE_SPHERES, MinimalOptiX.cpp:156-257 (aperture 0.5 as committed; 0 = demo/spheres_pinhole.png)
void buildSpheresScene(SceneDesc& s, uint32_t width, uint32_t height, float aperture = 0.5f);
// setupScene(const char*) + the per-scene camera of MinimalOptiX.cpp:258-353.
// sceneName: coffee|bedroom|diningroom|stormtrooper|spaceship|cornell|hyperion|dragon
// skipMissing: a mesh file that cannot be opened is skipped with a warning instead of
// throwing std::logic_error("Cannot load mesh file.") (MinimalOptiX.cpp:386-389).
void buildFileScene(SceneDesc& s, const std::string& baseSceneFolder, const std::string& sceneName,
                    uint32_t width, uint32_t height, bool skipMissing);
// SCENE_SPHERES_VIDEO frame 0, setUpVideo(nSpheres) MinimalOptiX.cpp:607-759
void buildRandomSpheresScene(SceneDesc& s, int nSpheres, uint32_t width, uint32_t height);
// BASELINE config 1: authored Cornell box of quads (asset absent in the reference)
void buildCornellQuadsScene(SceneDesc& s, uint32_t width, uint32_t height);
// BASELINE config 4 stand-in: box room + K transformed copies of the coffee meshes, Disney mix
void buildDiningStandInScene(SceneDesc& s, const std::string& baseSceneFolder, int copies, uint32_t width, uint32_t height);
// coffee scene + a lathe stand-in for its glass pot (Mesh010.obj is missing from the reference checkout)
void buildCoffeePotStandInScene(SceneDesc& s, const std::string& baseSceneFolder, uint32_t width, uint32_t height);
// BASELINE config 5 stand-in: ~nTris-triangle procedural displaced torus knot (glass) + floor + sphere light
void buildProceduralMillionScene(SceneDesc& s, int nTrisTarget, uint32_t width, uint32_t height);

// ---- animation of the spheres scene (MinimalOptiX.cpp:562-592, VideoParams MinimalOptiX.h:19-30) ----
struct VideoParams {
  const float gravity = 4000.f;
  const float attenuationCoef = 0.9f;
  float angle{ 0.0 };
  pt::v3 lookAt = pt::mk3(0.f, 0.f, 0.f);
  pt::v3 up = pt::mk3(0.f, 1.f, 0.f);
  std::vector<moptix_sphere_params> spheresParams;
};
// MinimalOptiX::move / animate: ballistic fall with bounces on the plane y = -0.5
void moveSphere(const VideoParams& vp, moptix_sphere_params& param, float time);
void animateSpheres(VideoParams& vp, float time);
// camera of updateVideo (MinimalOptiX.cpp:766-767): orbit of radius 20, rising with the angle
void videoCamera(const VideoParams& vp, float aspect, moptix_cam_params& cam);

// Push the description through the C ABI: clear_scene, set_params, add_material..., build_accel.
// Returns MOPTIX_OK or the failing call's error code.
int upload(const SceneDesc& s, moptix_context ctx);

}  // namespace moptix
