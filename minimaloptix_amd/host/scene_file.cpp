#include "scene_file.h"

#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <stdexcept>

#include "../csrc/pt_math.h"

void initDisneyParams(moptix_disney_params& d) {
  d.color = { 1.0f, 1.0f, 1.0f };
  d.emission = { 0.0f, 0.0f, 0.0f };
  d.metallic = 0.0f; d.subsurface = 0.0f; d.specular = 0.5f; d.roughness = 0.5f;
  d.specularTint = 0.0f; d.anisotropic = 0.0f; d.sheen = 0.0f; d.sheenTint = 0.5f;
  d.clearcoat = 0.0f; d.clearcoatGloss = 1.0f;
  d.brdfType = MOPTIX_BRDF_NORMAL;
  d.albedoID = 0;   // RT_TEXTURE_ID_NULL
}

namespace {

constexpr int kLineBytes = 2048;                 // the reference reads lines with fgets(…, 2048, …)

inline pt::v3 toV(const moptix_float3& f) { return pt::mk3(f.x, f.y, f.z); }
inline moptix_float3 toF(const pt::v3& a) { return { a.x, a.y, a.z }; }

// The grammar of a block, as data.  The reference (scene.cpp:37-51, 69-75, 98-99) runs EVERY sscanf of a block on EVERY line of
// it and ignores the results, so a line assigns to whichever keys it happens to match (" specular %f" also looks at a
// "specularTint" line and takes nothing from it), later lines overwrite earlier ones, and unknown lines are skipped silently.
// A key here is that sscanf format plus where its conversions land.
struct ScalarKey { const char* format; float moptix_disney_params::*field; };
struct VectorKey { const char* format; moptix_float3 moptix_disney_params::*field; };
const VectorKey kMaterialVectors[] = {
  { " color %f %f %f", &moptix_disney_params::color },
  { " emission %f %f %f", &moptix_disney_params::emission },
};
const ScalarKey kMaterialScalars[] = {
  { " metallic %f", &moptix_disney_params::metallic },         { " subsurface %f", &moptix_disney_params::subsurface },
  { " specular %f", &moptix_disney_params::specular },         { " specularTint %f", &moptix_disney_params::specularTint },
  { " roughness %f", &moptix_disney_params::roughness },       { " anisotropic %f", &moptix_disney_params::anisotropic },
  { " sheen %f", &moptix_disney_params::sheen },               { " sheenTint %f", &moptix_disney_params::sheenTint },
  { " clearcoat %f", &moptix_disney_params::clearcoat },       { " clearcoatGloss %f", &moptix_disney_params::clearcoatGloss },
};
struct LightVectorKey { const char* format; int which; };       // 0 position, 1 emission, 2 normal, 3 v1, 4 v2
const LightVectorKey kLightVectors[] = {
  { " position %f %f %f", 0 }, { " emission %f %f %f", 1 }, { " normal %f %f %f", 2 }, { " v1 %f %f %f", 3 }, { " v2 %f %f %f", 4 },
};

// Lines of a "{ … }" block: everything up to (not including) the first line with a closing brace goes to `each`.
// `line` is the caller's buffer on purpose: after the block it holds the closing line, and the block kinds that follow in
// the same pass of the outer loop are matched against THAT text (scene.cpp:59, 93, 104 test the stale `line`).
void forEachBlockLine(FILE* file, char* line, const std::function<void(const char*)>& each) {
  while (fgets(line, kLineBytes, file) && !strchr(line, '}')) each(line);
}

}  // namespace

Scene::Scene(const char* fileName) {
  FILE* file = fopen(fileName, "r");
  if (!file) throw std::runtime_error(std::string("Couldn't open ") + fileName + " for reading.");

  std::map<std::string, moptix_disney_params> materialByName;
  std::map<std::string, std::string> textureByName;
  char line[kLineBytes];

  // ---- block readers ---------------------------------------------------------------------------------------------
  auto readMaterial = [&](char* name) {                                  // scene.cpp:28-57; `name` may be renamed by a " name %s" line
    moptix_disney_params m;
    initDisneyParams(m);
    char texture[kLineBytes] = "";
    int brdf = m.brdfType;
    forEachBlockLine(file, line, [&](const char* l) {
      sscanf(l, " name %s", name);
      sscanf(l, kMaterialVectors[0].format, &(m.*kMaterialVectors[0].field).x, &(m.*kMaterialVectors[0].field).y, &(m.*kMaterialVectors[0].field).z);
      sscanf(l, " albedoTex %s", texture);
      sscanf(l, kMaterialVectors[1].format, &(m.*kMaterialVectors[1].field).x, &(m.*kMaterialVectors[1].field).y, &(m.*kMaterialVectors[1].field).z);
      for (const ScalarKey& k : kMaterialScalars) sscanf(l, k.format, &(m.*k.field));
      sscanf(l, " brdf %i", &brdf);
    });
    m.brdfType = brdf;
    m.albedoID = 0;                                                      // assigned when the texture is created (upload)
    materialByName[name] = m;
    textureByName[name] = texture;
  };
  auto readLight = [&]() {                                               // scene.cpp:59-91
    moptix_light_params light;
    memset(&light, 0, sizeof(light));
    moptix_float3 corner[2] = { { 0, 0, 0 }, { 0, 0, 0 } };
    char kind[20] = "None";
    forEachBlockLine(file, line, [&](const char* l) {
      moptix_float3* target[5] = { &light.position, &light.emission, &light.normal, &corner[0], &corner[1] };
      for (const LightVectorKey& k : kLightVectors) {
        if (k.which == 3) sscanf(l, " radius %f", &light.radius);        // the reference's order: position emission normal radius v1 v2 type
        sscanf(l, k.format, &target[k.which]->x, &target[k.which]->y, &target[k.which]->z);
      }
      sscanf(l, " type %19s", kind);
    });
    if (!strcmp(kind, "Quad")) {
      const pt::v3 u = toV(corner[0]) - toV(light.position), w = toV(corner[1]) - toV(light.position);
      light.shape = MOPTIX_LIGHT_QUAD;
      light.u = toF(u); light.v = toF(w);
      light.area = pt::length(pt::cross(u, w));
      light.normal = toF(pt::normalize(pt::cross(u, w)));
    } else if (!strcmp(kind, "Sphere")) {
      light.shape = MOPTIX_LIGHT_SPHERE;
      light.normal = toF(pt::normalize(toV(light.normal)));
      light.area = 4.0f * pt::kPi * light.radius * light.radius;
    } else {
      light.shape = -1;            // the reference leaves the shape uninitialised and throws "No shape for light." at set-up
    }
    lights.push_back(light);
  };
  auto readProperties = [&]() {                                          // scene.cpp:93-101; parsed, used by nothing
    forEachBlockLine(file, line, [&](const char* l) { sscanf(l, " width %i", &width); sscanf(l, " height %i", &height); });
  };
  auto readMesh = [&]() {                                                // scene.cpp:103-122
    forEachBlockLine(file, line, [&](const char* l) {
      char word[kLineBytes];
      if (sscanf(l, " file %s", word) == 1) meshNames.push_back(word);
      if (sscanf(l, " material %s", word) == 1) {
        // meshNames[i] pairs with materials[i] by position: an unknown material name silently shifts every later mesh
        const auto found = materialByName.find(word);
        if (found == materialByName.end()) { printf("Could not find material %s\n", word); return; }
        materials.push_back(found->second);
        textures.push_back(textureByName[word]);
      }
    });
  };

  // ---- dispatch: one pass per top-level line, the four kinds tested in this order on whatever `line` holds by then ----
  while (fgets(line, kLineBytes, file)) {
    if (line[0] == '#') continue;
    char name[kLineBytes] = { 0 };
    if (sscanf(line, " material %s", name) == 1) readMaterial(name);
    if (strstr(line, "light")) readLight();
    if (strstr(line, "properties")) readProperties();
    if (strstr(line, "mesh")) readMesh();
  }
  fclose(file);
}
