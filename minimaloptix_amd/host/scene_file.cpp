#include "scene_file.h"

#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>

#include "../csrc/pt_math.h"

static const int kMaxLineLength = 2048;

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
inline pt::v3 v(const moptix_float3& f) { return pt::mk3(f.x, f.y, f.z); }
inline moptix_float3 f(const pt::v3& a) { return { a.x, a.y, a.z }; }
}

// Line-oriented, same matching rules and the same order of checks as scene.cpp:18-123:
// every sscanf pattern is tried on every line of a block; the block kinds are matched on
// whatever `line` holds after the previous block was consumed.
Scene::Scene(const char* fileName) {
  FILE* file = fopen(fileName, "r");
  if (!file) throw std::runtime_error(std::string("Couldn't open ") + fileName + " for reading.");

  std::map<std::string, moptix_disney_params> materialMap;
  std::map<std::string, std::string> textureMap;
  char line[kMaxLineLength];

  while (fgets(line, kMaxLineLength, file)) {
    if (line[0] == '#') continue;

    char name[kMaxLineLength] = { 0 };

    if (sscanf(line, " material %s", name) == 1) {                      // scene.cpp:28-57
      moptix_disney_params material;
      initDisneyParams(material);
      char texName[kMaxLineLength] = "";
      int brdf = material.brdfType;
      while (fgets(line, kMaxLineLength, file)) {
        if (strchr(line, '}')) break;
        sscanf(line, " name %s", name);
        sscanf(line, " color %f %f %f", &material.color.x, &material.color.y, &material.color.z);
        sscanf(line, " albedoTex %s", texName);
        sscanf(line, " emission %f %f %f", &material.emission.x, &material.emission.y, &material.emission.z);
        sscanf(line, " metallic %f", &material.metallic);
        sscanf(line, " subsurface %f", &material.subsurface);
        sscanf(line, " specular %f", &material.specular);
        sscanf(line, " specularTint %f", &material.specularTint);
        sscanf(line, " roughness %f", &material.roughness);
        sscanf(line, " anisotropic %f", &material.anisotropic);
        sscanf(line, " sheen %f", &material.sheen);
        sscanf(line, " sheenTint %f", &material.sheenTint);
        sscanf(line, " clearcoat %f", &material.clearcoat);
        sscanf(line, " clearcoatGloss %f", &material.clearcoatGloss);
        sscanf(line, " brdf %i", &brdf);
      }
      material.brdfType = brdf;
      material.albedoID = 0;
      materialMap[name] = material;
      textureMap[name] = texName;
    }

    if (strstr(line, "light")) {                                         // scene.cpp:59-91
      moptix_light_params light;
      memset(&light, 0, sizeof(light));
      moptix_float3 v1 = { 0, 0, 0 }, v2 = { 0, 0, 0 };
      char lightType[20] = "None";
      while (fgets(line, kMaxLineLength, file)) {
        if (strchr(line, '}')) break;
        sscanf(line, " position %f %f %f", &light.position.x, &light.position.y, &light.position.z);
        sscanf(line, " emission %f %f %f", &light.emission.x, &light.emission.y, &light.emission.z);
        sscanf(line, " normal %f %f %f", &light.normal.x, &light.normal.y, &light.normal.z);
        sscanf(line, " radius %f", &light.radius);
        sscanf(line, " v1 %f %f %f", &v1.x, &v1.y, &v1.z);
        sscanf(line, " v2 %f %f %f", &v2.x, &v2.y, &v2.z);
        sscanf(line, " type %19s", lightType);
      }
      if (strcmp(lightType, "Quad") == 0) {
        light.shape = MOPTIX_LIGHT_QUAD;
        const pt::v3 u = v(v1) - v(light.position), w = v(v2) - v(light.position);
        light.u = f(u); light.v = f(w);
        light.area = pt::length(pt::cross(u, w));
        light.normal = f(pt::normalize(pt::cross(u, w)));
      } else if (strcmp(lightType, "Sphere") == 0) {
        light.shape = MOPTIX_LIGHT_SPHERE;
        light.normal = f(pt::normalize(v(light.normal)));
        light.area = 4.0f * pt::kPi * light.radius * light.radius;
      } else {
        light.shape = -1;   // the reference leaves it uninitialised and later throws "No shape for light."
      }
      lights.push_back(light);
    }

    if (strstr(line, "properties")) {                                    // scene.cpp:93-101
      while (fgets(line, kMaxLineLength, file)) {
        if (strchr(line, '}')) break;
        sscanf(line, " width %i", &width);
        sscanf(line, " height %i", &height);
      }
    }

    if (strstr(line, "mesh")) {                                          // scene.cpp:103-122
      while (fgets(line, kMaxLineLength, file)) {
        if (strchr(line, '}')) break;
        char nm[kMaxLineLength];
        if (sscanf(line, " file %s", nm) == 1) meshNames.push_back(nm);
        if (sscanf(line, " material %s", nm) == 1) {
          if (materialMap.find(nm) != materialMap.end()) {
            materials.push_back(materialMap[nm]);
            textures.push_back(textureMap[nm]);
          } else {
            printf("Could not find material %s\n", nm);
          }
        }
      }
    }
  }
  fclose(file);
}
