// scene_file.h -- the `.scene` text format of the reference's file scenes
// (scene.h:18-27, scene.cpp:5-124; forked there from knightcrawler25/Optix-PathTracer).
#pragma once
#include <string>
#include <vector>
#include "../../include/moptix.h"

// utils_host.cpp:101-116
void initDisneyParams(moptix_disney_params& disneyParams);

class Scene {
public:
  // Throws std::runtime_error when the file cannot be opened (the reference prints a
  // message and then dereferences the NULL FILE*, scene.cpp:6-18).
  explicit Scene(const char* fileName);
  std::vector<std::string> meshNames;
  std::vector<moptix_disney_params> materials;   // materials[i] belongs to meshNames[i]
  std::vector<std::string> textures;             // albedoTex name per mesh ("" = none)
  std::vector<moptix_light_params> lights;
  int width = 0;     // parsed, never used (scene.cpp:98-99)
  int height = 0;
};
