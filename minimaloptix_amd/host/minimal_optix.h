// minimal_optix.h -- headless mirror of the reference's `class MinimalOptiX`
// (MinimalOptiX.h:32-110): same member names, defaults and call order for the render
// path, with the Qt window and the OptiX context replaced by a canvas buffer and the
// C ABI of include/moptix.h.  The GUI, NVRTC and FFmpeg parts are out of scope.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/moptix.h"
#include "scene_desc.h"

typedef unsigned int uint;

class MinimalOptiX {
public:
  enum SceneId {                      // MinimalOptiX.h:36-47
    SCENE_SPHERES, SCENE_COFFEE, SCENE_BEDROOM, SCENE_DININGROOM, SCENE_STORMTROOPER,
    SCENE_SPACESHIP, SCENE_CORNELL, SCENE_HYPERION, SCENE_DRAGON, SCENE_SPHERES_VIDEO,
    // additions (BASELINE.json configs whose assets the reference does not ship)
    SCENE_CORNELL_QUADS, SCENE_RANDOM_SPHERES_500, SCENE_DINING_STANDIN, SCENE_MILLION_STANDIN,
    SCENE_COFFEE_POT_STANDIN
  };
  enum RayType { RAY_TYPE_RADIANCE, RAY_TYPE_SHADOW };

  explicit MinimalOptiX(int device = 0);      // ctor: setupContext() only; nothing is rendered
  ~MinimalOptiX();
  MinimalOptiX(const MinimalOptiX&) = delete;
  MinimalOptiX& operator=(const MinimalOptiX&) = delete;

  // utilities (MinimalOptiX.h:54-62)
  void setupContext();
  void setupScene();
  void setupScene(const char* sceneName);
  void renderScene(bool autoSave = false, std::string fileNamePrefix = "");
  void updateContent(float nAccumulation, bool clearBuffer);
  void saveCurrentFrame(bool popUpDialog, std::string fileNamePrefix = "");
  void imageDemo();
  void videoDemo();                                            // MinimalOptiX.cpp:112-117
  void record(int frames, const char* filename, bool saveFrames = false);   // :594-605 (frames are PNGs; no FFmpeg)
  void updateVideo();                                          // :761-778 one animated frame
  void animate(float time);                                    // :587-592
  moptix::VideoParams videoParams;

  // components
  std::vector<uint8_t> canvas;        // QImage::Format_RGB888, row 0 = top
  moptix_context context = nullptr;
  moptix::Aabb aabb;
  moptix::SceneDesc scene;            // what setupScene() built and uploaded
  std::string baseSceneFolder = "scenes/";

  // attributes (MinimalOptiX.h:81-89)
  SceneId sceneId = SCENE_SPHERES;
  uint fixedWidth = 1920u;
  uint fixedHeight = 1080u;
  uint nSuperSampling = 32u;
  uint rayMaxDepth = 256u;
  size_t nVertices = 0;
  size_t nFaces = 0;
  float rayMinIntensity = 0.001f;
  float rayEpsilonT = 0.001f;

  // seed schedule.  reproducible: launchSeed(i) = (int)tea<16>(i, baseSeed);
  // otherwise std::random_device as utils_host.cpp:118-122.
  bool reproducibleSeeds = true;
  uint baseSeed = 0u;
  bool skipMissingMeshes = true;      // coffee's Mesh010.obj is absent from the reference checkout
  std::string outputDir = ".";
  bool verbose = true;

  // multi-GPU (new): one process per GPU.  With nRanks > 1 renderScene() renders this rank's tiles of the frame
  // (moptix_set_partition), brings the context's RCCL communicator up (rank 0 writes the 128-byte id to commIdFile, the
  // others read it) and gathers the tiles into rank 0's accuBuffer (moptix_gather_tiles); only rank 0 resolves and saves.
  int rank = 0, nRanks = 1;
  std::string commIdFile;             // shared path for the communicator id; needed when nRanks > 1 (or to force a one-rank communicator)
  void setupCommunicator();

  // measurement of the last renderScene()
  double lastRenderMs = 0.0;          // device time of the render kernels
  moptix_accel_info lastAccel{};

  int randSeed();                     // utils_host.cpp:118-122 (next launch seed)

private:
  uint launchCounter = 0;
  void check(int rc, const char* what);
};
