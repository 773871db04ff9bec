#include "minimal_optix.h"

#include <chrono>
#include <cstdio>
#include <cstring>
#include <limits>
#include <thread>
#include <random>
#include <stdexcept>

#include "../csrc/pt_rng.h"
#include "image_io.h"

using namespace moptix;

MinimalOptiX::MinimalOptiX(int device) {
  int rc = moptix_create(&context, device);
  if (rc != MOPTIX_OK) throw std::runtime_error(std::string("moptix_create: ") + moptix_last_error(nullptr));
  setupContext();
}

MinimalOptiX::~MinimalOptiX() { if (context) moptix_destroy(context); }

void MinimalOptiX::check(int rc, const char* what) {
  if (rc != MOPTIX_OK) throw std::runtime_error(std::string(what) + ": " + moptix_last_error(context));
}

// MinimalOptiX.cpp:130-152.  Ray types, entry points and the OptiX stack size have no
// equivalent (the megakernel is iterative); the context variables travel in moptix_params
// together with the camera and are pushed by setupScene().
void MinimalOptiX::setupContext() {
  canvas.assign((size_t)fixedWidth * fixedHeight * 3, 0);
}

// utils_host.cpp:118-122
int MinimalOptiX::randSeed() {
  if (reproducibleSeeds) return (int)pt::tea16(launchCounter++, baseSeed);
  static auto engine = std::minstd_rand(std::random_device{}());
  static auto randGen = std::uniform_real_distribution<float>(-1.f, 1.f);
  return int(randGen(engine) * (float)std::numeric_limits<int>::max());
}

// MinimalOptiX.cpp:154-357
void MinimalOptiX::setupScene() {
  aabb.invalidate();
  switch (sceneId) {
    case SCENE_SPHERES: buildSpheresScene(scene, fixedWidth, fixedHeight); break;
    case SCENE_COFFEE: setupScene("coffee"); return;
    case SCENE_BEDROOM: setupScene("bedroom"); return;
    case SCENE_DININGROOM: setupScene("diningroom"); return;
    case SCENE_STORMTROOPER: setupScene("stormtrooper"); return;
    case SCENE_SPACESHIP: setupScene("spaceship"); return;
    case SCENE_CORNELL: setupScene("cornell"); return;
    case SCENE_HYPERION: setupScene("hyperion"); return;
    case SCENE_DRAGON: setupScene("dragon"); return;
    case SCENE_SPHERES_VIDEO:                                                                          // :355
      buildRandomSpheresScene(scene, 256, fixedWidth, fixedHeight);
      videoParams.spheresParams = scene.spheres; videoParams.angle = 0.f;
      break;
    case SCENE_CORNELL_QUADS: buildCornellQuadsScene(scene, fixedWidth, fixedHeight); break;
    case SCENE_RANDOM_SPHERES_500: buildRandomSpheresScene(scene, 497, fixedWidth, fixedHeight); break;
    case SCENE_DINING_STANDIN: buildDiningStandInScene(scene, baseSceneFolder, 6, fixedWidth, fixedHeight); break;
    case SCENE_MILLION_STANDIN: buildProceduralMillionScene(scene, 1000000, fixedWidth, fixedHeight); break;
    case SCENE_COFFEE_POT_STANDIN: buildCoffeePotStandInScene(scene, baseSceneFolder, fixedWidth, fixedHeight); break;
  }
  scene.params.rayMaxDepth = rayMaxDepth; scene.params.rayMinIntensity = rayMinIntensity; scene.params.rayEpsilonT = rayEpsilonT;
  aabb = scene.aabb; nVertices = scene.nVertices; nFaces = scene.nFaces;
  check(upload(scene, context), "upload scene");
}

// MinimalOptiX.cpp:359-538 (+ the camera placement of :258-353)
void MinimalOptiX::setupScene(const char* sceneName) {
  buildFileScene(scene, baseSceneFolder, sceneName, fixedWidth, fixedHeight, skipMissingMeshes);
  scene.params.rayMaxDepth = rayMaxDepth; scene.params.rayMinIntensity = rayMinIntensity; scene.params.rayEpsilonT = rayEpsilonT;
  aabb = scene.aabb; nVertices = scene.nVertices; nFaces = scene.nFaces;
  if (verbose) for (const std::string& w : scene.warnings) fprintf(stderr, "[MinimalOptiX] %s\n", w.c_str());
  check(upload(scene, context), "upload scene");
}

// Communicator of this rank: rank 0 makes the id (ncclGetUniqueId behind moptix_comm_unique_id) and publishes it as a file
// (written under a temporary name, then renamed); the other ranks wait for the file.  moptix_comm_init is collective.
void MinimalOptiX::setupCommunicator() {
  if (commIdFile.empty()) throw std::runtime_error("commIdFile is not set (needed for nRanks > 1)");
  uint8_t id[MOPTIX_COMM_ID_BYTES];
  if (rank == 0) {
    check(moptix_comm_unique_id(id), "comm unique id");
    const std::string tmp = commIdFile + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    if (!f || fwrite(id, 1, sizeof(id), f) != sizeof(id)) { if (f) fclose(f); throw std::runtime_error("cannot write " + tmp); }
    fclose(f);
    if (rename(tmp.c_str(), commIdFile.c_str()) != 0) throw std::runtime_error("cannot publish " + commIdFile);
  } else {
    bool got = false;
    for (int tries = 0; tries < 1200 && !got; tries++) {         // two minutes
      FILE* f = fopen(commIdFile.c_str(), "rb");
      if (f) { got = fread(id, 1, sizeof(id), f) == sizeof(id); fclose(f); }
      if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
    if (!got) throw std::runtime_error("no communicator id at " + commIdFile);
  }
  check(moptix_comm_init(context, id, rank, nRanks), "comm init");
}

// MinimalOptiX.cpp:540-560
void MinimalOptiX::renderScene(bool autoSave, std::string fileNamePrefix) {
  if (nRanks < 1 || rank < 0 || rank >= nRanks) throw std::runtime_error("bad rank / nRanks");
  check(moptix_set_partition(context, rank, nRanks), "set partition");
  setupScene();
  check(moptix_validate(context), "validate");                 // :542
  const bool multi = nRanks > 1 || !commIdFile.empty();
  if (multi) { setupCommunicator(); autoSave = false; }         // progressive snapshots would need a gather each: one-GPU only
  moptix_get_accel_info(context, &lastAccel);
  if (canvas.size() != (size_t)fixedWidth * fixedHeight * 3) canvas.assign((size_t)fixedWidth * fixedHeight * 3, 0);
  moptix_kernel_time(context, nullptr, nullptr, 1);
  std::vector<int32_t> seeds(nSuperSampling);
  for (uint i = 0; i < nSuperSampling; ++i) seeds[i] = randSeed();   // :545 one seed per launch
  uint checkpoint = 1;
  uint done = 0;
  while (done < nSuperSampling) {
    // the reference launches once per sample (:546); the launches between two snapshots are
    // fused into one kernel -- per pixel the samples are still added in launch order.
    uint upto = nSuperSampling;
    if (autoSave) { while (checkpoint <= done) checkpoint *= 2; upto = std::min(checkpoint, nSuperSampling); }
    check(moptix_render(context, seeds.data() + done, (int32_t)(upto - done)), "render");
    done = upto;
    if (autoSave && done == checkpoint) {                       // :547-553
      updateContent((float)done, false);
      saveCurrentFrame(false, fileNamePrefix + "_" + std::to_string(done));
    }
  }
  if (multi) check(moptix_gather_tiles(context, 0), "gather tiles");      // the frame's one exchange: every rank's tiles to rank 0
  updateContent((float)nSuperSampling, true);                   // :555 (ranks > 0: clears their accuBuffer; their canvas is partial)
  if (autoSave) saveCurrentFrame(false, fileNamePrefix);        // :556-558
  uint64_t n = 0;
  moptix_kernel_time(context, &lastRenderMs, &n, 0);
  if (verbose) fprintf(stderr, "vertices: %zu faces: %zu\n", nVertices, nFaces);   // :559
}

// MinimalOptiX.cpp:43-66 (normalise, clamp, flip rows, optionally clear) -- done on the device
void MinimalOptiX::updateContent(float nAccumulation, bool clearBuffer) {
  check(moptix_resolve_rgb8(context, nAccumulation, clearBuffer ? 1 : 0, canvas.data()), "resolve");
}

// MinimalOptiX.cpp:68-84
void MinimalOptiX::saveCurrentFrame(bool popUpDialog, std::string fileNamePrefix) {
  (void)popUpDialog;
  std::string fileName = outputDir + "/" + (fileNamePrefix.empty() ? std::string("frame") : fileNamePrefix) + ".png";
  if (!writePNG(fileName, canvas.data(), fixedWidth, fixedHeight)) throw std::runtime_error("cannot write " + fileName);
  if (verbose) fprintf(stderr, "Image saved to %s\n", fileName.c_str());
}

// MinimalOptiX.cpp:587-592
void MinimalOptiX::animate(float time) { animateSpheres(videoParams, time); }

// MinimalOptiX.cpp:761-778: advance the physics, rewrite every sphere, new orbit camera, re-render
void MinimalOptiX::updateVideo() {
  animate(0.002f);
  check(moptix_update_spheres(context, 0, videoParams.spheresParams.data(), (int32_t)videoParams.spheresParams.size()), "update spheres");
  videoCamera(videoParams, (float)fixedWidth / (float)fixedHeight, scene.params.cam);
  check(moptix_set_params(context, &scene.params), "set params");
  std::vector<int32_t> seeds(nSuperSampling);
  for (uint i = 0; i < nSuperSampling; ++i) seeds[i] = randSeed();
  check(moptix_render(context, seeds.data(), (int32_t)nSuperSampling), "render");
  updateContent((float)nSuperSampling, true);
}

// MinimalOptiX.cpp:594-605 (generateVideo is dead code in the reference; frames are saved as PNG)
void MinimalOptiX::record(int frames, const char* filename, bool saveFrames) {
  (void)filename;
  for (int i = 0; i < frames; ++i) {
    updateVideo();
    if (saveFrames) saveCurrentFrame(false, "video" + std::to_string(i));
  }
}

// MinimalOptiX.cpp:112-117
void MinimalOptiX::videoDemo() {
  nSuperSampling = 128u;
  sceneId = SCENE_SPHERES_VIDEO;
  renderScene(false, "VIDEO");
  record(1000, "test.mp4", true);
}

// MinimalOptiX.cpp:86-110, restricted to the scenes whose assets exist
void MinimalOptiX::imageDemo() {
  nSuperSampling = 4096u;
  sceneId = SCENE_COFFEE;
  renderScene(true, "coffee");
}
