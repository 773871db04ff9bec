// moptix_render -- headless replacement of the reference's Qt application (main.cpp:4-10 +
// MinimalOptiX ctor :9-33): picks a scene id, calls renderScene() and saves the canvas.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include <signal.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include "minimal_optix.h"

static void usage() {
  fprintf(stderr,
          "usage: moptix_render [--scene spheres|coffee|cornell_quads|random_spheres|random_spheres_256|dining_standin|million_standin|coffee_pot_standin|<file scene>]\n"
          "                     [--spp N] [--width W] [--height H] [--seed S] [--scenes DIR/] [--out PREFIX] [--outdir DIR]\n"
          "                     [--autosave] [--device D] [--random-seeds] [--strict-missing]\n"
          "       multi-GPU (one process per GPU, tile split + RCCL gather to rank 0, which writes the image):\n"
          "                     [--spawn N]  start N ranks of this program, rank r on device r, and wait for them\n"
          "                     [--spawn-same-device]  ... every rank on --device (a one-GPU box; needs a transport that accepts it,\n"
          "                                            MOPTIX_RCCL_LIB)   [--spawn-timeout S]  deadline of the ranks (default 600)\n"
          "                     [--rank R --ranks N --comm-file PATH]  one rank of a job started by something else\n");
}

int main(int argc, char** argv) {
  std::string scene = "spheres", prefix = "frame", scenes = "scenes/", outdir = ".";
  unsigned spp = 32, width = 1920, height = 1080, seed = 0;
  int device = 0; bool autosave = false, randomSeeds = false, strict = false;
  int rank = 0, ranks = 1, spawn = 0, spawnTimeout = 600; bool spawnSame = false; std::string commFile;
  for (int i = 1; i < argc; i++) {
    auto need = [&](const char* n) { if (i + 1 >= argc) { fprintf(stderr, "%s needs a value\n", n); exit(2); } return argv[++i]; };
    if (!strcmp(argv[i], "--scene")) scene = need("--scene");
    else if (!strcmp(argv[i], "--spp")) spp = (unsigned)atoi(need("--spp"));
    else if (!strcmp(argv[i], "--width")) width = (unsigned)atoi(need("--width"));
    else if (!strcmp(argv[i], "--height")) height = (unsigned)atoi(need("--height"));
    else if (!strcmp(argv[i], "--seed")) seed = (unsigned)strtoul(need("--seed"), nullptr, 0);
    else if (!strcmp(argv[i], "--scenes")) scenes = need("--scenes");
    else if (!strcmp(argv[i], "--out")) prefix = need("--out");
    else if (!strcmp(argv[i], "--outdir")) outdir = need("--outdir");
    else if (!strcmp(argv[i], "--device")) device = atoi(need("--device"));
    else if (!strcmp(argv[i], "--rank")) rank = atoi(need("--rank"));
    else if (!strcmp(argv[i], "--ranks")) ranks = atoi(need("--ranks"));
    else if (!strcmp(argv[i], "--comm-file")) commFile = need("--comm-file");
    else if (!strcmp(argv[i], "--spawn")) spawn = atoi(need("--spawn"));
    else if (!strcmp(argv[i], "--spawn-same-device")) spawnSame = true;
    else if (!strcmp(argv[i], "--spawn-timeout")) spawnTimeout = atoi(need("--spawn-timeout"));
    else if (!strcmp(argv[i], "--autosave")) autosave = true;
    else if (!strcmp(argv[i], "--random-seeds")) randomSeeds = true;
    else if (!strcmp(argv[i], "--strict-missing")) strict = true;
    else { usage(); return 2; }
  }
  if (spawn > 0) {
    // N ranks as child processes, forked before this process has touched the GPU (the parent never does); each child goes on
    // as rank r on device r -- no exec, a forked child that has not initialised HIP simply initialises it for itself
    const std::string idFile = outdir + "/.moptix_comm_" + std::to_string((long)getpid());
    remove(idFile.c_str());
    std::vector<pid_t> kids;
    bool child = false;
    for (int r = 0; r < spawn && !child; r++) {
      pid_t pid = fork();
      if (pid < 0) { perror("fork"); return 1; }
      if (pid == 0) { child = true; rank = r; ranks = spawn; if (!spawnSame) device = r; commFile = idFile; }
      else kids.push_back(pid);
    }
    if (!child) {
      // The ranks meet in collectives: one that dies (or never arrives) leaves its peers waiting for ever.  So the parent polls: the
      // first rank that fails, or the deadline, ends the others (SIGKILL), and the exit code says which it was (124 = deadline).
      int worst = 0; size_t live = kids.size();
      std::vector<bool> done(kids.size(), false);
      const time_t t0 = time(nullptr);
      while (live > 0) {
        for (size_t i = 0; i < kids.size(); i++) {
          if (done[i]) continue;
          int st = 0;
          if (waitpid(kids[i], &st, WNOHANG) == kids[i]) {
            done[i] = true; live--;
            const int rc = WIFEXITED(st) ? WEXITSTATUS(st) : 128;
            if (rc > worst) worst = rc;
          }
        }
        const bool late = time(nullptr) - t0 > spawnTimeout;
        if (live > 0 && (worst != 0 || late)) {
          fprintf(stderr, "moptix_render --spawn: %s; ending the other %zu rank(s)\n", late ? "deadline passed" : "a rank failed", live);
          for (size_t i = 0; i < kids.size(); i++) if (!done[i]) { kill(kids[i], SIGKILL); waitpid(kids[i], nullptr, 0); }
          if (late && worst == 0) worst = 124;
          break;
        }
        if (live > 0) usleep(20000);
      }
      remove(idFile.c_str());
      return worst;
    }
  }
  try {
    MinimalOptiX app(device);
    app.rank = rank; app.nRanks = ranks; app.commIdFile = commFile;
    app.fixedWidth = width; app.fixedHeight = height; app.nSuperSampling = spp;
    app.baseSeed = seed; app.reproducibleSeeds = !randomSeeds; app.skipMissingMeshes = !strict;
    app.baseSceneFolder = scenes; app.outputDir = outdir;
    app.setupContext();
    if (scene == "spheres") app.sceneId = MinimalOptiX::SCENE_SPHERES;
    else if (scene == "coffee") app.sceneId = MinimalOptiX::SCENE_COFFEE;
    else if (scene == "bedroom") app.sceneId = MinimalOptiX::SCENE_BEDROOM;
    else if (scene == "diningroom") app.sceneId = MinimalOptiX::SCENE_DININGROOM;
    else if (scene == "stormtrooper") app.sceneId = MinimalOptiX::SCENE_STORMTROOPER;
    else if (scene == "spaceship") app.sceneId = MinimalOptiX::SCENE_SPACESHIP;
    else if (scene == "cornell") app.sceneId = MinimalOptiX::SCENE_CORNELL;
    else if (scene == "hyperion") app.sceneId = MinimalOptiX::SCENE_HYPERION;
    else if (scene == "dragon") app.sceneId = MinimalOptiX::SCENE_DRAGON;
    else if (scene == "random_spheres_256") app.sceneId = MinimalOptiX::SCENE_SPHERES_VIDEO;
    else if (scene == "random_spheres") app.sceneId = MinimalOptiX::SCENE_RANDOM_SPHERES_500;
    else if (scene == "cornell_quads") app.sceneId = MinimalOptiX::SCENE_CORNELL_QUADS;
    else if (scene == "dining_standin") app.sceneId = MinimalOptiX::SCENE_DINING_STANDIN;
    else if (scene == "million_standin") app.sceneId = MinimalOptiX::SCENE_MILLION_STANDIN;
    else if (scene == "coffee_pot_standin") app.sceneId = MinimalOptiX::SCENE_COFFEE_POT_STANDIN;
    else { usage(); return 2; }
    app.renderScene(autosave, prefix);
    if (!autosave && rank == 0) app.saveCurrentFrame(false, prefix);
    fprintf(stderr, "render %.3f ms (device), BVH build %.3f ms, %u nodes, depth %u\n", app.lastRenderMs,
            app.lastAccel.buildMs, app.lastAccel.nNodes, app.lastAccel.treeDepth);
  } catch (const std::exception& e) {
    fprintf(stderr, "moptix_render: %s\n", e.what());
    return 1;
  }
  return 0;
}
