// moptix_api.hip -- implementation of the C ABI of include/moptix.h on HIP (gfx950).
// Host-side staging of the scene, upload, LBVH build, launches, read-back, measurement.
// There is no CPU path: every entry point that computes needs a HIP device.
#include <hip/hip_runtime.h>
#include <chrono>
#include <thread>
// RCCL's types and prototypes: from its header where there is one, else the handful this file needs (the library is bound at
// run time by name, see load_rccl; a build box without the RCCL package -- or `make EXTRA=-DMOPTIX_NO_RCCL_HEADER` -- still
// builds the whole library, and moptix_comm_* answer MOPTIX_ERR_STATE where no librccl can be loaded).
#if !defined(MOPTIX_NO_RCCL_HEADER) && __has_include(<rccl/rccl.h>)
#include <rccl/rccl.h>
#else
extern "C" {
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5, ncclRemoteError = 6, ncclInProgress = 7 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclUint8 = 1, ncclInt32 = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5, ncclFloat16 = 6, ncclFloat32 = 7, ncclFloat = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclSum = 0, ncclProd = 1, ncclMax = 2, ncclMin = 3 } ncclRedOp_t;
ncclResult_t ncclGetUniqueId(ncclUniqueId* uniqueId);
ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId commId, int rank);
ncclResult_t ncclCommDestroy(ncclComm_t comm);
const char* ncclGetErrorString(ncclResult_t result);
ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, int root, ncclComm_t comm, hipStream_t stream);
ncclResult_t ncclGroupStart();
ncclResult_t ncclGroupEnd();
ncclResult_t ncclCommGetAsyncError(ncclComm_t comm, ncclResult_t* asyncError);
ncclResult_t ncclCommAbort(ncclComm_t comm);
}
#endif
#include <rocprim/rocprim.hpp>

#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/moptix.h"
#include "lbvh.h"
#include "megakernel.h"
#include "pt_upload.h"

using namespace pt;

namespace {

std::string g_lastError;   // errors without a context (create)

template <class T> struct DevBuf {
  T* p = nullptr; size_t n = 0;
  hipError_t ensure(size_t count) {
    if (count <= n && p) return hipSuccess;
    if (p) { (void)hipFree(p); p = nullptr; n = 0; }
    hipError_t e = hipMalloc((void**)&p, sizeof(T) * (count ? count : 1));
    if (e == hipSuccess) n = count ? count : 1;
    return e;
  }
  hipError_t upload(const std::vector<T>& v, hipStream_t s) {
    hipError_t e = ensure(v.size());
    if (e != hipSuccess || v.empty()) return e;
    return hipMemcpyAsync(p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice, s);
  }
  void release() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

}  // namespace

struct moptix_context_t {
  int device = 0;
  int numCUs = 256;
  hipStream_t stream = nullptr; bool ownStream = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr;
  std::string err;

  moptix_params params{}; bool haveParams = false;

  // host staging (copied from the caller, as OptiX copies on setUserData / map+memcpy)
  std::vector<DevMaterial> mats;
  std::vector<DevSphere> spheres; std::vector<int> sphereMat;
  std::vector<DevQuad> quads;
  std::vector<DevLight> lights;
  std::vector<float> facePos, faceNrm; std::vector<int> faceHasNrm, faceMat;
  std::vector<TriUV> faceUV; bool anyUV = false;
  struct HostTexture { int width, height; std::vector<float> rgba; };
  std::vector<HostTexture> textures;
  bool sceneDirty = true, accelBuilt = false;

  // device
  DevBuf<DevMaterial> dMats; DevBuf<DevSphere> dSpheres; DevBuf<int> dSphereMat; DevBuf<DevQuad> dQuads; DevBuf<DevLight> dLights;
  DevBuf<float> dFacePos, dFaceNrm; DevBuf<int> dFaceHasNrm, dFaceMat;
  DevBuf<TriUV> dFaceUV; DevBuf<float> dTexels; DevBuf<DevTexture> dTextures;
  LbvhResult bvh;
  DevBuf<float> dAccum; float* accumBound = nullptr; size_t accumPixels = 0;
  DevBuf<int> dSeeds; DevBuf<int> dWork; DevBuf<unsigned long long> dCounters; DevBuf<int> dOverflow; DevBuf<uint8_t> dRgb8;
  DevBuf<uint8_t> dPoolCold; DevBuf<float> dSampleBuf;

  int rank = 0, nRanks = 1;
  double glassFaceShare = 0.0;       // triangles whose material is glass (no next-event estimation at their hits), set by build_accel
  int optExitThreshold = 16, optLeafSize = 4, optBlocksPerCU = 3, optVariant = 3;
  int optTileMajor = 3;              // all samples of a pixel back to back, pixels with the deepest paths of earlier launches first
  long long tileHistoryTiles = -1;
  DevBuf<unsigned int> dTileCost, dTileCostSorted; DevBuf<int> dTileOrder, dTileIota; DevBuf<uint8_t> dSortTmp;
  int optStarveLanes = 16, optSampleBufMB = 16384, optLeafThreshold = 16, optSwapLanes = 32;
  unsigned long long lastExtra[5] = { 0, 0, 0, 0, 0 };

  std::vector<int> seedStaging;
  int optWatchdogMs = 600000;
  int optNodeFormat = 0;             // the node record the packet kernel fetches (pt_types.h): 64, 128, or 0 = whichever costs this scene less (choose_node_format)
  int nodeFormatUsed = 128;          // the verdict for this build (get_option "node_format_used")
  bool formatDecided = false;
  unsigned long long probeCounts[4] = { 0, 0, 0, 0 };     // node steps, triangle tests of the probe rays under Node128; the same under Node64
  int optFastShading = 0;
  int optBuilder = 1;
  int optAnalyticQueue = -1;          // -1 = by primitive count
  int optAutoPacket = 1;
  int lastVariant = -1;              // what the last render ran (get_option "kernel_variant_used")
  int countedSpanUs = -1, countedTailUs = -1;   // last counted launch: first wave in -> last wave out, and the part of it after the last work item was handed out
  int optCommBlocking = 0;           // 1 = plain ncclCommInitRank even where a non-blocking communicator is available
  bool commNonBlocking = false;      // the communicator was made with config.blocking = 0 (calls may return ncclInProgress: comm_settle)
  bool poisoned = false;             // a dead collective's kernels are still on the stream (comm_teardown): every call fails from here on
  int optCommTimeoutMs = 120000;     // deadline of a collective's completion (comm_wait); the first collective of a communicator also sets its links up
  int optShadowRule = 1;             // 1 = a shadow ray is decided by its nearest any-hit surface (default), 0 = SURVEY A2's order-independent rule
  bool variantExplicit = false;      // kernel_variant was set by the caller: no automatic choice
  int optSlotsInUse = -1;            // -1 = chosen per launch from its size
  int optDrainBelow = 64;            // a workgroup of the packet kernel with this many paths left hands them to the drain kernel (0 = off)
  int optAuxDepth = 16;              // variant 4: depth from which a path's shadow rays get slots of their own (0 = off)
  double kernelMs = 0.0, reduceMs = 0.0; uint64_t nLaunches = 0;
  bool asyncPending = false;
  // multi-GPU (one process per GPU): RCCL communicator of this rank + staging for the tile gather
  ncclComm_t comm = nullptr; int commRank = 0, commRanks = 1;
  DevBuf<float> dTileSend, dTileRecv;
};

namespace {

int fail(moptix_context c, int code, const std::string& msg) {
  if (c) c->err = msg; else g_lastError = msg;
  return code;
}
int hipFail(moptix_context c, hipError_t e, const char* what) {
  return fail(c, MOPTIX_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIPCHK(c, x, what) do { hipError_t e_ = (x); if (e_ != hipSuccess) return hipFail((c), e_, (what)); } while (0)

float* accum_ptr(moptix_context c) { return c->accumBound ? c->accumBound : c->dAccum.p; }

int ensure_accum(moptix_context c) {
  const size_t px = (size_t)c->params.width * c->params.height;
  if (c->accumBound) { c->accumPixels = px; return MOPTIX_OK; }
  if (c->dAccum.p && c->accumPixels == px) return MOPTIX_OK;
  // createBuffer(RT_BUFFER_INPUT_OUTPUT, FLOAT3, W, H) zeroed by the host (MinimalOptiX.cpp:144-147)
  HIPCHK(c, c->dAccum.ensure(3 * px), "alloc accuBuffer");
  HIPCHK(c, hipMemsetAsync(c->dAccum.p, 0, sizeof(float) * 3 * px, c->stream), "clear accuBuffer");
  c->accumPixels = px;
  return MOPTIX_OK;
}

void fill_view(moptix_context c, SceneView& v) {
  memset(&v, 0, sizeof(v));
  const moptix_params& p = c->params;
  v.width = (int)p.width; v.height = (int)p.height;
  v.maxDepth = (int)p.rayMaxDepth; v.minIntensity = p.rayMinIntensity; v.epsT = p.rayEpsilonT;
  v.bg = to_v3(p.bgColor); v.cam = make_cam(p.cam);
  v.nSpheres = (int)c->spheres.size(); v.spheres = c->dSpheres.p; v.sphereMat = c->dSphereMat.p;
  v.nQuads = (int)c->quads.size(); v.quads = c->dQuads.p;
  v.nLights = (int)c->lights.size(); v.lights = c->dLights.p;
  v.nMaterials = (int)c->mats.size(); v.mats = c->dMats.p;
  v.anyDisneyAnalytic = 0;
  for (size_t i = 0; i < c->spheres.size(); i++) if (c->mats[c->sphereMat[i]].kind == MAT_DISNEY) v.anyDisneyAnalytic = 1;
  for (size_t i = 0; i < c->quads.size(); i++) if (c->mats[c->quads[i].mat].kind == MAT_DISNEY) v.anyDisneyAnalytic = 1;
  v.shadowNearest = 0;
  if (c->optShadowRule != 0)          // option "shadow_rule": 0 keeps SURVEY A2's order-independent rule whatever the scene holds
    for (const DevMaterial& m : c->mats) if (m.kind == MAT_DISNEY && m.brdfType == BRDF_GLASS) v.shadowNearest = 1;
  v.nTris = c->bvh.nTris; v.rootRef = c->bvh.nTris > 0 ? c->bvh.rootRef : kEmptyRef;
  v.nodes = c->bvh.nodes; v.nodes64 = c->nodeFormatUsed == 64 ? c->bvh.nodes64 : nullptr; v.tris = c->bvh.tris; v.triShade = c->bvh.shade;
  v.triUV = (c->anyUV && c->bvh.nTris > 0) ? c->dFaceUV.p : nullptr;
  v.nTextures = (int)c->textures.size(); v.textures = c->dTextures.p;
}

// Which node record the packet kernel fetches for this scene.  The 64-byte form saves three of seven look-ups per node step but
// its boxes are a grid step larger.  Curved meshes hardly notice (coffee: +2 % node steps, +2 % triangle tests, frame -2.7 %);
// a ray that leaves a large axis-aligned face does -- the face's exact box is thinner than tmin and culls itself, its quantised
// box is a grid step of the PARENT thick and the ray starts inside it: the dining-room stand-in, whose walls are two triangles
// each, tests 17 % more triangles (it lost 6 % in round 3 and is level since round 4).  Static measures of the tree (surface-area inflation: 0.3 % for the dining
// room, 0.7 % for coffee) and synthetic rays miss this, so the scene is asked with its own paths: one sample per pixel of a
// 128-pixel-wide grid over the camera's view, cut at depth 6, walked under both forms; the counts are priced with the per-step
// costs fitted to coffee, the coffee pot, the glass knot and the dining room (a triangle test = 1.3 node steps of the
// 128-byte form; a 64-byte step = 0.93 of one since round 5, 0.78 in round 4).  Decided at the first render after a build or a change of frame size; a
// later change of camera keeps the verdict (moptix_set_params).
constexpr int kProbeWidth = 128;
int choose_node_format(moptix_context c) {
  c->formatDecided = true;
  c->nodeFormatUsed = 128;
  for (auto& v : c->probeCounts) v = 0;
  if (c->bvh.nNodes <= 0 || !c->bvh.nodes64) return MOPTIX_OK;      // no tree, or one without a 64-byte form (lbvh.h)
  if (c->optNodeFormat != 0) { c->nodeFormatUsed = c->optNodeFormat; return MOPTIX_OK; }
  SceneView v; fill_view(c, v);
  v.nodes64 = c->bvh.nodes64;
  const int w = std::min(kProbeWidth, v.width), h = std::max(1, (int)((long long)v.height * w / std::max(1, v.width)));
  v.width = w; v.height = h;
  const size_t threads = ((size_t)w * h + 255) / 256 * 256;
  unsigned long long* dOut = nullptr; int* dOvf = nullptr;
  hipError_t e = hipMalloc((void**)&dOut, 4 * sizeof(unsigned long long));
  if (e == hipSuccess) e = hipMemsetAsync(dOut, 0, 4 * sizeof(unsigned long long), c->stream);
  if (e == hipSuccess && c->bvh.stackBound > megakernel_lds_stack_entries())
    e = hipMalloc((void**)&dOvf, sizeof(int) * threads * (size_t)(c->bvh.stackBound - megakernel_lds_stack_entries() + 1));
  if (e == hipSuccess) e = launch_probe_paths(c->stream, v, 0, false, dOut, dOvf);
  if (e == hipSuccess) e = launch_probe_paths(c->stream, v, 0, true, dOut + 2, dOvf);
  if (e == hipSuccess) e = hipMemcpyAsync(c->probeCounts, dOut, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (dOut) (void)hipFree(dOut);
  if (dOvf) (void)hipFree(dOvf);
  if (e != hipSuccess) return hipFail(c, e, "node format probe");
  // Round 4 re-fit (profiles/r04_node_format.txt): the 64-byte step lost 45 instructions (sign-selected plane words) and the leaf pass
  // its dependent fetches, so a triangle test weighs 1.3 node steps instead of 3.5 and a 64-byte step 0.78 of a 128-byte one.
  const double cost128 = (double)c->probeCounts[0] + 1.3 * (double)c->probeCounts[1];
  // Round 5 re-fit (profiles/r05_node_format.txt): the 128-byte step is fetched by the ray's signs now (pt_path.h: no min / max per plane
  // pair), which makes it the cheaper step in instructions (100 against 130) and leaves the 64-byte one its four gathers against seven:
  // a 64-byte step = 0.93 of a 128-byte one.  Coffee, coffee + pot and the glass knot keep the 64-byte nodes (2-3 % faster), the dining
  // room -- whose quantised wall boxes cost it 17 % more triangle tests -- goes back to the 128-byte ones (38.7 against 40.6 ms).
  const double cost64 = 0.93 * (double)c->probeCounts[2] + 1.3 * (double)c->probeCounts[3];
  c->nodeFormatUsed = cost64 < cost128 ? 64 : 128;
  if (getenv("MOPTIX_DEBUG"))
    fprintf(stderr, "[moptix] node format probe (%dx%d paths): 128-byte nodes %llu steps %llu triangle tests, 64-byte %llu / %llu -> %d\n", w, h,
            c->probeCounts[0], c->probeCounts[1], c->probeCounts[2], c->probeCounts[3], c->nodeFormatUsed);
  return MOPTIX_OK;
}

int check_ready(moptix_context c) {
  if (!c) return fail(nullptr, MOPTIX_ERR_INVALID, "null context");
  if (c->poisoned) return fail(c, MOPTIX_ERR_COMM, "this context is unusable: kernels of an aborted collective never left its stream");
  if (!c->haveParams) return fail(c, MOPTIX_ERR_STATE, "moptix_set_params has not been called");
  if (!c->accelBuilt) return fail(c, MOPTIX_ERR_STATE, "moptix_build_accel has not been called since the scene changed");
  return MOPTIX_OK;
}

int read_stats(moptix_context c, moptix_stats* stats) {
  unsigned long long h[40 + 768 + 8 + 2 * kCensusRegions] = {};
  HIPCHK(c, hipMemcpy(h, c->dCounters.p, sizeof(unsigned long long) * (c->dCounters.n >= 816 + 2 * kCensusRegions ? 816 + 2 * kCensusRegions : (c->dCounters.n >= 816 ? 816 : 40)), hipMemcpyDeviceToHost), "read counters");
  stats->samples = h[0]; stats->primaryRays = h[1]; stats->bounceRays = h[2]; stats->shadowRays = h[3];
  stats->nodeFetches = h[4]; stats->triTests = h[5]; stats->closestHits = h[6]; stats->lightLoads = h[7];
  stats->analyticTests = h[8]; stats->traversalSteps = h[9]; stats->activeLaneSteps = h[10];
  stats->shadeBatches = h[11]; stats->shadeBatchLanes = h[12];
  // timeline of a counted launch (100 MHz s_memrealtime stamps of the queue kernels): options "counted_span_us" / "counted_tail_us"
  c->countedSpanUs = -1; c->countedTailUs = -1;
  if (h[38] && h[36] != ~0ull) { c->countedSpanUs = (int)((h[38] - h[36]) / 100); c->countedTailUs = (h[37] != ~0ull && h[38] > h[37]) ? (int)((h[38] - h[37]) / 100) : 0; }
  if (getenv("MOPTIX_DEBUG")) {
    fprintf(stderr, "[moptix] batches %llu lanes %llu full %llu allidle %llu waitingSum %llu\n", h[11], h[12], h[13], h[14], h[15]);
    const double tt = (double)h[21];
    fprintf(stderr, "[moptix] wave time: batch %.1f%% refill %.1f%% node %.1f%% leaf %.1f%% finish %.1f%% (steps %llu, cycles/wave %.3g)\n",
            100 * h[16] / tt, 100 * h[17] / tt, 100 * h[18] / tt, 100 * h[19] / tt, 100 * h[20] / tt, h[9], tt);
    fprintf(stderr, "[moptix] idle spins %llu\n", h[14]);
    // absolute pass clocks (s_memtime ticks summed over the waves) with the launch's span in the same ticks per wave: for a launch that
    // is one path walking (tools/gpu_lone_path.py) the passes are serial, so span - (batch + node + leaf) is what the scheduler costs
    if (h[38] && h[36] != ~0ull && h[38] > h[36])
      fprintf(stderr, "[moptix] pass ticks: batch %llu (load %llu run %llu store %llu) node %llu leaf %llu txn %llu lock %llu local %llu idle %llu | waves' time %llu over a span of %.1f us | "
                      "batches %llu node runs %llu leaf passes %llu iterations %llu transactions %llu\n", h[16], h[28], h[29], h[30], h[18], h[19], h[26], h[25], h[24], h[27], h[21],
              (double)(h[38] - h[36]) * 1e-2, h[11], h[39], h[22], h[32], h[31]);
    if (h[32]) fprintf(stderr, "[moptix] swap detail: local %.1f%% lock-wait %.1f%% txn %.1f%% idle %.1f%% | batch detail: load %.1f%% run %.1f%% store %.1f%% | "
                       "iterations %llu transactions %llu (cycles/iter %.0f)\n", 100 * h[24] / tt, 100 * h[25] / tt, 100 * h[26] / tt, 100 * h[27] / tt,
                       100 * h[28] / tt, 100 * h[29] / tt, 100 * h[30] / tt, h[32], h[31], tt / (double)h[32]);
    if (h[38]) fprintf(stderr, "[moptix] timeline (100 MHz clock): items ran out %.2f ms after the first wave started, last wave left %.2f ms after that\n",
                       (double)(h[37] - h[36]) * 1e-5, (double)(h[38] - h[37]) * 1e-5);
    if (h[38] && c->dCounters.n >= 808) {
      int lastB = 0; for (int b = 0; b < 256; b++) if (h[40 + b]) lastB = b;
      const int ranOut = (int)((double)(h[37] - h[36]) * 1e-5);
      fprintf(stderr, "[moptix] samples finishing per ms from the moment the items ran out (count, mean depth, max depth):");
      for (int b = ranOut > 2 ? ranOut - 2 : 0; b <= lastB; b++)
        fprintf(stderr, " [%d: %llu %.1f %llu]", b, h[40 + b], h[40 + b] ? (double)h[552 + b] / (double)h[40 + b] : 0.0, h[296 + b]);
      fprintf(stderr, "\n");
    }
    if (h[39]) fprintf(stderr, "[moptix] node runs %llu: node-ready slots waiting in the wave's ring %.1f, leaf ring %.1f (averages at the start of a run)\n",
                       h[39], (double)h[15] / (double)h[39], (double)h[13] / (double)h[39]);
    if (h[32]) fprintf(stderr, "[moptix] batch iterations executing on_result %llu, on_lights %llu, new item %llu (batches %llu)\n", h[33], h[34], h[35], h[11]);
    if (c->dCounters.n >= 816 && (h[808] | h[809])) {
      const double rays = (double)(h[1] + h[2] + h[3]);
      fprintf(stderr, "[moptix] slot-record rows (16 B each) per ray: shading visit loads %.2f stores %.2f | leaf pass loads %.2f stores %.2f | total %.1f B per ray in %.2f shading visits and %.2f leaf visits per ray\n",
              h[808] / rays, h[809] / rays, h[810] / rays, h[811] / rays, 16.0 * (double)(h[808] + h[809] + h[810] + h[811]) / rays, (double)h[12] / rays, (double)h[23] / rays);
    }
    if (c->dCounters.n >= 816 && (h[812] | h[813])) fprintf(stderr, "[moptix] hand-over: %llu paths handed over by the packet kernel, %llu taken by the drain kernel; samples finished: %llu by the packet kernel + %llu by the drain kernel (of %llu)\n", h[812], h[813], h[815], h[814], h[0]);
    if (h[22]) fprintf(stderr, "[moptix] node steps %llu (%.1f lanes avg), leaf passes %llu (%.1f lanes avg)\n", h[9] - h[22],
            (double)(h[10] - h[23]) / (double)(h[9] - h[22]), h[22], (double)h[23] / (double)h[22]);
    if (c->dCounters.n >= 816 + 2 * kCensusRegions && h[816 + kCensusRegions]) {      // lane census of the divergent regions (pt_path.h census<>)
      static const char* names[kCensusRegions] = { "result visit (on_result_packet)", "  miss", "  closest hit (hit_attributes + material)", "    light material",
        "    depth cap", "    lambertian", "    metal", "    glass", "    disney GLASS", "    disney (on_lights_packet)", "      light 0 faces: pdf + eval", "      light 1 faces: pdf + eval",
        "      light 2 faces: pdf + eval", "      bounce: pdf + eval", "new work item (begin_sample)", "leaf pass: triangle 0 tested", "leaf pass: triangle 1 tested",
        "leaf pass: triangle 2 tested", "leaf pass: triangle 3 tested", "  triangle hit accepted by tri_test", "  shadow result folded", "      light draw (per light)", "      bounce: disney_sample", "node step: branched tail (stack nearly full)" };
      fprintf(stderr, "[moptix] lane census: region | waves that entered | lanes that entered | lanes per wave (of 64)\n");
      for (int i = 0; i < kCensusRegions; i++)
        if (h[816 + kCensusRegions + i]) fprintf(stderr, "[moptix]   %-46s %12llu %14llu %6.1f\n", names[i], h[816 + kCensusRegions + i], h[816 + i], (double)h[816 + i] / (double)h[816 + kCensusRegions + i]);
    }
  }
  return MOPTIX_OK;
}

// One batch of launches = [trace kernel: every (pixel, sample) work item -> per-sample buffer]
// + [ordered reduction: accuBuffer[pixel] += samples in launch order].  Batches larger than the
// sample-buffer budget (or 2^31 work items) are cut into passes of whole launches.
int do_render(moptix_context c, const int32_t* seeds, int32_t nSeeds, bool counted, bool blocking, moptix_stats* stats) {
  int rc = check_ready(c);
  if (rc != MOPTIX_OK) return rc;
  if (nSeeds < 0 || (nSeeds > 0 && !seeds)) return fail(c, MOPTIX_ERR_INVALID, "bad seeds");
  HIPCHK(c, hipSetDevice(c->device), "hipSetDevice");
  if ((rc = moptix_sync(c)) != MOPTIX_OK) return rc;     // one batch in flight at a time
  if ((rc = ensure_accum(c)) != MOPTIX_OK) return rc;
  if (nSeeds == 0) return MOPTIX_OK;
  if (!c->formatDecided && (rc = choose_node_format(c)) != MOPTIX_OK) return rc;

  LaunchArgs a;
  memset(&a, 0, sizeof(a));
  fill_view(c, a.scene);
  a.accum = accum_ptr(c);
  const int tilesX = ((int)c->params.width + 7) / 8, tilesY = ((int)c->params.height + 7) / 8;
  const long long nTiles = (long long)tilesX * tilesY;
  const long long localTiles = (nTiles + c->nRanks - 1) / c->nRanks;     // one tile of every group of nRanks (megakernel.h item_to_pixel)
  if (localTiles * 64 > 0x7fffffffLL) return fail(c, MOPTIX_ERR_LIMIT, "frame too large");
  a.nItems = (int)(localTiles * 64); a.tilesX = tilesX; a.rank = c->rank; a.nRanks = c->nRanks;
  a.exitThreshold = c->optExitThreshold; a.leafThreshold = c->optLeafThreshold;
  if (a.nItems == 0) return MOPTIX_OK;
  const long long budget = (long long)c->optSampleBufMB << 20;
  long long perPass = budget / ((long long)a.nItems * 12);
  int nBlocks = c->numCUs * c->optBlocksPerCU;
  // the work counter is a 32-bit int that every path slot bumps once more after the items ran out
  const long long counterSlack = (long long)c->numCUs * std::max(4, c->optBlocksPerCU) * 1024 + 65536;      // the lean queue kernel runs four workgroups per CU
  if ((long long)a.nItems + counterSlack > 0x7fffffffLL) return fail(c, MOPTIX_ERR_LIMIT, "frame too large");
  perPass = std::min(perPass, (0x7fffffffLL - counterSlack) / a.nItems);
  perPass = std::max(1LL, std::min(perPass, (long long)nSeeds));

  const bool hasTris = a.scene.rootRef != kEmptyRef;
  // Scenes without triangles ("NoAccel"): the per-lane kernel, or ("analytic_queue" = 1) the queue kernel, where every ray
  // is finished by the brute-force lists at set-up, inside a full 64-lane batch, and the slots cycle through the batches.
  // With the lists read by scalar loads, four spheres per trip (pt_path.h trav_begin): random_spheres (497 + 33 primitives)
  // 60.6 ms per-lane, 47.8 ms queue (round 1: 92.3); cornell_quads (16 quads) 15.0 / 17.3 ms.  -1 = queue from 64 primitives on.
  const bool analyticQueue = c->optAnalyticQueue >= 0 ? c->optAnalyticQueue != 0 : (c->spheres.size() + c->quads.size() >= 64);
  // variant 4 (packetkernel.hip, one shading visit per bounce): triangle scenes with at most three lights and no Disney
  // material on an analytic primitive; anything else runs on variant 3.
  // "auto_packet" (default on): while "kernel_variant" has not been set, variant 4 is what such a scene runs on from 1e6
  // samples and 16 launches on.  Its paths have the shorter critical path (one rank's share of an 8-way split of the
  // benchmark frame: 66.7 against 78.4 ms) and, since the scene tables are read as constants (pt_types.h load_uniform),
  // its visits are the cheaper ones as well: whole frame 442 against 452 ms, dining room at 64 spp 162 against 193 ms.
  const bool packetOk = hasTris && a.scene.nLights <= 3 && !a.scene.anyDisneyAnalytic;
  const double nSamples = (double)a.nItems * (double)nSeeds;
  // (until round 4 mostly-glass scenes stayed on variant 3: no shadow rays to pack, and variant 4's wider records cost 10 % there; with
  // this round's leaf pass and routing the two are level on the glass knot -- 56.2 against 57.0 ms -- so the packet kernel serves both)
  const bool autoPacket = nSamples >= 1.0e6 && nSeeds >= 16;
  const bool usePacket = packetOk && (c->optVariant == 4 || (!c->variantExplicit && c->optAutoPacket != 0 && autoPacket));
  const bool useQueue = !usePacket && (c->optVariant >= 3) && (hasTris || analyticQueue);
  const bool leanQueue = useQueue && !hasTris;      // scenes without triangles: queuekernel_lean.hip
  c->lastVariant = usePacket ? 4 : useQueue ? 3 : 0;
  a.starveLanes = c->optStarveLanes; a.swapLanes = c->optSwapLanes;
  // Slots without a path are what deep paths borrow for their shadow rays (packetkernel.hip, "aux_depth"); once the work
  // items run out there are plenty, before that only the ones kept free here.  A launch under 1e8 samples (an 8-way share
  // of the benchmark frame) is short enough for its tail to matter more than the throughput of 64 more paths per pool:
  // 66.7 ms with 448 of 512 slots in use against 70.1 ms with all of them; a 4-way share: 130.3 against 125.0 ms.
  // A scene that is mostly glass has hardly any shadow rays to borrow slots for: all slots carry paths there (glass knot at 16 spp: 53.9 against 55.9 ms).
  a.slotsInUse = c->optSlotsInUse >= 0 ? c->optSlotsInUse : (usePacket && c->optAuxDepth > 0 && nSamples < 1.0e8 && c->glassFaceShare <= 0.5 ? packetkernel_slots() * 7 / 8 : 0);
  a.auxDepth = usePacket ? c->optAuxDepth : 0;
  a.watchdogTicks = (unsigned long long)c->optWatchdogMs * 100000ull;      // s_memrealtime counts at 100 MHz
  int drainBelow = 0;      // variant 4: the packet kernel's workgroups hand their last paths to the drain kernel (drainkernel.hip)
  if (usePacket) {
    a.ovfDepth = std::max(0, c->bvh.stackBound - packetkernel_lds_stack_entries() + 1);
    if (a.ovfDepth > 0) {
      HIPCHK(c, c->dOverflow.ensure(packetkernel_overflow_ints(nBlocks, a.ovfDepth)), "alloc stack overflow area");
      a.stackOverflow = c->dOverflow.p;
    }
    HIPCHK(c, c->dPoolCold.ensure(packetkernel_cold_bytes(nBlocks)), "alloc path pool");
    a.poolCold = c->dPoolCold.p;
    // the launch's last paths are finished by the drain kernel (drainkernel.hip; "drain_below" = 0 keeps them in the packet kernel)
    // (not under "shadow_rule" 0 in a scene with glass: there a shadow ray's attenuation is a PRODUCT over the glass surfaces it crosses, taken in
    // traversal order, and the drain kernel's order is not the packet kernel's)
    bool glassMaterial = false;
    for (const DevMaterial& m : c->mats) if (m.kind == MAT_DISNEY && m.brdfType == BRDF_GLASS) glassMaterial = true;
    drainBelow = (glassMaterial && !a.scene.shadowNearest) ? 0 : c->optDrainBelow;
  } else if (useQueue && leanQueue) {
    // no tree to walk (queuekernel_lean.hip): a fourth workgroup per CU instead of path slots and stack entries
    if (c->optBlocksPerCU == 3) nBlocks = c->numCUs * 4;
    a.ovfDepth = 0;
    HIPCHK(c, c->dPoolCold.ensure(queuekernel_cold_bytes_lean(nBlocks)), "alloc path pool");
    a.poolCold = c->dPoolCold.p;
  } else if (useQueue) {
    a.ovfDepth = std::max(0, c->bvh.stackBound - queuekernel_lds_stack_entries() + 1);
    if (a.ovfDepth > 0) {
      HIPCHK(c, c->dOverflow.ensure(queuekernel_overflow_ints(nBlocks, a.ovfDepth)), "alloc stack overflow area");
      a.stackOverflow = c->dOverflow.p;
    }
    HIPCHK(c, c->dPoolCold.ensure(queuekernel_cold_bytes(nBlocks)), "alloc path pool");
    a.poolCold = c->dPoolCold.p;
  } else {
    const int ldsStack = megakernel_lds_stack_entries();
    if (c->bvh.stackBound > ldsStack) {
      const size_t need = (size_t)(c->bvh.stackBound - ldsStack + 1) * nBlocks * 256;
      HIPCHK(c, c->dOverflow.ensure(need), "alloc stack overflow area");
      a.stackOverflow = c->dOverflow.p;
    }
  }
  // the per-sample buffer is the one large allocation (up to "sample_buffer_mb", 16 GB by default; two pipelined contexts
  // hold one each): when the device cannot give it, run more and smaller passes instead of failing the render
  for (;;) {
    const hipError_t e = c->dSampleBuf.ensure((size_t)perPass * a.nItems * 3);
    if (e == hipSuccess) break;
    if (e != hipErrorOutOfMemory || perPass <= 1) return hipFail(c, e, "alloc per-sample buffer");
    (void)hipGetLastError();
    perPass = (perPass + 1) / 2;
  }
  a.sampleBuf = c->dSampleBuf.p;
  // [0] work counter, [1] watchdog flag, then (variant 4) the drain list (megakernel.h kDrain*): counters, capacity, threshold, entries
  HIPCHK(c, c->dWork.ensure(2 + drain_list_ints(nBlocks, drainBelow)), "alloc work counter");
  {
    const int hdr[2 + kDrainEntries] = { 0, 0, 0, 0, 0, 0, nBlocks * drainBelow, drainBelow };
    HIPCHK(c, hipMemcpyAsync(c->dWork.p, hdr, sizeof(hdr), hipMemcpyHostToDevice, c->stream), "init work counter");
    HIPCHK(c, hipStreamSynchronize(c->stream), "sync");      // hdr lives on this stack frame
  }
  a.tileMajor = (useQueue || usePacket) ? c->optTileMajor : 0;
  a.tileOrder = nullptr; a.tileCost = nullptr;
  a.unitShift = a.tileMajor == 3 ? 0 : 6;
  const long long historyUnits = (localTiles * 64) >> a.unitShift;
  if (a.tileMajor) {
    if (c->tileHistoryTiles != historyUnits) {           // new frame size / partition / granularity: forget the history
      HIPCHK(c, c->dTileCost.ensure((size_t)historyUnits), "alloc tile cost");
      HIPCHK(c, c->dTileCostSorted.ensure((size_t)historyUnits), "alloc tile cost");
      HIPCHK(c, c->dTileOrder.ensure((size_t)historyUnits), "alloc tile order");
      std::vector<int> iota((size_t)historyUnits);
      for (size_t i = 0; i < iota.size(); i++) iota[i] = (int)i;
      HIPCHK(c, c->dTileIota.upload(iota, c->stream), "upload tile ids");
      HIPCHK(c, hipMemsetAsync(c->dTileCost.p, 0, sizeof(unsigned int) * (size_t)historyUnits, c->stream), "zero tile cost");
      size_t tmpBytes = 0;
      HIPCHK(c, rocprim::radix_sort_pairs_desc(nullptr, tmpBytes, c->dTileCost.p, c->dTileCostSorted.p, c->dTileIota.p, c->dTileOrder.p,
                                               (size_t)historyUnits, 0, 32, c->stream), "size tile sort");
      HIPCHK(c, c->dSortTmp.ensure(tmpBytes), "alloc sort scratch");
      HIPCHK(c, hipStreamSynchronize(c->stream), "sync tile history");    // iota staging dies here
      c->tileHistoryTiles = historyUnits;
    }
    a.tileOrder = c->dTileOrder.p; a.tileCost = c->dTileCost.p;
  }
  a.workCounter = c->dWork.p;
  if (counted) {
    HIPCHK(c, c->dCounters.ensure(40 + 768 + 8 + 2 * kCensusRegions), "alloc counters");
    HIPCHK(c, hipMemsetAsync(c->dCounters.p, 0, sizeof(unsigned long long) * (40 + 768 + 8 + 2 * kCensusRegions), c->stream), "zero counters");
    HIPCHK(c, hipMemsetAsync(c->dCounters.p + 36, 0xff, sizeof(unsigned long long) * 2, c->stream), "init min counters");
    a.counters = c->dCounters.p;
  }
#ifdef PT_EVLOG      // experiment build only (packetkernel.hip PT_EV): the event log stands in for the counters of an UNCOUNTED launch
  DevBuf<unsigned long long> evLog;
  HIPCHK(c, evLog.ensure(65536 + 8), "alloc event log");      // always there: the kernel logs whenever it meets a path deeper than 100 bounces
  HIPCHK(c, hipMemsetAsync(evLog.p, 0, sizeof(unsigned long long) * (65536 + 8), c->stream), "zero event log");
  a.evLog = evLog.p;
#endif
  c->seedStaging.assign(seeds, seeds + nSeeds);   // lives in the context: the copy below may still be in flight when an async render returns
  HIPCHK(c, c->dSeeds.upload(c->seedStaging, c->stream), "upload seeds");

  for (long long first = 0; first < nSeeds; first += perPass) {
    const int n = (int)std::min(perPass, (long long)nSeeds - first);
    a.seeds = c->dSeeds.p + first; a.nSeeds = n; a.nWork = n * a.nItems;
    HIPCHK(c, hipMemsetAsync(c->dWork.p, 0, (2 + kDrainCap) * sizeof(int), c->stream), "zero work counter");      // counters only: capacity and threshold stay
    if (a.tileMajor && a.tileCost) {
      // tiles in descending order of the deepest path seen so far (stable: ties stay in raster order)
      size_t tmpBytes = c->dSortTmp.n;
      HIPCHK(c, rocprim::radix_sort_pairs_desc(c->dSortTmp.p, tmpBytes, c->dTileCost.p, c->dTileCostSorted.p, c->dTileIota.p, c->dTileOrder.p,
                                               (size_t)historyUnits, 0, 32, c->stream), "sort tiles");
    }
    HIPCHK(c, hipEventRecord(c->ev0, c->stream), "event");
    if (usePacket) HIPCHK(c, launch_packetkernel(c->stream, a, nBlocks, counted, c->optFastShading != 0), "launch packet megakernel");
    if (usePacket && drainBelow > 0) HIPCHK(c, launch_drainkernel(c->stream, a, c->numCUs, counted, c->optFastShading != 0), "launch drain kernel");
    else if (useQueue && leanQueue) HIPCHK(c, launch_queuekernel_lean(c->stream, a, nBlocks, counted, c->optFastShading != 0), "launch queue megakernel (lean)");
    else if (useQueue) HIPCHK(c, launch_queuekernel(c->stream, a, nBlocks, counted, c->optFastShading != 0), "launch queue megakernel");
    else HIPCHK(c, launch_megakernel(c->stream, a, nBlocks, counted), "launch megakernel");
    HIPCHK(c, hipEventRecord(c->ev1, c->stream), "event");
    HIPCHK(c, launch_reduce_samples(c->stream, a), "launch sample reduction");
    HIPCHK(c, hipEventRecord(c->ev2, c->stream), "event");
    c->asyncPending = true;
    const bool last = first + perPass >= nSeeds;
    if (!last || blocking) { if ((rc = moptix_sync(c)) != MOPTIX_OK) return rc; }
  }
#ifdef PT_EVLOG
  if (blocking && !counted && getenv("MOPTIX_EVLOG")) {
    std::vector<unsigned long long> h(65536);
    HIPCHK(c, hipMemcpy(h.data(), evLog.p, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost), "read event log");
    FILE* f = fopen(getenv("MOPTIX_EVLOG"), "wb");
    if (f) { fwrite(h.data(), sizeof(unsigned long long), (size_t)std::min<unsigned long long>(h[0], 65000ull) + 1, f); fclose(f); }
  }
  evLog.release();
#endif
  if (blocking && a.tileCost && getenv("MOPTIX_DEBUG")) {       // how the deepest-path history is distributed over the tiles
    std::vector<unsigned int> cost((size_t)historyUnits);
    HIPCHK(c, hipMemcpy(cost.data(), c->dTileCost.p, sizeof(unsigned int) * cost.size(), hipMemcpyDeviceToHost), "read tile cost");
    size_t hist[6] = { 0, 0, 0, 0, 0, 0 };                        // 0, 8..15, 16..63, 64..255, 256+, first half of the tiles holding 256+
    for (size_t i = 0; i < cost.size(); i++) {
      const unsigned int v = cost[i];
      hist[v == 0 ? 0 : v < 16 ? 1 : v < 64 ? 2 : v < 256 ? 3 : 4]++;
      if (v >= 256 && i < cost.size() / 2) hist[5]++;
    }
    fprintf(stderr, "[moptix] depth history (%zu units): none %zu, depth 8-15 %zu, 16-63 %zu, 64-255 %zu, capped %zu (of which %zu in the first half)\n",
            cost.size(), hist[0], hist[1], hist[2], hist[3], hist[4], hist[5]);
  }
  if (blocking && counted && stats) return read_stats(c, stats);
  return MOPTIX_OK;
}

// ---- tile split: a rank's tiles <-> a dense buffer (work-item order of megakernel.h item_to_pixel) ----
struct TileDeal { int nItems, tilesX, rank, nRanks, width, height; };
__device__ __forceinline__ bool deal_pixel(const TileDeal& d, int i, int& pixel) {
  const int lt = i >> 6, in = i & 63;
  const int gt = lt * d.nRanks + (d.rank + lt) % d.nRanks;
  const int tx = gt % d.tilesX, ty = gt / d.tilesX;
  const int x = tx * 8 + (in & 7), y = ty * 8 + (in >> 3);
  pixel = y * d.width + x;
  return (x < d.width) & (y < d.height);
}
__global__ void __launch_bounds__(256) k_pack_tiles(const float* __restrict__ accum, float* __restrict__ packed, TileDeal d) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= d.nItems) return;
  int px; float r = 0.f, g = 0.f, b = 0.f;
  if (deal_pixel(d, i, px)) { r = accum[3 * (size_t)px]; g = accum[3 * (size_t)px + 1]; b = accum[3 * (size_t)px + 2]; }
  packed[3 * (size_t)i] = r; packed[3 * (size_t)i + 1] = g; packed[3 * (size_t)i + 2] = b;
}
__global__ void __launch_bounds__(256) k_unpack_tiles(const float* __restrict__ packed, float* __restrict__ accum, TileDeal d) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= d.nItems) return;
  int px;
  if (deal_pixel(d, i, px)) { accum[3 * (size_t)px] = packed[3 * (size_t)i]; accum[3 * (size_t)px + 1] = packed[3 * (size_t)i + 1]; accum[3 * (size_t)px + 2] = packed[3 * (size_t)i + 2]; }
}
int tile_deal(moptix_context c, int rank, int nRanks, TileDeal& d) {
  if (!c->haveParams) return fail(c, MOPTIX_ERR_STATE, "no params");
  if (nRanks < 1 || rank < 0 || rank >= nRanks) return fail(c, MOPTIX_ERR_INVALID, "bad partition");
  const int tilesX = ((int)c->params.width + 7) / 8, tilesY = ((int)c->params.height + 7) / 8;
  const long long localTiles = ((long long)tilesX * tilesY + nRanks - 1) / nRanks;
  if (localTiles * 64 > 0x7fffffffLL) return fail(c, MOPTIX_ERR_LIMIT, "frame too large");
  d.nItems = (int)(localTiles * 64); d.tilesX = tilesX; d.rank = rank; d.nRanks = nRanks;
  d.width = (int)c->params.width; d.height = (int)c->params.height;
  return MOPTIX_OK;
}
// RCCL is bound at the first moptix_comm_* call, not at load time: a host process that already carries an RCCL (PyTorch
// ships its own librccl.so.1) must keep exactly one copy, and a process that never goes multi-GPU needs none.  dlopen by
// soname returns the copy that is already loaded, else the one on this library's run path (/opt/rocm/lib).
struct RcclApi {
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr; decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr; decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclSend) Send = nullptr; decltype(&ncclRecv) Recv = nullptr; decltype(&ncclReduce) Reduce = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr; decltype(&ncclGroupEnd) GroupEnd = nullptr;
  // optional (used by comm_wait when the library has them): error state of a communicator without blocking, and tearing one down
  // while its kernels are still on the stream
  decltype(&ncclCommGetAsyncError) CommGetAsyncError = nullptr; decltype(&ncclCommAbort) CommAbort = nullptr;
  // optional: a NON-BLOCKING communicator (config.blocking = 0).  With a blocking one ncclSend / ncclGroupEnd / ncclReduce may sit inside
  // the library while the links to a peer are set up -- a peer that is alive but never calls blocks the host there, where no deadline of
  // ours can reach.  A non-blocking communicator returns ncclInProgress instead and the state is polled (comm_settle)
  decltype(&ncclCommInitRankConfig) CommInitRankConfig = nullptr;
  bool ok = false; std::string error;
};
// MOPTIX_RCCL_LIB names another library with the same nine entry points (a transport plug point; tests/rccl_loopback is a
// loop-back transport that lets the N > 1 branches below run as N processes on a ONE-GPU box, where RCCL itself refuses a
// communicator whose ranks share a device).  Loaded once; the C++11 static makes the first call thread-safe.
RcclApi load_rccl() {
  RcclApi api;
  const char* override_ = getenv("MOPTIX_RCCL_LIB");
  void* h = nullptr;
  if (override_ && *override_) {
    h = dlopen(override_, RTLD_NOW | RTLD_LOCAL);
    if (!h) { api.error = std::string("cannot load MOPTIX_RCCL_LIB: ") + dlerror(); return api; }
  } else {
    h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) { api.error = std::string("cannot load librccl: ") + dlerror(); return api; }
  }
  bool all = true;
  auto sym = [&](const char* n) { void* p = dlsym(h, n); if (!p) { all = false; api.error = std::string("librccl lacks ") + n; } return p; };
  api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId"); api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
  api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy"); api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
  api.Send = (decltype(api.Send))sym("ncclSend"); api.Recv = (decltype(api.Recv))sym("ncclRecv"); api.Reduce = (decltype(api.Reduce))sym("ncclReduce");
  api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart"); api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
  api.ok = all;
  api.CommGetAsyncError = (decltype(api.CommGetAsyncError))dlsym(h, "ncclCommGetAsyncError");
  api.CommAbort = (decltype(api.CommAbort))dlsym(h, "ncclCommAbort");
  api.CommInitRankConfig = (decltype(api.CommInitRankConfig))dlsym(h, "ncclCommInitRankConfig");
  return api;
}
RcclApi& rccl() {
  static RcclApi api = load_rccl();
  return api;
}
int ncclFail(moptix_context c, ncclResult_t r, const char* what) {
  return fail(c, MOPTIX_ERR_HIP, std::string(what) + ": " + rccl().GetErrorString(r));
}
#define RCCL_READY(c) do { if (!rccl().ok) return fail((c), MOPTIX_ERR_STATE, rccl().error); } while (0)
#define NCCLCHK(c, x, what) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) return ncclFail((c), r_, (what)); } while (0)

// The end of a collective: wait for the context's stream WITH A DEADLINE.  A collective's kernels spin on the device until every
// peer has joined; a peer that died or never calls leaves them spinning, and a bare hipStreamSynchronize would then block this
// rank for good (only the CLI's --spawn parent has its own deadline).  So the stream is polled; each poll asks the communicator
// for an asynchronous error (a peer's process gone, a link down); on an error or after "comm_timeout_ms" the communicator is
// ABORTED (ncclCommAbort makes its kernels leave), the stream is given a bounded time to drain, and the call returns
// MOPTIX_ERR_COMM: the host is expected to exit (bench.py, class MinimalOptiX and dist.py raise).  The context keeps working
// as a one-rank context; a new communicator needs moptix_comm_init again.
// Tears a communicator down whose kernels or host-side operations cannot complete.  With ncclCommAbort its kernels leave and its
// resources go; WITHOUT it the communicator is leaked -- ncclCommDestroy waits for outstanding work, i.e. for the very thing that
// does not come.  The stream gets a bounded time to drain; if it has not by then the context is marked unusable (every later call
// returns MOPTIX_ERR_COMM): kernels of a dead collective still sit on its stream.
int comm_teardown(moptix_context c, const char* what, const std::string& why) {
  using clock = std::chrono::steady_clock;
  bool leaked = false;
  if (c->comm) {
    if (rccl().CommAbort) (void)rccl().CommAbort(c->comm); else leaked = true;
    c->comm = nullptr; c->commRank = 0; c->commRanks = 1; c->commNonBlocking = false;
  }
  const auto t1 = clock::now();                     // the aborted kernels leave; never wait for them without a bound either
  while (hipStreamQuery(c->stream) == hipErrorNotReady && clock::now() - t1 < std::chrono::seconds(10)) std::this_thread::sleep_for(std::chrono::milliseconds(1));
  const bool busy = hipStreamQuery(c->stream) == hipErrorNotReady;
  if (busy) c->poisoned = true;
  return fail(c, MOPTIX_ERR_COMM, std::string(what) + ": " + why + (leaked ? "; the communicator was abandoned (this library has no ncclCommAbort)" : "; the communicator was aborted") +
                                  (busy ? "; its kernels are still on the stream: this context is unusable from here on" : ""));
}
// After a call on a NON-BLOCKING communicator: ncclInProgress means the library is still working on it in the background (setting links
// up, waiting for the peer's side of a connection); nothing else may be issued on the communicator until that has settled.  Polled
// against the same deadline as the device side ("comm_timeout_ms"); a hard error or the deadline tears the communicator down.
int comm_settle(moptix_context c, ncclResult_t r, const char* what) {
  using clock = std::chrono::steady_clock;
  if (r == ncclSuccess) return MOPTIX_OK;
  if (r != ncclInProgress || !c->commNonBlocking || !c->comm) return ncclFail(c, r, what);
  const auto t0 = clock::now();
  const auto deadline = std::chrono::milliseconds(c->optCommTimeoutMs);
  for (unsigned spin = 0;; spin++) {
    ncclResult_t st = ncclSuccess;
    const ncclResult_t q = rccl().CommGetAsyncError(c->comm, &st);
    if (q != ncclSuccess) return comm_teardown(c, what, std::string("ncclCommGetAsyncError: ") + rccl().GetErrorString(q));
    if (st == ncclSuccess) return MOPTIX_OK;
    if (st != ncclInProgress) return comm_teardown(c, what, std::string("communicator reports ") + rccl().GetErrorString(st));
    if (clock::now() - t0 > deadline)
      return comm_teardown(c, what, "still in progress on the host after comm_timeout_ms = " + std::to_string(c->optCommTimeoutMs) + " (a peer is alive but has not made its call)");
    if (spin < 4096) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(100));
  }
}
int comm_wait(moptix_context c, const char* what) {
  using clock = std::chrono::steady_clock;
  const auto t0 = clock::now();
  const auto deadline = std::chrono::milliseconds(c->optCommTimeoutMs);
  std::string why;
  for (unsigned spin = 0;; spin++) {
    const hipError_t q = hipStreamQuery(c->stream);
    if (q == hipSuccess) return MOPTIX_OK;
    if (q != hipErrorNotReady) return hipFail(c, q, what);
    ncclResult_t aerr = ncclSuccess;
    if (c->comm && rccl().CommGetAsyncError && rccl().CommGetAsyncError(c->comm, &aerr) == ncclSuccess && aerr != ncclSuccess && aerr != ncclInProgress) {
      why = std::string("communicator reports ") + rccl().GetErrorString(aerr); break;
    }
    if (clock::now() - t0 > deadline) { why = "no completion within comm_timeout_ms = " + std::to_string(c->optCommTimeoutMs) + " (a peer is missing or late)"; break; }
    if (spin < 4096) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(100));
  }
  return comm_teardown(c, what, why);
}

}  // namespace

extern "C" {

const char* moptix_version(void) { return "minimaloptix_amd 0.1 (gfx950)"; }

const char* moptix_last_error(moptix_context ctx) { return ctx ? ctx->err.c_str() : g_lastError.c_str(); }

int moptix_create(moptix_context* out, int device) {
  if (!out) return fail(nullptr, MOPTIX_ERR_INVALID, "null out pointer");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return fail(nullptr, MOPTIX_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
  if (device < 0 || device >= n) return fail(nullptr, MOPTIX_ERR_INVALID, "device index out of range");
  e = hipSetDevice(device);
  if (e != hipSuccess) return hipFail(nullptr, e, "hipSetDevice");
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, device);
  if (e != hipSuccess) return hipFail(nullptr, e, "hipGetDeviceProperties");
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, MOPTIX_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", kernels are built for gfx950 only");
  moptix_context c = new moptix_context_t();
  c->device = device; c->numCUs = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) { delete c; return hipFail(nullptr, e, "hipStreamCreate"); }
  c->ownStream = true;
  if ((e = hipEventCreate(&c->ev0)) != hipSuccess || (e = hipEventCreate(&c->ev1)) != hipSuccess || (e = hipEventCreate(&c->ev2)) != hipSuccess) {
    (void)moptix_destroy(c);                       // releases the stream and whichever events exist
    return hipFail(nullptr, e, "hipEventCreate");
  }
  *out = c;
  return MOPTIX_OK;
}

int moptix_destroy(moptix_context c) {
  if (!c) return MOPTIX_ERR_INVALID;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  c->dMats.release(); c->dSpheres.release(); c->dSphereMat.release(); c->dQuads.release(); c->dLights.release();
  c->dFacePos.release(); c->dFaceNrm.release(); c->dFaceHasNrm.release(); c->dFaceMat.release();
  c->dFaceUV.release(); c->dTexels.release(); c->dTextures.release();
  lbvh_free(&c->bvh);
  c->dPoolCold.release(); c->dSampleBuf.release();
  c->dTileCost.release(); c->dTileCostSorted.release(); c->dTileOrder.release(); c->dTileIota.release(); c->dSortTmp.release();
  c->dAccum.release(); c->dSeeds.release(); c->dWork.release(); c->dCounters.release(); c->dOverflow.release(); c->dRgb8.release();
  c->dTileSend.release(); c->dTileRecv.release();
  if (c->comm) { (void)rccl().CommDestroy(c->comm); c->comm = nullptr; }
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->ev2) (void)hipEventDestroy(c->ev2);
  if (c->ownStream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return MOPTIX_OK;
}

int moptix_set_stream(moptix_context c, void* hipStream) {
  if (!c) return MOPTIX_ERR_INVALID;
  (void)hipStreamSynchronize(c->stream);
  if (c->ownStream && c->stream) (void)hipStreamDestroy(c->stream);
  if (hipStream) { c->stream = (hipStream_t)hipStream; c->ownStream = false; }
  else { HIPCHK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking), "hipStreamCreate"); c->ownStream = true; }
  return MOPTIX_OK;
}

int moptix_set_params(moptix_context c, const moptix_params* p) {
  if (!c || !p) return fail(c, MOPTIX_ERR_INVALID, "null argument");
  if (p->width == 0 || p->height == 0) return fail(c, MOPTIX_ERR_INVALID, "zero-sized launch");
  // the node step sorts entry distances by their bit patterns (pt_path.h ChildKey): distances are clamped to tmin, which must not be negative
  if (!(p->rayEpsilonT >= 0.0f)) return fail(c, MOPTIX_ERR_INVALID, "rayEpsilonT must be >= 0");
  const bool resized = !c->haveParams || p->width != c->params.width || p->height != c->params.height;
  // The node format is chosen by walking the scene's own paths from the camera of the first render after a build
  // (choose_node_format: a device allocation, two probe launches and a stream synchronisation).  A new frame size asks again; a
  // new CAMERA alone does not: a moving camera (updateVideo, MinimalOptiX.cpp:761-778) would otherwise pay that blocking probe in
  // every frame, inside moptix_render_async as well.  set_option("node_format", 0) asks again explicitly.
  if (!c->haveParams || resized) c->formatDecided = false;
  c->params = *p; c->haveParams = true;
  if (resized && !c->accumBound) { c->accumPixels = 0; }
  return MOPTIX_OK;
}

int moptix_clear_scene(moptix_context c) {
  if (!c) return MOPTIX_ERR_INVALID;
  c->mats.clear(); c->spheres.clear(); c->sphereMat.clear(); c->quads.clear(); c->lights.clear();
  c->facePos.clear(); c->faceNrm.clear(); c->faceHasNrm.clear(); c->faceMat.clear();
  c->faceUV.clear(); c->anyUV = false; c->textures.clear();
  c->sceneDirty = true; c->accelBuilt = false;
  return MOPTIX_OK;
}

int moptix_add_texture(moptix_context c, const float* rgba, int32_t width, int32_t height, int32_t* outTexId) {
  if (!c || !rgba || width <= 0 || height <= 0) return fail(c, MOPTIX_ERR_INVALID, "bad texture");
  moptix_context_t::HostTexture t;
  t.width = width; t.height = height;
  t.rgba.assign(rgba, rgba + 4 * (size_t)width * (size_t)height);
  c->textures.push_back(std::move(t));
  if (outTexId) *outTexId = (int32_t)c->textures.size();     // ids start at 1; 0 is RT_TEXTURE_ID_NULL
  c->sceneDirty = true; c->accelBuilt = false;
  return MOPTIX_OK;
}

int moptix_add_material(moptix_context c, const moptix_material* m, int32_t* outMatId) {
  if (!c || !m) return fail(c, MOPTIX_ERR_INVALID, "null argument");
  if (m->kind < MOPTIX_MAT_LAMBERTIAN || m->kind > MOPTIX_MAT_LIGHT) return fail(c, MOPTIX_ERR_INVALID, "unknown material kind");
  if (m->kind == MOPTIX_MAT_DISNEY && (m->disney.albedoID < 0 || (size_t)m->disney.albedoID > c->textures.size()))
    return fail(c, MOPTIX_ERR_INVALID, "albedoID names no texture (add textures before the materials that use them)");
  c->mats.push_back(make_dev_material(*m));
  if (outMatId) *outMatId = (int32_t)c->mats.size() - 1;
  c->sceneDirty = true; c->accelBuilt = false;
  return MOPTIX_OK;
}

static int check_mat(moptix_context c, int32_t id) {
  if (id < 0 || id >= (int32_t)c->mats.size()) return fail(c, MOPTIX_ERR_INVALID, "material id out of range");
  return MOPTIX_OK;
}

int moptix_add_spheres(moptix_context c, const moptix_sphere_params* s, const int32_t* matIds, int32_t n) {
  if (!c || n < 0 || (n > 0 && (!s || !matIds))) return fail(c, MOPTIX_ERR_INVALID, "bad argument");
  for (int32_t i = 0; i < n; i++) {
    if (check_mat(c, matIds[i]) != MOPTIX_OK) return MOPTIX_ERR_INVALID;
    c->spheres.push_back(make_dev_sphere(s[i])); c->sphereMat.push_back(matIds[i]);
  }
  c->sceneDirty = true; c->accelBuilt = false;
  return MOPTIX_OK;
}

int moptix_add_quads(moptix_context c, const moptix_quad_params* q, const int32_t* matIds, int32_t n) {
  if (!c || n < 0 || (n > 0 && (!q || !matIds))) return fail(c, MOPTIX_ERR_INVALID, "bad argument");
  for (int32_t i = 0; i < n; i++) {
    if (check_mat(c, matIds[i]) != MOPTIX_OK) return MOPTIX_ERR_INVALID;
    c->quads.push_back(make_dev_quad(q[i], matIds[i]));
  }
  c->sceneDirty = true; c->accelBuilt = false;
  return MOPTIX_OK;
}

int moptix_add_mesh(moptix_context c, const float* positions, int32_t nVerts, const float* normals, int32_t nNormals,
                    const float* texcoords, int32_t nTexcoords, const int32_t* vIdx, const int32_t* nIdx, const int32_t* tIdx,
                    int32_t nFaces, int32_t matId) {
  if (!c || nFaces < 0 || nVerts < 0 || (nFaces > 0 && (!positions || !vIdx))) return fail(c, MOPTIX_ERR_INVALID, "bad argument");
  if (check_mat(c, matId) != MOPTIX_OK) return MOPTIX_ERR_INVALID;
  for (int32_t f = 0; f < 3 * nFaces; f++)
    if (vIdx[f] < 0 || vIdx[f] >= nVerts) return fail(c, MOPTIX_ERR_INVALID, "vertex index out of range");
  const bool meshHasNormals = normals && nNormals > 0 && nIdx;      // normalBuffer.size() != 0, Geometry.cu:136
  const bool meshHasUVs = texcoords && nTexcoords > 0 && tIdx;      // texcoordBuffer.size() != 0, Geometry.cu:141
  for (int32_t f = 0; f < nFaces; f++) {
    bool hasN = meshHasNormals;
    for (int k = 0; k < 3; k++) {
      const float* p = positions + 3 * (size_t)vIdx[3 * f + k];
      c->facePos.push_back(p[0]); c->facePos.push_back(p[1]); c->facePos.push_back(p[2]);
      if (hasN && (nIdx[3 * f + k] < 0 || nIdx[3 * f + k] >= nNormals)) hasN = false;
    }
    for (int k = 0; k < 3; k++) {
      if (hasN) { const float* q = normals + 3 * (size_t)nIdx[3 * f + k]; c->faceNrm.push_back(q[0]); c->faceNrm.push_back(q[1]); c->faceNrm.push_back(q[2]); }
      else { c->faceNrm.push_back(0.f); c->faceNrm.push_back(0.f); c->faceNrm.push_back(0.f); }
    }
    c->faceHasNrm.push_back(hasN ? 1 : 0);
    c->faceMat.push_back(matId);
    TriUV uv; memset(&uv, 0, sizeof(uv));
    if (meshHasUVs) {
      const int32_t* ti = tIdx + 3 * (size_t)f;
      if (ti[0] >= 0 && ti[0] < nTexcoords && ti[1] >= 0 && ti[1] < nTexcoords && ti[2] >= 0 && ti[2] < nTexcoords) {
        uv.u0 = texcoords[2 * (size_t)ti[0]]; uv.v0 = texcoords[2 * (size_t)ti[0] + 1];
        uv.u1 = texcoords[2 * (size_t)ti[1]]; uv.v1 = texcoords[2 * (size_t)ti[1] + 1];
        uv.u2 = texcoords[2 * (size_t)ti[2]]; uv.v2 = texcoords[2 * (size_t)ti[2] + 1];
        uv.hasUV = 1; c->anyUV = true;
      }
    }
    c->faceUV.push_back(uv);
  }
  c->sceneDirty = true; c->accelBuilt = false;
  return MOPTIX_OK;
}

int moptix_set_lights(moptix_context c, const moptix_light_params* lights, int32_t n) {
  if (!c || n < 0 || (n > 0 && !lights)) return fail(c, MOPTIX_ERR_INVALID, "bad argument");
  c->lights.clear();
  for (int32_t i = 0; i < n; i++) {
    if (lights[i].shape != MOPTIX_LIGHT_SPHERE && lights[i].shape != MOPTIX_LIGHT_QUAD) return fail(c, MOPTIX_ERR_INVALID, "No shape for light.");
    c->lights.push_back(make_dev_light(lights[i]));
  }
  c->sceneDirty = true; c->accelBuilt = false;
  return MOPTIX_OK;
}

int moptix_update_spheres(moptix_context c, int32_t first, const moptix_sphere_params* s, int32_t n) {
  if (!c || !s || first < 0 || n < 0 || (size_t)first + (size_t)n > c->spheres.size()) return fail(c, MOPTIX_ERR_INVALID, "bad sphere range");
  for (int32_t i = 0; i < n; i++) c->spheres[first + i] = make_dev_sphere(s[i]);
  if (c->accelBuilt && n > 0) {
    HIPCHK(c, hipSetDevice(c->device), "hipSetDevice");
    HIPCHK(c, hipMemcpyAsync(c->dSpheres.p + first, c->spheres.data() + first, sizeof(DevSphere) * n, hipMemcpyHostToDevice, c->stream), "update spheres");
    HIPCHK(c, hipStreamSynchronize(c->stream), "sync");
  }
  return MOPTIX_OK;
}

int moptix_build_accel(moptix_context c, const char* kind) {
  if (!c || !kind) return fail(c, MOPTIX_ERR_INVALID, "null argument");
  const bool trbvh = !strcmp(kind, "Trbvh") || !strcmp(kind, "Lbvh");
  if (!trbvh && strcmp(kind, "NoAccel")) return fail(c, MOPTIX_ERR_INVALID, std::string("unknown acceleration: ") + kind);
  const int nFaces = (int)c->faceMat.size();
  if (!trbvh && nFaces > 0) return fail(c, MOPTIX_ERR_INVALID, "NoAccel with triangle meshes is not supported; use Trbvh");
  // Whatever happens below, the tree the context holds is not valid for the scene any more: a failed build must not leave accelBuilt set
  // with a freed or stale tree behind it.
  c->accelBuilt = false;
  // The kernels address every scene table with a 32-bit byte offset from its base (pt_types.h at32) and a leaf reference holds its first
  // record in 28 bits; the builder indexes triangles with 32-bit ints.  Checked on the INPUT, before anything is uploaded or built: the node
  // table has at most one four-wide node per two triangles (pt_lbvh.h), so the triangle count bounds every table.
  if ((unsigned long long)nFaces >= (1ull << 28) || (unsigned long long)nFaces * sizeof(Node128) / 2 >= (1ull << 32))
    return fail(c, MOPTIX_ERR_LIMIT, "acceleration structure too large: triangle / node tables must stay below 4 GB (2^28 triangles at most)");
  if ((unsigned long long)c->mats.size() * sizeof(DevMaterial) >= (1ull << 32)) return fail(c, MOPTIX_ERR_LIMIT, "material table must stay below 4 GB");
  HIPCHK(c, hipSetDevice(c->device), "hipSetDevice");
  HIPCHK(c, c->dMats.upload(c->mats, c->stream), "upload materials");
  HIPCHK(c, c->dSpheres.upload(c->spheres, c->stream), "upload spheres");
  HIPCHK(c, c->dSphereMat.upload(c->sphereMat, c->stream), "upload sphere materials");
  HIPCHK(c, c->dQuads.upload(c->quads, c->stream), "upload quads");
  HIPCHK(c, c->dLights.upload(c->lights, c->stream), "upload lights");
  {  // texture buffers: all texels in one allocation, one descriptor per sampler
    std::vector<float> texels; std::vector<size_t> offs;
    for (const auto& t : c->textures) { offs.push_back(texels.size()); texels.insert(texels.end(), t.rgba.begin(), t.rgba.end()); }
    HIPCHK(c, c->dTexels.upload(texels, c->stream), "upload texels");
    std::vector<DevTexture> desc;
    for (size_t i = 0; i < c->textures.size(); i++)
      desc.push_back(DevTexture{ reinterpret_cast<const v4*>(c->dTexels.p + offs[i]), c->textures[i].width, c->textures[i].height });
    HIPCHK(c, c->dTextures.upload(desc, c->stream), "upload texture descriptors");
    HIPCHK(c, hipStreamSynchronize(c->stream), "sync texture upload");   // staging vectors die here
  }
  lbvh_free(&c->bvh);
  if (nFaces > 0) {
    if (c->anyUV) HIPCHK(c, c->dFaceUV.upload(c->faceUV, c->stream), "upload face texcoords");
    HIPCHK(c, c->dFacePos.upload(c->facePos, c->stream), "upload face positions");
    HIPCHK(c, c->dFaceNrm.upload(c->faceNrm, c->stream), "upload face normals");
    HIPCHK(c, c->dFaceHasNrm.upload(c->faceHasNrm, c->stream), "upload face flags");
    {
      std::vector<int> words(c->faceMat.size());      // material id + what the face is to a shadow ray (pt_types.h SHADOW_*)
      for (size_t f = 0; f < words.size(); f++) { const DevMaterial& m = c->mats[c->faceMat[f]]; words[f] = face_mat_word(c->faceMat[f], shadow_class(m.kind, m.brdfType)); }
      HIPCHK(c, c->dFaceMat.upload(words, c->stream), "upload face materials");
      HIPCHK(c, hipStreamSynchronize(c->stream), "sync face material upload");      // the staging vector dies here
    }
    size_t glassFaces = 0;
    for (int m : c->faceMat) glassFaces += (c->mats[m].kind == MAT_GLASS || (c->mats[m].kind == MAT_DISNEY && c->mats[m].brdfType == BRDF_GLASS)) ? 1 : 0;
    c->glassFaceShare = (double)glassFaces / (double)nFaces;
    HIPCHK(c, lbvh_build(c->stream, c->dFacePos.p, c->dFaceNrm.p, c->dFaceHasNrm.p, c->dFaceMat.p, nFaces, c->optLeafSize, c->optBuilder, &c->bvh), "LBVH build");
    // (the same limits again on what was built: belt and braces)
    const unsigned long long lim = 1ull << 32;
    if ((unsigned long long)c->bvh.nTris * sizeof(Tri48) >= lim || (unsigned long long)c->bvh.nTris * sizeof(TriShade) >= lim ||
        (unsigned long long)c->bvh.nNodes * sizeof(Node128) >= lim || (unsigned long long)c->bvh.nTris >= (1ull << 28)) {
      lbvh_free(&c->bvh);
      return fail(c, MOPTIX_ERR_LIMIT, "acceleration structure too large: triangle / node tables must stay below 4 GB (2^28 triangles at most)");
    }
  }
  HIPCHK(c, hipStreamSynchronize(c->stream), "sync after upload");
  c->formatDecided = false;            // choose_node_format at the next render: it needs the camera
  c->sceneDirty = false; c->accelBuilt = true;
  c->tileHistoryTiles = -1;            // new scene: forget which tiles had deep paths
  return MOPTIX_OK;
}

int moptix_get_accel_info(moptix_context c, moptix_accel_info* out) {
  if (!c || !out) return MOPTIX_ERR_INVALID;
  memset(out, 0, sizeof(*out));
  out->nTriangles = (uint32_t)c->bvh.nTris; out->nNodes = (uint32_t)c->bvh.nNodes; out->maxLeafSize = (uint32_t)c->bvh.leafSize;
  out->treeDepth = (uint32_t)c->bvh.depth; out->buildMs = c->bvh.buildMs;
  out->nodeBytes = (uint64_t)c->bvh.nNodes * (sizeof(Node128) + (c->bvh.nodes64 ? sizeof(Node64) : 0)); out->triBytes = (uint64_t)c->bvh.nTris * sizeof(Tri48);
  return MOPTIX_OK;
}

int moptix_validate(moptix_context c) {
  int rc = check_ready(c);
  if (rc != MOPTIX_OK) return rc;
  if (c->mats.empty()) return fail(c, MOPTIX_ERR_STATE, "scene has no materials");
  if (c->spheres.empty() && c->quads.empty() && c->faceMat.empty()) return fail(c, MOPTIX_ERR_STATE, "scene has no geometry");
  for (const DevMaterial& m : c->mats)
    if (m.kind == MAT_DISNEY && m.brdfType != BRDF_NORMAL && m.brdfType != BRDF_GLASS) return fail(c, MOPTIX_ERR_INVALID, "bad brdfType");
  return MOPTIX_OK;
}

int moptix_launch(moptix_context c, int32_t randSeed) { return do_render(c, &randSeed, 1, false, true, nullptr); }
int moptix_render(moptix_context c, const int32_t* seeds, int32_t nSeeds) { return do_render(c, seeds, nSeeds, false, true, nullptr); }
int moptix_render_async(moptix_context c, const int32_t* seeds, int32_t nSeeds) { return do_render(c, seeds, nSeeds, false, false, nullptr); }
int moptix_render_counted(moptix_context c, const int32_t* seeds, int32_t nSeeds, moptix_stats* out) {
  if (out) memset(out, 0, sizeof(*out));
  return do_render(c, seeds, nSeeds, true, true, out);
}

int moptix_sync(moptix_context c) {
  if (!c) return MOPTIX_ERR_INVALID;
  HIPCHK(c, hipStreamSynchronize(c->stream), "stream synchronize");
  if (c->asyncPending) {
    float ms = 0.f, ms2 = 0.f;
    if (hipEventElapsedTime(&ms, c->ev0, c->ev1) == hipSuccess) { c->kernelMs += ms; c->nLaunches++; }
    if (hipEventElapsedTime(&ms2, c->ev1, c->ev2) == hipSuccess) c->reduceMs += ms2;
    c->asyncPending = false;
    int flags[2] = { 0, 0 };
    if (c->dWork.p) HIPCHK(c, hipMemcpy(flags, c->dWork.p, sizeof(flags), hipMemcpyDeviceToHost), "read watchdog flag");
    if (flags[1] != 0) return fail(c, MOPTIX_ERR_HIP, "render kernel hit its watchdog (option watchdog_ms); this pass was not added to accuBuffer");
  }
  return MOPTIX_OK;
}

int moptix_set_partition(moptix_context c, int32_t rank, int32_t nRanks) {
  if (!c || nRanks < 1 || rank < 0 || rank >= nRanks) return fail(c, MOPTIX_ERR_INVALID, "bad partition");
  if (rank != c->rank || nRanks != c->nRanks) c->tileHistoryTiles = -1;   // other tiles: the deep-path history does not apply
  c->rank = rank; c->nRanks = nRanks;
  return MOPTIX_OK;
}

int moptix_set_option(moptix_context c, const char* name, int32_t value) {
  if (!c || !name) return MOPTIX_ERR_INVALID;
  if (!strcmp(name, "exit_threshold")) { if (value < 0 || value > 64) return fail(c, MOPTIX_ERR_INVALID, "exit_threshold in [0,64]"); c->optExitThreshold = value; }
  else if (!strcmp(name, "leaf_size")) { if (value < 1 || value > kMaxLeaf) return fail(c, MOPTIX_ERR_INVALID, "leaf_size in [1,8]"); if (value != c->optLeafSize) c->accelBuilt = false; c->optLeafSize = value; }
  else if (!strcmp(name, "blocks_per_cu")) { if (value < 1 || value > 8) return fail(c, MOPTIX_ERR_INVALID, "blocks_per_cu in [1,8]"); c->optBlocksPerCU = value; }
  else if (!strcmp(name, "kernel_variant")) {
    if (value != -1 && value != 0 && value != 3 && value != 4) return fail(c, MOPTIX_ERR_INVALID, "kernel_variant in {-1,0,3,4} (1 and 2 were removed in round 3)");
    // -1: back to the library's own choice per launch (do_render: the packet kernel for long launches of eligible scenes)
    c->optVariant = value < 0 ? 3 : value; c->variantExplicit = value >= 0;
  }
  else if (!strcmp(name, "sample_buffer_mb")) { if (value < 1) return fail(c, MOPTIX_ERR_INVALID, "sample_buffer_mb >= 1"); c->optSampleBufMB = value; }
  else if (!strcmp(name, "leaf_threshold")) { if (value < 1 || value > 64) return fail(c, MOPTIX_ERR_INVALID, "leaf_threshold in [1,64]"); c->optLeafThreshold = value; }
  else if (!strcmp(name, "swap_lanes")) { if (value < 1 || value > 64) return fail(c, MOPTIX_ERR_INVALID, "swap_lanes in [1,64]"); c->optSwapLanes = value; }
  else if (!strcmp(name, "starve_lanes")) { if (value < 1 || value > 64) return fail(c, MOPTIX_ERR_INVALID, "starve_lanes in [1,64]"); c->optStarveLanes = value; }
  else if (!strcmp(name, "tile_major")) { if (value < 0 || value > 3) return fail(c, MOPTIX_ERR_INVALID, "tile_major in {0,1,2,3}"); c->optTileMajor = value; }
  else if (!strcmp(name, "auto_packet")) { if (value < 0 || value > 1) return fail(c, MOPTIX_ERR_INVALID, "auto_packet in {0,1}"); c->optAutoPacket = value; }
  else if (!strcmp(name, "analytic_queue")) { if (value < -1 || value > 1) return fail(c, MOPTIX_ERR_INVALID, "analytic_queue in {-1,0,1}"); c->optAnalyticQueue = value; }
  else if (!strcmp(name, "aux_depth")) { if (value < 0 || value > 100000) return fail(c, MOPTIX_ERR_INVALID, "aux_depth in [0,100000]"); c->optAuxDepth = value; }
  else if (!strcmp(name, "drain_below")) { if (value < 0 || value > 64) return fail(c, MOPTIX_ERR_INVALID, "drain_below in [0,64]"); c->optDrainBelow = value; }
  else if (!strcmp(name, "slots_in_use")) { if (value < -1 || value > 1024) return fail(c, MOPTIX_ERR_INVALID, "slots_in_use in [-1,1024]"); c->optSlotsInUse = value; }
  else if (!strcmp(name, "builder")) { if (value < 0 || value > 1) return fail(c, MOPTIX_ERR_INVALID, "builder in {0,1}"); if (value != c->optBuilder) c->accelBuilt = false; c->optBuilder = value; }
  else if (!strcmp(name, "fast_shading")) { if (value < 0 || value > 1) return fail(c, MOPTIX_ERR_INVALID, "fast_shading in {0,1}"); c->optFastShading = value; }
  else if (!strcmp(name, "node_format")) { if (value != 0 && value != 64 && value != 128) return fail(c, MOPTIX_ERR_INVALID, "node_format in {0,64,128}"); c->formatDecided = false; c->optNodeFormat = value; }
  else if (!strcmp(name, "shadow_rule")) { if (value < 0 || value > 1) return fail(c, MOPTIX_ERR_INVALID, "shadow_rule in {0,1}"); c->optShadowRule = value; }
  else if (!strcmp(name, "watchdog_ms")) { if (value < 1) return fail(c, MOPTIX_ERR_INVALID, "watchdog_ms >= 1"); c->optWatchdogMs = value; }
  else if (!strcmp(name, "comm_timeout_ms")) { if (value < 1) return fail(c, MOPTIX_ERR_INVALID, "comm_timeout_ms >= 1"); c->optCommTimeoutMs = value; }
  else if (!strcmp(name, "forget_history")) { c->tileHistoryTiles = -1; }      // the next launch orders its work like a context's first (measurement of a cold frame)
  else if (!strcmp(name, "comm_blocking")) { if (value < 0 || value > 1) return fail(c, MOPTIX_ERR_INVALID, "comm_blocking in {0,1}"); c->optCommBlocking = value; }
  else return fail(c, MOPTIX_ERR_INVALID, std::string("unknown option: ") + name);
  return MOPTIX_OK;
}

int moptix_get_option(moptix_context c, const char* name, int32_t* value) {
  if (!c || !name || !value) return MOPTIX_ERR_INVALID;
  if (!strcmp(name, "exit_threshold")) *value = c->optExitThreshold;
  else if (!strcmp(name, "leaf_size")) *value = c->optLeafSize;
  else if (!strcmp(name, "blocks_per_cu")) *value = c->optBlocksPerCU;
  else if (!strcmp(name, "kernel_variant")) *value = c->variantExplicit ? c->optVariant : -1;
  else if (!strcmp(name, "sample_buffer_mb")) *value = c->optSampleBufMB;
  else if (!strcmp(name, "leaf_threshold")) *value = c->optLeafThreshold;
  else if (!strcmp(name, "swap_lanes")) *value = c->optSwapLanes;
  else if (!strcmp(name, "starve_lanes")) *value = c->optStarveLanes;
  else if (!strcmp(name, "tile_major")) *value = c->optTileMajor;
  else if (!strcmp(name, "watchdog_ms")) *value = c->optWatchdogMs;
  else if (!strcmp(name, "comm_timeout_ms")) *value = c->optCommTimeoutMs;
  else if (!strcmp(name, "comm_blocking")) *value = c->optCommBlocking;
  else if (!strcmp(name, "comm_nonblocking_used")) *value = c->commNonBlocking ? 1 : 0;
  else if (!strcmp(name, "node_format")) *value = c->optNodeFormat;
  else if (!strcmp(name, "node_format_used")) *value = c->nodeFormatUsed;
  else if (!strcmp(name, "fast_shading")) *value = c->optFastShading;
  else if (!strcmp(name, "builder")) *value = c->optBuilder;
  else if (!strcmp(name, "aux_depth")) *value = c->optAuxDepth;
  else if (!strcmp(name, "slots_in_use")) *value = c->optSlotsInUse;
  else if (!strcmp(name, "drain_below")) *value = c->optDrainBelow;
  else if (!strcmp(name, "analytic_queue")) *value = c->optAnalyticQueue;
  else if (!strcmp(name, "auto_packet")) *value = c->optAutoPacket;
  else if (!strcmp(name, "kernel_variant_used")) *value = c->lastVariant;
  else if (!strcmp(name, "shadow_rule")) *value = c->optShadowRule;
  else if (!strcmp(name, "counted_span_us")) *value = c->countedSpanUs;
  else if (!strcmp(name, "counted_tail_us")) *value = c->countedTailUs;
  else if (!strcmp(name, "path_slots")) *value = packetkernel_slots();
  else if (!strcmp(name, "comm_ranks")) *value = c->comm ? c->commRanks : 0;
  else if (!strcmp(name, "num_cus")) *value = c->numCUs;
  else return fail(c, MOPTIX_ERR_INVALID, std::string("unknown option: ") + name);
  return MOPTIX_OK;
}

int moptix_accum_read(moptix_context c, float* dstHost) {
  if (!c || !dstHost) return fail(c, MOPTIX_ERR_INVALID, "null argument");
  if (!c->haveParams) return fail(c, MOPTIX_ERR_STATE, "no params");
  HIPCHK(c, hipSetDevice(c->device), "hipSetDevice");
  int rc = ensure_accum(c);
  if (rc != MOPTIX_OK) return rc;
  HIPCHK(c, hipMemcpyAsync(dstHost, accum_ptr(c), sizeof(float) * 3 * c->accumPixels, hipMemcpyDeviceToHost, c->stream), "read accuBuffer");
  HIPCHK(c, hipStreamSynchronize(c->stream), "sync");
  return MOPTIX_OK;
}

int moptix_accum_clear(moptix_context c) {
  if (!c) return MOPTIX_ERR_INVALID;
  if (!c->haveParams) return fail(c, MOPTIX_ERR_STATE, "no params");
  HIPCHK(c, hipSetDevice(c->device), "hipSetDevice");
  int rc = ensure_accum(c);
  if (rc != MOPTIX_OK) return rc;
  HIPCHK(c, hipMemsetAsync(accum_ptr(c), 0, sizeof(float) * 3 * c->accumPixels, c->stream), "clear accuBuffer");
  HIPCHK(c, hipStreamSynchronize(c->stream), "sync");
  return MOPTIX_OK;
}

int moptix_accum_device_ptr(moptix_context c, void** devPtr) {
  if (!c || !devPtr) return MOPTIX_ERR_INVALID;
  if (!c->haveParams) return fail(c, MOPTIX_ERR_STATE, "no params");
  HIPCHK(c, hipSetDevice(c->device), "hipSetDevice");
  int rc = ensure_accum(c);
  if (rc != MOPTIX_OK) return rc;
  HIPCHK(c, hipStreamSynchronize(c->stream), "sync");
  *devPtr = accum_ptr(c);
  return MOPTIX_OK;
}

int moptix_accum_bind(moptix_context c, void* devPtr) {
  if (!c) return MOPTIX_ERR_INVALID;
  c->accumBound = (float*)devPtr;
  c->accumPixels = 0;
  return MOPTIX_OK;
}

int moptix_resolve_rgb8(moptix_context c, float nAccumulation, int clearBuffer, uint8_t* dstHost) {
  if (!c || !dstHost) return fail(c, MOPTIX_ERR_INVALID, "null argument");
  if (!c->haveParams) return fail(c, MOPTIX_ERR_STATE, "no params");
  HIPCHK(c, hipSetDevice(c->device), "hipSetDevice");
  int rc = ensure_accum(c);
  if (rc != MOPTIX_OK) return rc;
  const size_t bytes = 3 * c->accumPixels;
  HIPCHK(c, c->dRgb8.ensure(bytes), "alloc rgb8");
  HIPCHK(c, launch_resolve_rgb8(c->stream, accum_ptr(c), (int)c->params.width, (int)c->params.height, nAccumulation, clearBuffer, c->dRgb8.p), "resolve kernel");
  HIPCHK(c, hipMemcpyAsync(dstHost, c->dRgb8.p, bytes, hipMemcpyDeviceToHost, c->stream), "read rgb8");
  HIPCHK(c, hipStreamSynchronize(c->stream), "sync");
  return MOPTIX_OK;
}

int moptix_kernel_time(moptix_context c, double* totalMs, uint64_t* nLaunches, int reset) {
  if (!c) return MOPTIX_ERR_INVALID;
  if (totalMs) *totalMs = c->kernelMs;
  if (nLaunches) *nLaunches = c->nLaunches;
  if (reset) { c->kernelMs = 0.0; c->reduceMs = 0.0; c->nLaunches = 0; }
  return MOPTIX_OK;
}

int moptix_reduce_time(moptix_context c, double* totalMs) {
  if (!c || !totalMs) return MOPTIX_ERR_INVALID;
  *totalMs = c->reduceMs;
  return MOPTIX_OK;
}

int moptix_debug_read_nodes64(moptix_context c, void* nodes64) {
  if (!c || !nodes64) return MOPTIX_ERR_INVALID;
  if (!c->accelBuilt) return fail(c, MOPTIX_ERR_STATE, "no acceleration structure");
  HIPCHK(c, hipSetDevice(c->device), "hipSetDevice");
  if (c->bvh.nNodes > 0 && !c->bvh.nodes64) return fail(c, MOPTIX_ERR_STATE, "this tree has no 64-byte form (a node is wider than 1e10 units)");
  if (c->bvh.nNodes > 0) HIPCHK(c, hipMemcpy(nodes64, c->bvh.nodes64, sizeof(Node64) * c->bvh.nNodes, hipMemcpyDeviceToHost), "read nodes64");
  return MOPTIX_OK;
}

int moptix_debug_read_accel(moptix_context c, void* nodes, void* tris, int32_t* triPrimIds) {
  if (!c) return MOPTIX_ERR_INVALID;
  if (!c->accelBuilt) return fail(c, MOPTIX_ERR_STATE, "no acceleration structure");
  HIPCHK(c, hipSetDevice(c->device), "hipSetDevice");
  if (nodes && c->bvh.nNodes > 0) HIPCHK(c, hipMemcpy(nodes, c->bvh.nodes, sizeof(Node128) * c->bvh.nNodes, hipMemcpyDeviceToHost), "read nodes");
  if ((tris || triPrimIds) && c->bvh.nTris > 0) {
    std::vector<Tri48> h(c->bvh.nTris);
    HIPCHK(c, hipMemcpy(h.data(), c->bvh.tris, sizeof(Tri48) * c->bvh.nTris, hipMemcpyDeviceToHost), "read tris");
    if (tris) memcpy(tris, h.data(), sizeof(Tri48) * h.size());
    if (triPrimIds) for (size_t i = 0; i < h.size(); i++) triPrimIds[i] = h[i].prim;
  }
  return MOPTIX_OK;
}

int moptix_debug_trace(moptix_context c, const float* rays, int32_t n, float* outT, int32_t* outPrim) {
  int rc = check_ready(c);
  if (rc != MOPTIX_OK) return rc;
  if (n < 0 || (n > 0 && (!rays || !outT || !outPrim))) return fail(c, MOPTIX_ERR_INVALID, "bad argument");
  if (n == 0) return MOPTIX_OK;
  HIPCHK(c, hipSetDevice(c->device), "hipSetDevice");
  float *dR = nullptr, *dT = nullptr; int* dP = nullptr; int* dOvf = nullptr;
  hipError_t e = hipMalloc((void**)&dR, sizeof(float) * 8 * (size_t)n);
  if (e == hipSuccess) e = hipMalloc((void**)&dT, sizeof(float) * (size_t)n);
  if (e == hipSuccess) e = hipMalloc((void**)&dP, sizeof(int) * (size_t)n);
  if (e == hipSuccess && c->bvh.stackBound > megakernel_lds_stack_entries()) {
    const size_t threads = ((size_t)n + 255) / 256 * 256;
    e = hipMalloc((void**)&dOvf, sizeof(int) * threads * (size_t)(c->bvh.stackBound - megakernel_lds_stack_entries() + 1));
  }
  SceneView v; fill_view(c, v);
  if (e == hipSuccess) e = hipMemcpyAsync(dR, rays, sizeof(float) * 8 * (size_t)n, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = launch_debug_trace(c->stream, v, dR, n, dT, dP, dOvf);
  if (e == hipSuccess) e = hipMemcpyAsync(outT, dT, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(outPrim, dP, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  if (dR) (void)hipFree(dR);
  if (dT) (void)hipFree(dT);
  if (dP) (void)hipFree(dP);
  if (dOvf) (void)hipFree(dOvf);
  if (e != hipSuccess) return hipFail(c, e, "debug trace");
  return MOPTIX_OK;
}

// ---- multi-GPU collectives: one process per GPU, RCCL over xGMI ---------------------------------

int moptix_comm_unique_id(uint8_t* id128) {
  if (!id128) return fail(nullptr, MOPTIX_ERR_INVALID, "null id");
  static_assert(sizeof(ncclUniqueId) == MOPTIX_COMM_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId id;
  if (!rccl().ok) return fail(nullptr, MOPTIX_ERR_STATE, rccl().error);
  ncclResult_t r = rccl().GetUniqueId(&id);
  if (r != ncclSuccess) return ncclFail(nullptr, r, "ncclGetUniqueId");
  memcpy(id128, &id, sizeof(id));
  return MOPTIX_OK;
}

int moptix_comm_init(moptix_context c, const uint8_t* id128, int32_t rank, int32_t nRanks) {
  if (!c || !id128 || nRanks < 1 || rank < 0 || rank >= nRanks) return fail(c, MOPTIX_ERR_INVALID, "bad communicator arguments");
  RCCL_READY(c);
  HIPCHK(c, hipSetDevice(c->device), "hipSetDevice");
  if (c->comm) { (void)rccl().CommDestroy(c->comm); c->comm = nullptr; }
  ncclUniqueId id; memcpy(&id, id128, sizeof(id));
  c->commNonBlocking = false;
  // Non-blocking where the library can do it and can also be polled and aborted ("comm_blocking" = 1 forces the plain form): every later
  // call then returns at once, ncclInProgress while the library still works on it, and comm_settle polls that state against
  // "comm_timeout_ms" -- a peer that is alive but never calls can no longer hold this rank inside ncclGroupEnd / ncclSend for good.
  if (!c->optCommBlocking && rccl().CommInitRankConfig && rccl().CommGetAsyncError && rccl().CommAbort) {
    ncclConfig_t cfg = NCCL_CONFIG_INITIALIZER;
    cfg.blocking = 0;
    const ncclResult_t r = rccl().CommInitRankConfig(&c->comm, nRanks, id, rank, &cfg);
    c->commNonBlocking = true; c->commRank = rank; c->commRanks = nRanks;
    const int rc = comm_settle(c, r, "ncclCommInitRankConfig");
    if (rc != MOPTIX_OK) { c->commNonBlocking = false; c->commRank = 0; c->commRanks = 1; return rc; }
    return MOPTIX_OK;
  }
  NCCLCHK(c, rccl().CommInitRank(&c->comm, nRanks, id, rank), "ncclCommInitRank");
  c->commRank = rank; c->commRanks = nRanks;
  return MOPTIX_OK;
}

int moptix_comm_destroy(moptix_context c) {
  if (!c) return MOPTIX_ERR_INVALID;
  if (c->comm) { HIPCHK(c, hipSetDevice(c->device), "hipSetDevice"); (void)hipStreamSynchronize(c->stream); NCCLCHK(c, rccl().CommDestroy(c->comm), "ncclCommDestroy"); c->comm = nullptr; }
  c->commRank = 0; c->commRanks = 1;
  return MOPTIX_OK;
}

int moptix_packed_tile_floats(moptix_context c, int32_t nRanks, uint64_t* out) {
  if (!c || !out) return MOPTIX_ERR_INVALID;
  TileDeal d; int rc = tile_deal(c, 0, nRanks, d);
  if (rc != MOPTIX_OK) return rc;
  *out = 3ull * (uint64_t)d.nItems;
  return MOPTIX_OK;
}

int moptix_pack_tiles(moptix_context c, int32_t rank, int32_t nRanks, float* dstDevice) {
  if (!c || !dstDevice) return fail(c, MOPTIX_ERR_INVALID, "null argument");
  TileDeal d; int rc = tile_deal(c, rank, nRanks, d);
  if (rc != MOPTIX_OK) return rc;
  HIPCHK(c, hipSetDevice(c->device), "hipSetDevice");
  if ((rc = ensure_accum(c)) != MOPTIX_OK) return rc;
  k_pack_tiles<<<dim3((d.nItems + 255) / 256), dim3(256), 0, c->stream>>>(accum_ptr(c), dstDevice, d);
  HIPCHK(c, hipGetLastError(), "pack tiles");
  HIPCHK(c, hipStreamSynchronize(c->stream), "sync");
  return MOPTIX_OK;
}

int moptix_unpack_tiles(moptix_context c, int32_t rank, int32_t nRanks, const float* srcDevice) {
  if (!c || !srcDevice) return fail(c, MOPTIX_ERR_INVALID, "null argument");
  TileDeal d; int rc = tile_deal(c, rank, nRanks, d);
  if (rc != MOPTIX_OK) return rc;
  HIPCHK(c, hipSetDevice(c->device), "hipSetDevice");
  if ((rc = ensure_accum(c)) != MOPTIX_OK) return rc;
  k_unpack_tiles<<<dim3((d.nItems + 255) / 256), dim3(256), 0, c->stream>>>(srcDevice, accum_ptr(c), d);
  HIPCHK(c, hipGetLastError(), "unpack tiles");
  HIPCHK(c, hipStreamSynchronize(c->stream), "sync");
  return MOPTIX_OK;
}

int moptix_gather_tiles(moptix_context c, int32_t dstRank) {
  if (!c) return MOPTIX_ERR_INVALID;
  if (!c->comm) return fail(c, MOPTIX_ERR_STATE, "moptix_comm_init has not been called");
  if (c->nRanks != c->commRanks || c->rank != c->commRank) return fail(c, MOPTIX_ERR_STATE, "moptix_set_partition does not match the communicator's rank / size");
  if (dstRank < 0 || dstRank >= c->commRanks) return fail(c, MOPTIX_ERR_INVALID, "bad destination rank");
  int rc = moptix_sync(c);
  if (rc != MOPTIX_OK) return rc;
  TileDeal d;
  if ((rc = tile_deal(c, c->rank, c->nRanks, d)) != MOPTIX_OK) return rc;
  HIPCHK(c, hipSetDevice(c->device), "hipSetDevice");
  if ((rc = ensure_accum(c)) != MOPTIX_OK) return rc;
  const size_t cnt = 3 * (size_t)d.nItems;                   // the same on every rank: whole groups of nRanks tiles
  const int n = c->commRanks;
  if (n == 1) return MOPTIX_OK;                              // the frame is already in place
  const dim3 grid((d.nItems + 255) / 256), block(256);
  if (c->rank != dstRank) {
    HIPCHK(c, c->dTileSend.ensure(cnt), "alloc tile staging");
    k_pack_tiles<<<grid, block, 0, c->stream>>>(accum_ptr(c), c->dTileSend.p, d);
    HIPCHK(c, hipGetLastError(), "pack tiles");
    if ((rc = comm_settle(c, rccl().Send(c->dTileSend.p, cnt, ncclFloat, dstRank, c->comm, c->stream), "ncclSend")) != MOPTIX_OK) return rc;
  } else {
    HIPCHK(c, c->dTileRecv.ensure(cnt * (size_t)n), "alloc tile staging");
    NCCLCHK(c, rccl().GroupStart(), "ncclGroupStart");
    ncclResult_t recvErr = ncclSuccess;
    for (int r = 0; r < n && (recvErr == ncclSuccess || recvErr == ncclInProgress); r++)
      if (r != dstRank) recvErr = rccl().Recv(c->dTileRecv.p + cnt * (size_t)r, cnt, ncclFloat, r, c->comm, c->stream);
    const ncclResult_t endErr = rccl().GroupEnd();           // always: a group left open would swallow every later call of this thread
    if (recvErr != ncclSuccess && recvErr != ncclInProgress) return ncclFail(c, recvErr, "ncclRecv");
    // the receives are on the stream only once the group has settled (non-blocking communicator): the unpack kernels go behind them
    if ((rc = comm_settle(c, endErr, "ncclGroupEnd")) != MOPTIX_OK) return rc;
    for (int r = 0; r < n; r++) {                            // the other ranks' tiles into this rank's accuBuffer
      if (r == dstRank) continue;
      TileDeal dr = d; dr.rank = r;
      k_unpack_tiles<<<grid, block, 0, c->stream>>>(c->dTileRecv.p + cnt * (size_t)r, accum_ptr(c), dr);
    }
    HIPCHK(c, hipGetLastError(), "unpack tiles");
  }
  return comm_wait(c, "moptix_gather_tiles");
}

int moptix_reduce_frame(moptix_context c, int32_t dstRank) {
  if (!c) return MOPTIX_ERR_INVALID;
  if (!c->comm) return fail(c, MOPTIX_ERR_STATE, "moptix_comm_init has not been called");
  if (dstRank < 0 || dstRank >= c->commRanks) return fail(c, MOPTIX_ERR_INVALID, "bad destination rank");
  int rc = moptix_sync(c);
  if (rc != MOPTIX_OK) return rc;
  HIPCHK(c, hipSetDevice(c->device), "hipSetDevice");
  if ((rc = ensure_accum(c)) != MOPTIX_OK) return rc;
  if (c->commRanks > 1 && (rc = comm_settle(c, rccl().Reduce(accum_ptr(c), accum_ptr(c), 3 * c->accumPixels, ncclFloat, ncclSum, dstRank, c->comm, c->stream), "ncclReduce")) != MOPTIX_OK)
    return rc;
  return c->commRanks > 1 ? comm_wait(c, "moptix_reduce_frame") : moptix_sync(c);
}

}  // extern "C"
