// megakernel.h -- launch interface of megakernel.hip
#pragma once
#include <hip/hip_runtime.h>
#include "pt_types.h"

namespace pt {

struct LaunchArgs {
  SceneView scene;
  const int* seeds; int nSeeds;     // launch seeds, one sample per pixel each (device memory)
  float* accum;                     // accuBuffer: float3 W*H, row 0 = bottom
  float* sampleBuf;                 // per-sample results: float3 [nSeeds][nItems]
  int* workCounter;                 // [0] global work-item counter, [1] watchdog flag (both zeroed before the launch); variant 4: the drain list behind them (kDrain*)
  int nItems;                       // pixels-slots of this rank = local tiles * 64
  int nWork;                        // work items = nSeeds * nItems, item k = (sample k / nItems, slot k % nItems)
  int tilesX; int rank, nRanks;     // 8x8 tile grid + tile-interleaved partition
  int exitThreshold;                // leave the traversal loop below this many active lanes
  int leafThreshold;                // run the leaf pass once this many lanes are parked at a leaf
  int* stackOverflow;               // per-thread spill area for trees deeper than the LDS stack (or null)
  unsigned long long* counters;     // 16 x u64 (counting build only)
  // variants 3 and 4 (queuekernel.hip, packetkernel.hip)
  void* poolCold;                   // path-slot records in HBM
  int starveLanes;                  // run a partial batch once this many lanes of the wave have nothing to traverse
  int swapLanes;                    // leave the node loop once this many lanes stand at a leaf / have finished
  int ovfDepth;                     // ints of stack overflow per slot
  int slotsInUse;                   // path slots per pool a launch uses (0 = all); fewer slots = shorter critical path
  int auxDepth;                     // variant 4: paths this deep trace their shadow rays in borrowed slots, beside the continuation (0 = off)
  unsigned long long watchdogTicks; // a wave gives up after this many 100 MHz ticks (sets workCounter[1]; the pass is then not reduced)
  // hand-out order of the work items (queuekernel.hip): tile-major, tiles with the deepest paths first
  int tileMajor;                    // 0 = sample-major in raster tile order (item k handed out as k); 1, 2 = by tile; 3 = by pixel
  int unitShift;                    // log2 of the slots per history unit: 6 = 8x8 tile, 0 = pixel
  const int* tileOrder;             // unit visited i-th (device, nItems >> unitShift entries) or nullptr = raster order
  unsigned int* tileCost;           // per unit: deepest path seen so far (device) or nullptr
#ifdef PT_EVLOG
  unsigned long long* evLog;        // experiment build (packetkernel.hip PT_EV): [0] events so far, [1..] the events
#endif
};
constexpr int kDeepPath = 8;        // paths at least this deep are recorded in tileCost
constexpr int kDrainDeep = 24;      // paths at least this deep go to the front of the drain list
// The launch's last paths (packetkernel.hip -> drainkernel.hip): a workgroup of the packet kernel that is down to `below` paths hands each of them
// over at its next packet boundary -- the slot record in poolCold is complete then -- and leaves.  The list lives behind the work counter
// (LaunchArgs::workCounter + kDrainList) and NOT in LaunchArgs: sixteen more bytes of kernel arguments moved the packet kernel's register
// allocation and cost 0.8 % of the benchmark frame (NOTEBOOK.md round 6).
//   [kDrainDeepN] deep paths handed over (entries upwards from kDrainEntries), [kDrainNext] the drain kernel's hand-out counter,
//   [kDrainOtherN] the other paths (entries downwards from kDrainEntries + cap - 1), [kDrainCap] entries the list holds (workgroups x below),
//   [kDrainBelow] the threshold (0 = off); an entry = the path's slot record, as an index into poolCold
constexpr int kDrainList = 2, kDrainDeepN = 0, kDrainNext = 1, kDrainOtherN = 2, kDrainCap = 4, kDrainBelow = 5, kDrainEntries = 6;

#if defined(__HIPCC__)
// work item k -> (sample index, pixel).  Slot i = k % nItems is the (i & 63)-th pixel of this
// rank's (i >> 6)-th 8x8 tile.  Tiles are dealt to the ranks in raster order, nRanks at a time, and the deal
// rotates by one rank from each group of nRanks tiles to the next (tile-interleaved multi-GPU partition,
// SURVEY 8e): a plain t % nRanks gives every rank the same columns in every row when nRanks divides the
// tiles of a row (1920 / 8 = 240 tiles, 8 ranks), and the ranks' ray counts then differ by 3-4 %.
// False for pixels outside the frame (this includes the tiles past the end of a rank's last group).
__device__ __forceinline__ bool item_to_pixel(const LaunchArgs& a, int k, int& sample, int& pixel) {
  sample = k / a.nItems;
  const int i = k - sample * a.nItems;
  const int lt = i >> 6, in = i & 63;
  const int gt = lt * a.nRanks + (a.rank + lt) % a.nRanks;
  const int tx = gt % a.tilesX, ty = gt / a.tilesX;
  const int x = tx * 8 + (in & 7), y = ty * 8 + (in >> 3);
  pixel = y * a.scene.width + x;
  return (x < a.scene.width) & (y < a.scene.height);
}
// Hand-out index k -> canonical work item (the index item_to_pixel and the per-sample buffer use).  Tile-major:
// all samples of a tile are handed out back to back, tile after tile in tileOrder; the paths that bounce 256 times
// (a chain of ~1000 dependent rays, ~28 ms) then start early and overlap the bulk instead of forming the tail.
__device__ __forceinline__ int handout_to_item(const LaunchArgs& a, int k) {
  if (!a.tileMajor) return k;
  if (a.tileMajor == 3) {                               // all samples of a pixel back to back, deepest pixels first
    const int ui = k / a.nSeeds, sample = k - ui * a.nSeeds;
    return sample * a.nItems + (a.tileOrder ? a.tileOrder[ui] : ui);
  }
  const int per = a.nSeeds << 6;                       // items of one tile
  const int ti = k / per, r = k - ti * per;
  const int lt = a.tileOrder ? a.tileOrder[ti] : ti;
  if (a.tileMajor == 2) {                               // experiment: consecutive items = samples of ONE pixel
    const int in = r / a.nSeeds, sample = r - in * a.nSeeds;
    return sample * a.nItems + (lt << 6) + in;
  }
  return (r >> 6) * a.nItems + (lt << 6) + (r & 63);
}
// Camera.cu:39 result of one sample; Camera.cu:41 (the add) happens in k_reduce_samples
__device__ __forceinline__ void store_sample(const LaunchArgs& a, int item, v3 value) {
  float* sp = a.sampleBuf + 3 * (size_t)item;
  sp[0] = value.x; sp[1] = value.y; sp[2] = value.z;
}
#endif

int megakernel_lds_stack_entries();
hipError_t launch_reduce_samples(hipStream_t stream, const LaunchArgs& a);
hipError_t launch_megakernel(hipStream_t stream, const LaunchArgs& a, int nBlocks, bool counted);
int queuekernel_lds_stack_entries();
int queuekernel_slots();
size_t queuekernel_cold_bytes(int nBlocks);
size_t queuekernel_overflow_ints(int nBlocks, int ovfDepth);
hipError_t launch_queuekernel(hipStream_t stream, const LaunchArgs& a, int nBlocks, bool counted, bool fastShading);
// queuekernel_lean.hip: the same kernel with four workgroups per CU, for scenes without triangles
int queuekernel_lds_stack_entries_lean();
int queuekernel_slots_lean();
size_t queuekernel_cold_bytes_lean(int nBlocks);
size_t queuekernel_overflow_ints_lean(int nBlocks, int ovfDepth);
hipError_t launch_queuekernel_lean(hipStream_t stream, const LaunchArgs& a, int nBlocks, bool counted, bool fastShading);
int packetkernel_lds_stack_entries();
int packetkernel_slots();             // path slots per workgroup (variant 4)
size_t packetkernel_cold_bytes(int nBlocks);
size_t packetkernel_overflow_ints(int nBlocks, int ovfDepth);
hipError_t launch_packetkernel(hipStream_t stream, const LaunchArgs& a, int nBlocks, bool counted, bool fastShading);
// drainkernel.hip: finishes the paths the packet kernel's workgroups handed over (LaunchArgs::drainList); same stream, right after it
hipError_t launch_drainkernel(hipStream_t stream, const LaunchArgs& a, int nCUs, bool counted, bool fastShading);
size_t drain_list_ints(int nBlocks, int drainBelow);      // ints behind LaunchArgs::workCounter + kDrainList
hipError_t launch_debug_trace(hipStream_t stream, const SceneView& sc, const float* dRays, int n, float* dT, int* dPrim, int* stackOverflow);
// node steps and triangle tests of one sample per pixel of sc.width x sc.height, paths cut at depth 6, under one node format
// (dOut[0..1] += ; megakernel.hip k_probe_paths)
hipError_t launch_probe_paths(hipStream_t stream, const SceneView& sc, int launchSeed, bool node64, unsigned long long* dOut, int* stackOverflow);
hipError_t launch_resolve_rgb8(hipStream_t stream, float* accum, int width, int height, float nAccumulation, int clearBuffer, uint8_t* dOut);

}  // namespace pt
