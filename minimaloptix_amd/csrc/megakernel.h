// megakernel.h -- launch interface of megakernel.hip
#pragma once
#include <hip/hip_runtime.h>
#include "pt_types.h"

namespace pt {

struct LaunchArgs {
  SceneView scene;
  const int* seeds; int nSeeds;     // launch seeds, one sample per pixel each (device memory)
  float* accum;                     // accuBuffer: float3 W*H, row 0 = bottom
  int* workCounter;                 // global work-item counter (zeroed before the launch)
  int nWork;                        // work items of this rank = local tiles * 64
  int tilesX; int rank, nRanks;     // 8x8 tile grid + tile-interleaved partition
  int exitThreshold;                // leave the traversal loop below this many active lanes
  int* stackOverflow;               // per-thread spill area for trees deeper than the LDS stack (or null)
  unsigned long long* counters;     // 11 x u64 (counting build only)
};

int megakernel_lds_stack_entries();
hipError_t launch_megakernel(hipStream_t stream, const LaunchArgs& a, int nBlocks, bool counted);
hipError_t launch_debug_trace(hipStream_t stream, const SceneView& sc, const float* dRays, int n, float* dT, int* dPrim, int* stackOverflow);
hipError_t launch_resolve_rgb8(hipStream_t stream, float* accum, int width, int height, float nAccumulation, int clearBuffer, uint8_t* dOut);

}  // namespace pt
