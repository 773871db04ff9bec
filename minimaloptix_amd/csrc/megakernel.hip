// megakernel.hip -- the path-tracing megakernel for gfx950 (CDNA4, wave64).
//
// Replaces context->launch(0,W,H) x nSuperSampling (MinimalOptiX.cpp:544-546) and everything
// OptiX runs inside it: Camera.cu raygen, rtTrace + Trbvh/NoAccel traversal, the Geometry.cu
// intersectors, the Material.cu/disney.h closest-hit + any-hit programs and miss.cu.
//
// Execution model (DESIGN.md "Megakernel"):
//   * persistent grid: blocksPerCU x 256 CUs workgroups of 256 threads (4 waves); lanes pull
//     pixels from one global work counter (8x8-pixel tiles in raster order, so one wave's
//     first 64 pixels form a tile) -- every lane always owns one path;
//   * a lane keeps its pixel for all nSeeds samples and adds them in launch order in
//     registers (bit-identical to nSeeds separate launches; one accumulator read+write per
//     pixel instead of one per sample);
//   * per wave the loop alternates  [A] path state machine (shade / regenerate / next ray)
//     and [B] BVH traversal.  The traversal state is resumable, and the wave leaves [B] as
//     soon as fewer than `exitThreshold` lanes are still traversing while others wait for
//     shading -- wave-level compaction of the ray population in time rather than by moving
//     rays between lanes;
//   * traversal stack: 32 entries per lane in LDS, laid out [entry][lane] so a push/pop is a
//     conflict-free ds_write_b32/ds_read_b32; deeper trees spill to a global overflow area.
#include <hip/hip_runtime.h>

#include "megakernel.h"
#include "pt_path.h"

namespace pt {

namespace {

constexpr int kBlockThreads = 256;
constexpr int kWavesPerBlock = kBlockThreads / 64;
constexpr int kLdsStack = 32;          // entries per lane kept in LDS

// LDS stack with global overflow.  lds points at this lane's column ([entry][lane] layout).
struct LaneStack {
  int* lds;
  int* ovf;       // this lane's overflow column (stride = ovfStride) or nullptr
  int ovfStride;
  __device__ __forceinline__ void store(int sp, int v) {
    if (sp < kLdsStack) lds[sp * 64] = v;
    else ovf[(size_t)(sp - kLdsStack) * ovfStride] = v;
  }
  __device__ __forceinline__ int load(int sp) const {
    return sp < kLdsStack ? lds[sp * 64] : ovf[(size_t)(sp - kLdsStack) * ovfStride];
  }
  __device__ __forceinline__ bool roomy(int sp) const { return sp + 3 <= kLdsStack; }
  __device__ __forceinline__ void store_fast(int sp, int v) { lds[sp * 64] = v; }
  static constexpr bool kFlat = false;      // pt_path.h node_step_nearfar: this stack takes the branched tail
  __device__ __forceinline__ bool fits_fast(int, int) const { return false; }
  __device__ __forceinline__ int peek_fast(int) const { return 0; }
};
struct NoStack {
  __device__ __forceinline__ void store(int, int) {}
  __device__ __forceinline__ int load(int) const { return kTravDone; }
  __device__ __forceinline__ bool roomy(int) const { return false; }
  __device__ __forceinline__ void store_fast(int, int) {}
  static constexpr bool kFlat = false;      // pt_path.h node_step_nearfar: this stack takes the branched tail
  __device__ __forceinline__ bool fits_fast(int, int) const { return false; }
  __device__ __forceinline__ int peek_fast(int) const { return 0; }
};

__device__ __forceinline__ int popc64(unsigned long long m) { return __popcll(m); }

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

template <bool CNT, bool HAS_TRIS>
__global__ void __launch_bounds__(kBlockThreads) pt_megakernel(const LaunchArgs a) {
  __shared__ int ldsStack[HAS_TRIS ? kWavesPerBlock * kLdsStack * 64 : 1];
  const SceneView& sc = a.scene;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

  using Stack = typename std::conditional<HAS_TRIS, LaneStack, NoStack>::type;
  Stack st;
  if constexpr (HAS_TRIS) {
    st.lds = ldsStack + wave * (kLdsStack * 64) + lane;
    const int gthread = blockIdx.x * kBlockThreads + threadIdx.x;
    st.ovfStride = gridDim.x * kBlockThreads;
    st.ovf = a.stackOverflow ? a.stackOverflow + gthread : nullptr;
  }

  PathState ps;
  ps.mode = M_NEW_PIXEL; ps.pixel = 0; ps.item = 0; ps.accum = mk3(0, 0, 0);
  ps.thr = mk3(0, 0, 0); ps.rad = mk3(0, 0, 0); ps.depth = 0; ps.seed = 0;
  ps.o = mk3(0, 0, 0); ps.d = mk3(0, 0, 1); ps.tmin = 0; ps.tmax = 0; ps.kind = RK_RADIANCE;
  ps.N = mk3(0, 0, 1); ps.V = mk3(0, 0, 1); ps.mat = 0; ps.light = 0; ps.pendW = mk3(0, 0, 0); ps.pendInv = 0;
  Trav tv;
  tv.node = kTravDone; tv.sp = 0; tv.started = 0; tv.tbest = 0; tv.bestPrim = -1; tv.bestTri = -1;
  tv.beta = 0; tv.gamma = 0; tv.att = mk3(1, 1, 1); tv.inv = mk3(0, 0, 0); tv.noi = mk3(0, 0, 0);
  Counters ct = {};
  uint32_t waveSteps = 0, activeLaneSteps = 0;

  for (;;) {
    // ---- [A] path state machine: run until this lane owns a ray again or is out of work ----
    while (ps.mode != M_TRACE && ps.mode != M_DONE) {
      if (ps.mode == M_RESULT) {
        on_result<CNT>(sc, ps, tv, ct);
      } else if (ps.mode == M_LIGHTS) {
        on_lights<CNT>(sc, ps, ct);
      } else if (ps.mode == M_NEW_SAMPLE) {
        store_sample(a, ps.item, ps.accum);                          // the finished sample -> per-sample buffer
        ps.mode = M_NEW_PIXEL;
      } else {  // M_NEW_PIXEL: next (pixel, sample) work item
        const int k = atomicAdd(a.workCounter, 1);     // hipcc aggregates this per wave
        int s;
        if (k >= a.nWork) { ps.mode = M_DONE; }
        else if (item_to_pixel(a, k, s, ps.pixel)) { ps.item = k; begin_sample<CNT>(sc, ps, a.seeds[s], ct); }
      }
    }
    if (__ballot(ps.mode != M_DONE) == 0ull) break;

    // ---- [B] traversal ----
    if (ps.mode == M_TRACE && !tv.started) trav_begin<CNT>(sc, ps, tv, ct);
    if constexpr (HAS_TRIS) {
      for (;;) {
        const bool active = (ps.mode == M_TRACE) & (tv.node != kTravDone);
        const unsigned long long am = __ballot(active);
        if (am == 0ull) break;
        const int nActive = popc64(am);
        if (nActive < a.exitThreshold) {
          // leave only if somebody is actually waiting to be shaded / regenerated
          const unsigned long long wm = __ballot((ps.mode == M_TRACE) & (tv.node == kTravDone));
          if (wm != 0ull) break;
        }
        // while-while: node steps until enough lanes are parked at a leaf, then one leaf pass
        const bool atNode = active & (tv.node >= 0);
        const unsigned long long nm = __ballot(atNode);
        const int nLeaf = nActive - popc64(nm);
        if (nm != 0ull && nLeaf < a.leafThreshold) {
          if (CNT) { waveSteps++; activeLaneSteps += (uint32_t)popc64(nm); }
          if (atNode) trav_node_step<CNT>(sc, ps, tv, st, ct);
        } else {
          if (CNT) { waveSteps++; activeLaneSteps += (uint32_t)nLeaf; }
          if (active & (tv.node < 0)) trav_leaf_step<CNT>(sc, ps, tv, st, ct);
        }
      }
    } else {
      tv.node = kTravDone;
    }
    if (ps.mode == M_TRACE && tv.node == kTravDone) { ps.mode = M_RESULT; tv.started = 0; }
  }

  if constexpr (CNT) {
    unsigned long long* c = a.counters;
    const uint32_t v[9] = { wave_sum(ct.samples), wave_sum(ct.primaryRays), wave_sum(ct.bounceRays), wave_sum(ct.shadowRays),
                            wave_sum(ct.nodeFetches), wave_sum(ct.triTests), wave_sum(ct.closestHits), wave_sum(ct.lightLoads),
                            wave_sum(ct.analyticTests) };
    if (lane == 0) {
      for (int i = 0; i < 9; i++) atomicAdd(&c[i], (unsigned long long)v[i]);
      atomicAdd(&c[9], (unsigned long long)waveSteps);
      atomicAdd(&c[10], (unsigned long long)activeLaneSteps);
    }
  }
}

// nearest-hit queries for explicit rays (moptix_debug_trace)
__global__ void __launch_bounds__(kBlockThreads) k_debug_trace(SceneView sc, const float* rays, int n, float* outT, int* outPrim, int* stackOverflow) {
  __shared__ int ldsStack[kWavesPerBlock * kLdsStack * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  LaneStack st;
  st.lds = ldsStack + wave * (kLdsStack * 64) + lane;
  st.ovfStride = gridDim.x * kBlockThreads;
  st.ovf = stackOverflow ? stackOverflow + i : nullptr;
  if (i >= n) return;
  PathState ps = {};
  ps.o = mk3(rays[8 * i], rays[8 * i + 1], rays[8 * i + 2]); ps.d = mk3(rays[8 * i + 3], rays[8 * i + 4], rays[8 * i + 5]);
  ps.tmin = rays[8 * i + 6]; ps.tmax = rays[8 * i + 7]; ps.kind = RK_RADIANCE; ps.mode = M_TRACE;
  Trav tv = {}; Counters ct = {};
  trav_begin<false>(sc, ps, tv, ct);
  while (tv.node != kTravDone) trav_step<false>(sc, ps, tv, st, ct);
  outT[i] = tv.tbest; outPrim[i] = tv.bestPrim;
}

// Probe of a node format (first render after a build, option node_format = 0): the scene's own paths.  One thread per pixel
// of a coarse grid over the camera's view (sc.width x sc.height here are the grid's, so the primary rays fan out over the same
// frustum) runs the per-ray path state machine for one sample -- camera ray, closest hit, three shadow rays, bounce -- until
// the path ends or is kProbeDepth deep, walking the tree in the format under test.  out[0] += node steps, out[1] += triangle
// tests.  Counts, not times: the same scene, camera and parameters give the same verdict on every run.
constexpr int kProbeDepth = 6;
template <bool N64>
__global__ void __launch_bounds__(kBlockThreads) k_probe_paths(SceneView sc, int launchSeed, unsigned long long* out, int* stackOverflow) {
  __shared__ int ldsStack[kWavesPerBlock * kLdsStack * 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  LaneStack st;
  st.lds = ldsStack + wave * (kLdsStack * 64) + lane;
  st.ovfStride = gridDim.x * kBlockThreads;
  st.ovf = stackOverflow ? stackOverflow + i : nullptr;
  Counters ct = {};
  if (i < sc.width * sc.height) {
    PathState ps = {};
    ps.pixel = i; ps.item = 0; ps.accum = mk3(0, 0, 0); ps.N = mk3(0, 0, 1); ps.V = mk3(0, 0, 1);
    Trav tv = {};
    tv.node = kTravDone; tv.bestPrim = -1; tv.bestTri = -1; tv.att = mk3(1, 1, 1);
    begin_sample<true>(sc, ps, launchSeed, ct);
    while (ps.mode != M_DONE && ps.mode != M_NEW_SAMPLE && ps.depth <= kProbeDepth) {
      if (ps.mode == M_TRACE) {
        trav_begin<true>(sc, ps, tv, ct);
        while (tv.node != kTravDone) trav_step<true, N64>(sc, ps, tv, st, ct);
        ps.mode = M_RESULT;
      } else if (ps.mode == M_RESULT) on_result<true>(sc, ps, tv, ct);
      else if (ps.mode == M_LIGHTS) on_lights<true>(sc, ps, ct);
      else break;
    }
  }
  const uint32_t nf = wave_sum(ct.nodeFetches), tt = wave_sum(ct.triTests);
  if (lane == 0) { atomicAdd(&out[0], (unsigned long long)nf); atomicAdd(&out[1], (unsigned long long)tt); }
}

// updateContent (MinimalOptiX.cpp:43-66): normalise, clamp, flip rows, 8-bit, optional clear.
// QColor::setRedF stores qRound(v*65535) and QImage::Format_RGB888 keeps its high byte.
__global__ void k_resolve_rgb8(float* accum, int width, int height, float nAccumulation, int clearBuffer, uint8_t* out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= width * height) return;
  const int i = idx / width, j = idx % width;
  float* src = accum + 3 * (size_t)idx;
  uint8_t* dst = out + 3 * ((size_t)(height - i - 1) * width + j);
  for (int c = 0; c < 3; c++) {
    const float v = clampf(src[c] / nAccumulation, 0.f, 1.f);
    dst[c] = (uint8_t)(((uint32_t)(v * 65535.0f + 0.5f)) >> 8);
    if (clearBuffer) src[c] = 0.0f;
  }
}

}  // namespace

int megakernel_lds_stack_entries() { return kLdsStack; }

hipError_t launch_megakernel(hipStream_t stream, const LaunchArgs& a, int nBlocks, bool counted) {
  const bool hasTris = a.scene.rootRef != kEmptyRef;
  dim3 grid(nBlocks), block(kBlockThreads);
  if (counted) {
    if (hasTris) pt_megakernel<true, true><<<grid, block, 0, stream>>>(a);
    else         pt_megakernel<true, false><<<grid, block, 0, stream>>>(a);
  } else {
    if (hasTris) pt_megakernel<false, true><<<grid, block, 0, stream>>>(a);
    else         pt_megakernel<false, false><<<grid, block, 0, stream>>>(a);
  }
  return hipGetLastError();
}

hipError_t launch_debug_trace(hipStream_t stream, const SceneView& sc, const float* dRays, int n, float* dT, int* dPrim, int* stackOverflow) {
  const int blocks = (n + kBlockThreads - 1) / kBlockThreads;
  k_debug_trace<<<blocks, kBlockThreads, 0, stream>>>(sc, dRays, n, dT, dPrim, stackOverflow);
  return hipGetLastError();
}

hipError_t launch_probe_paths(hipStream_t stream, const SceneView& sc, int launchSeed, bool node64, unsigned long long* dOut, int* stackOverflow) {
  const int blocks = (sc.width * sc.height + kBlockThreads - 1) / kBlockThreads;
  if (node64) k_probe_paths<true><<<blocks, kBlockThreads, 0, stream>>>(sc, launchSeed, dOut, stackOverflow);
  else        k_probe_paths<false><<<blocks, kBlockThreads, 0, stream>>>(sc, launchSeed, dOut, stackOverflow);
  return hipGetLastError();
}

// Camera.cu:41 for every launch of the batch: accu[pixel] += sample, strictly in launch order,
// so the result is bit-identical to nSeeds separate one-sample launches.
__global__ void k_reduce_samples(LaunchArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.nItems) return;
  if (a.workCounter[1] != 0) return;      // the trace kernel gave up (watchdog): its sample buffer is incomplete
  int s, pixel;
  if (!item_to_pixel(a, i, s, pixel)) return;
  float* px = a.accum + 3 * (size_t)pixel;
  v3 acc = mk3(px[0], px[1], px[2]);
  for (int k = 0; k < a.nSeeds; k++) {
    const float* sp = a.sampleBuf + 3 * ((size_t)k * a.nItems + i);
    acc = acc + mk3(sp[0], sp[1], sp[2]);
  }
  px[0] = acc.x; px[1] = acc.y; px[2] = acc.z;
}

hipError_t launch_reduce_samples(hipStream_t stream, const LaunchArgs& a) {
  k_reduce_samples<<<(a.nItems + 255) / 256, 256, 0, stream>>>(a);
  return hipGetLastError();
}

hipError_t launch_resolve_rgb8(hipStream_t stream, float* accum, int width, int height, float nAccumulation, int clearBuffer, uint8_t* dOut) {
  const int n = width * height;
  k_resolve_rgb8<<<(n + 255) / 256, 256, 0, stream>>>(accum, width, height, nAccumulation, clearBuffer, dOut);
  return hipGetLastError();
}

}  // namespace pt
