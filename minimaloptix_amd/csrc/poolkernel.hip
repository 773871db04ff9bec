// poolkernel.hip -- megakernel variant 1: wave-level ray compaction through a path pool.
//
// Same per-path arithmetic as megakernel.hip (pt_path.h), different scheduling.  Profiling
// variant 0 on coffee showed ~18 % VALU lane utilisation: lanes that finished their ray idle
// until the wave leaves the traversal loop, and the shading code then runs for a handful of
// lanes.  Here every wave owns P > 64 path *slots*; the 64 lanes are workers:
//
//   Q_TRAV  slots whose next ray is ready          -> a lane pops one, traverses it to the end
//   Q_SHADE slots whose ray has finished with a hit / a shadow result -> closest-hit shading
//   Q_GEN   slots that need a new sample or pixel (miss, path ended, start-up)
//
//   * traversal always runs with (nearly) all 64 lanes busy: a lane whose ray finishes writes
//     the hit record into the slot, pushes the slot on Q_SHADE/Q_GEN and pops the next ready
//     ray from Q_TRAV (compaction of the live rays of 2..3 waves' worth of paths into one wave);
//   * shading / regeneration run as batches of up to 64 slots of the same stage (sorting of
//     the paths by the program they need), so the divergent material code runs on full waves.
//
// Slot state: the hot part (ray + hit record, 48 B) lives in LDS, the cold part (payload,
// Disney context, pixel bookkeeping, 112 B) in a per-wave region of HBM that only this wave
// touches.  Queues are 16-bit rings in LDS with wave-uniform head/count.  Everything is
// wave-synchronous: no workgroup barrier, no atomics except the global pixel counter.
// A slot owns its pixel for all nSeeds samples, added in launch order => bit-identical to
// variant 0 and to nSeeds separate launches.
#include <hip/hip_runtime.h>

#include "megakernel.h"
#include "pt_path.h"

namespace pt {

namespace {

constexpr int kBlockThreads = 256;
constexpr int kWaves = kBlockThreads / 64;
constexpr int kStackN = 16;             // LDS stack entries per lane; deeper levels spill to HBM
constexpr int kRing = 256;              // ring capacity (>= P, power of two)

enum { Q_TRAV = 0, Q_SHADE = 1, Q_GEN = 2 };

struct alignas(16) SlotCold {           // 6 x 16 B
  int mode, pixel, item, depth;
  uint32_t seed; float thrx, thry, thrz;
  float radx, rady, radz; int mat;
  float Nx, Ny, Nz; int light;
  float Vx, Vy, Vz; float pendInv;
  float pwx, pwy, pwz; float cdx;      // cd*: PathState::cdlin (textured materials)
  float cdy, cdz; int pad1, pad2;
};
static_assert(sizeof(SlotCold) == 112, "SlotCold layout");

struct LaneStack16 {
  int* lds; int* ovf; int ovfStride;
  __device__ __forceinline__ void store(int sp, int v) {
    if (sp < kStackN) lds[sp * 64] = v; else ovf[(size_t)(sp - kStackN) * ovfStride] = v;
  }
  __device__ __forceinline__ int load(int sp) const {
    return sp < kStackN ? lds[sp * 64] : ovf[(size_t)(sp - kStackN) * ovfStride];
  }
  __device__ __forceinline__ bool roomy(int sp) const { return sp + 3 <= kStackN; }
  __device__ __forceinline__ void store_fast(int sp, int v) { lds[sp * 64] = v; }
};

__device__ __forceinline__ int lane_rank(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// hit record kept in the slot: >= 0 triangle record index, < 0 (and != kMiss) ~analytic prim id
constexpr int kMiss = (int)0x80000000;

#ifndef PT_POOL_WAVES_PER_SIMD
#define PT_POOL_WAVES_PER_SIMD 1
#endif
template <bool CNT, int P>
__global__ void __launch_bounds__(kBlockThreads, PT_POOL_WAVES_PER_SIMD) pt_poolkernel(const LaunchArgs a) {
  static_assert(P <= kRing && (P % 64) == 0, "pool size");
  __shared__ v4 sHot[kWaves][3][P];
  __shared__ int sStack[kWaves][kStackN * 64];
  __shared__ unsigned short sQueue[kWaves][3][kRing];

  const SceneView& sc = a.scene;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v4* hot0 = sHot[wave][0]; v4* hot1 = sHot[wave][1]; v4* hot2 = sHot[wave][2];
  unsigned short (*queue)[kRing] = sQueue[wave];
  const int gwave = blockIdx.x * kWaves + wave;
  SlotCold* cold = reinterpret_cast<SlotCold*>(a.poolCold) + (size_t)gwave * P;

  LaneStack16 st;
  st.lds = sStack[wave] + lane;
  st.ovfStride = gridDim.x * kBlockThreads;
  st.ovf = a.stackOverflow ? a.stackOverflow + (blockIdx.x * kBlockThreads + threadIdx.x) : nullptr;

  // wave-uniform queue bookkeeping
  int qHead[3] = { 0, 0, 0 }, qCount[3] = { 0, 0, 0 };
  int nDone = 0;

  auto q_push = [&](int q, bool pred, int slot) {
    const unsigned long long m = __ballot(pred);
    if (m == 0ull) return;
    const int base = qHead[q] + qCount[q];
    if (pred) queue[q][(base + lane_rank(m)) & (kRing - 1)] = (unsigned short)slot;
    qCount[q] += __popcll(m);
  };
  // lanes with `want` receive a slot (or -1); at most qCount[q] are served, lowest lanes first
  auto q_pop = [&](int q, bool want) -> int {
    const unsigned long long m = __ballot(want);
    const int n = min(__popcll(m), qCount[q]);
    int slot = -1;
    if (want) { const int r = lane_rank(m); if (r < n) slot = queue[q][(qHead[q] + r) & (kRing - 1)]; }
    qHead[q] = (qHead[q] + n) & (kRing - 1);
    qCount[q] -= n;
    return slot;
  };

  // ---- start-up: every slot needs a pixel ----
  for (int s = lane; s < P; s += 64) {
    SlotCold c = {};
    c.mode = M_NEW_PIXEL;
    cold[s] = c;
    queue[Q_GEN][s] = (unsigned short)s;
  }
  qCount[Q_GEN] = P;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");

  // ---- traversal job of this lane ----
  int job = -1;                   // slot being traversed, -1 = idle
  PathState ray;                  // only o, d, tmin, tmax, kind are used here
  ray.mode = M_TRACE; ray.tmin = sc.epsT; ray.o = mk3(0, 0, 0); ray.d = mk3(0, 0, 1); ray.tmax = 0; ray.kind = RK_RADIANCE;
  Trav tv; tv.node = kTravDone; tv.sp = 0; tv.started = 0; tv.tbest = 0; tv.bestPrim = -1; tv.bestTri = -1;
  tv.beta = 0; tv.gamma = 0; tv.att = mk3(1, 1, 1); tv.inv = mk3(0, 0, 0); tv.noi = mk3(0, 0, 0);
  Counters ct = {};
  uint32_t leafPasses = 0, leafLanes = 0, waveSteps = 0, activeLaneSteps = 0, batchLanes = 0, batches = 0, dbgFull = 0, dbgIdle = 0, dbgWaiting = 0; unsigned long long tBatch = 0, tRefill = 0, tNode = 0, tLeaf = 0, tFin = 0, tStamp = 0; const unsigned long long tStart = CNT ? __builtin_amdgcn_s_memtime() : 0ull;
#define PT_STAMP(acc) do { if (CNT) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc += now_ - tStamp; tStamp = now_; } } while (0)
  tStamp = tStart;

  // Run the path state machine for up to 64 slots popped from queue q.
  auto run_batch = [&](int q) {
    const int slot = q_pop(q, true);
    const bool have = slot >= 0;
    if (CNT) { batches++; batchLanes += (uint32_t)__popcll(__ballot(have)); }
    PathState ps; Trav res;
    ps.mode = M_DONE;
    if (have) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const SlotCold c = cold[slot];
      const v4 h0 = hot0[slot], h1 = hot1[slot], h2 = hot2[slot];
      ps.mode = c.mode; ps.pixel = c.pixel; ps.item = c.item; ps.depth = c.depth; ps.seed = c.seed;
      ps.thr = mk3(c.thrx, c.thry, c.thrz); ps.rad = mk3(c.radx, c.rady, c.radz); ps.mat = c.mat;
      ps.N = mk3(c.Nx, c.Ny, c.Nz); ps.light = c.light; ps.V = mk3(c.Vx, c.Vy, c.Vz); ps.pendInv = c.pendInv;
      ps.pendW = mk3(c.pwx, c.pwy, c.pwz); ps.accum = mk3(0, 0, 0); ps.cdlin = mk3(c.cdx, c.cdy, c.cdz);
      ps.o = mk3(h0.x, h0.y, h0.z); ps.tmax = h0.w; ps.d = mk3(h1.x, h1.y, h1.z); ps.kind = f2i(h1.w); ps.tmin = sc.epsT;
      // hit record -> the Trav fields on_result() consumes
      res.tbest = h2.x; res.beta = h2.y; res.gamma = h2.z; res.att = mk3(h2.x, h2.y, h2.z);
      const int ref = f2i(h2.w);
      res.bestTri = ref >= 0 ? ref : -1;
      res.bestPrim = (ref == kMiss) ? -1 : (ref >= 0 ? sc.nSpheres + sc.nQuads : ~ref);
      if (ps.mode == M_TRACE) ps.mode = M_RESULT;
    }
    const bool shadeBatch = (q == Q_SHADE);
    for (;;) {
      // a finished sample is stored right away (12-byte write); fetching the next work item and
      // generating its camera ray is left to a Q_GEN batch so that it runs on a full wave
      if (have && ps.mode == M_NEW_SAMPLE) { store_sample(a, ps.item, ps.accum); ps.mode = M_NEW_PIXEL; }
      const bool run = have && ps.mode != M_TRACE && ps.mode != M_DONE && !(shadeBatch && ps.mode == M_NEW_PIXEL);
      if (__ballot(run) == 0ull) break;
      if (run) {
        if (ps.mode == M_RESULT) {
          on_result<CNT>(sc, ps, res, ct);
        } else if (ps.mode == M_LIGHTS) {
          on_lights<CNT>(sc, ps, ct);
        } else {  // M_NEW_PIXEL: next (pixel, sample) work item
          const int k = atomicAdd(a.workCounter, 1);
          int s;
          if (k >= a.nWork) { ps.mode = M_DONE; }
          else if (item_to_pixel(a, k, s, ps.pixel)) { ps.item = k; begin_sample<CNT>(sc, ps, a.seeds[s], ct); }
        }
      }
    }
    if (have) {
      SlotCold c;
      c.mode = ps.mode; c.pixel = ps.pixel; c.item = ps.item; c.depth = ps.depth; c.seed = ps.seed;
      c.thrx = ps.thr.x; c.thry = ps.thr.y; c.thrz = ps.thr.z; c.radx = ps.rad.x; c.rady = ps.rad.y; c.radz = ps.rad.z; c.mat = ps.mat;
      c.Nx = ps.N.x; c.Ny = ps.N.y; c.Nz = ps.N.z; c.light = ps.light; c.Vx = ps.V.x; c.Vy = ps.V.y; c.Vz = ps.V.z; c.pendInv = ps.pendInv;
      c.pwx = ps.pendW.x; c.pwy = ps.pendW.y; c.pwz = ps.pendW.z; c.cdx = ps.cdlin.x;
      c.cdy = ps.cdlin.y; c.cdz = ps.cdlin.z; c.pad1 = 0; c.pad2 = 0;
      cold[slot] = c;
      if (ps.mode == M_TRACE) {
        v4 h0, h1;
        h0.x = ps.o.x; h0.y = ps.o.y; h0.z = ps.o.z; h0.w = ps.tmax;
        h1.x = ps.d.x; h1.y = ps.d.y; h1.z = ps.d.z; h1.w = i2f(ps.kind);
        hot0[slot] = h0; hot1[slot] = h1;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    q_push(Q_TRAV, have && ps.mode == M_TRACE, slot);
    q_push(Q_GEN, have && ps.mode == M_NEW_PIXEL, slot);
    nDone += __popcll(__ballot(have && ps.mode == M_DONE));
  };

  for (;;) {
    // ---- pick a shading / regeneration batch, or refill the traversal lanes ----
    int bq = -1;
    if (qCount[Q_SHADE] >= 64) { bq = Q_SHADE; if (CNT) dbgFull++; }   // a full wave of the same stage is waiting
    else if (qCount[Q_GEN] >= 64) { bq = Q_GEN; if (CNT) dbgFull++; }
    else {
      unsigned long long idleMask = __ballot(job < 0);
      if (idleMask != 0ull && qCount[Q_TRAV] > 0) {      // hand ready rays to idle lanes
        const int slot = q_pop(Q_TRAV, job < 0);
        if (slot >= 0) {
          job = slot;
          const v4 h0 = hot0[slot], h1 = hot1[slot];
          ray.o = mk3(h0.x, h0.y, h0.z); ray.tmax = h0.w; ray.d = mk3(h1.x, h1.y, h1.z); ray.kind = f2i(h1.w);
          trav_begin<CNT>(sc, ray, tv, ct);
        }
        idleMask = __ballot(job < 0);
      }
      const int nIdle = __popcll(idleMask);
      const int waiting = qCount[Q_SHADE] + qCount[Q_GEN];
      if (nIdle == 64) {                                 // nothing to traverse: drain a stage, or finish
        if (waiting == 0) break;                         // every slot is done
        bq = (qCount[Q_SHADE] >= qCount[Q_GEN]) ? Q_SHADE : Q_GEN;
        if (CNT) { dbgIdle++; dbgWaiting += (uint32_t)waiting; }
      } else if (nIdle >= a.starveLanes && qCount[Q_TRAV] == 0 && waiting > 0) {
        bq = (qCount[Q_SHADE] >= qCount[Q_GEN]) ? Q_SHADE : Q_GEN;   // traversal is starving: make rays
        if (CNT) dbgWaiting += (uint32_t)waiting;
      }
    }
    PT_STAMP(tRefill);
    if (bq >= 0) { run_batch(bq); PT_STAMP(tBatch); continue; }

    // ---- traversal steps ----
    for (;;) {
      // while-while: node steps until enough lanes are parked at a leaf (or have finished) ...
      for (;;) {
        const bool atNode = (job >= 0) & (tv.node >= 0) & (tv.node != kTravDone);
        const unsigned long long nm = __ballot(atNode);
        if (nm == 0ull) break;
        const int nLeaf = __popcll(__ballot((job >= 0) & (tv.node < 0)));
        const int nFin = __popcll(__ballot((job >= 0) & (tv.node == kTravDone)));
        if (nLeaf >= a.leafThreshold || nFin >= a.refillLanes) break;
        if (CNT) { waveSteps++; activeLaneSteps += (uint32_t)__popcll(nm); }
        if (atNode) trav_node_step<CNT>(sc, ray, tv, st, ct);
      }
      PT_STAMP(tNode);
      // ... then one leaf pass for the parked lanes
      {
        const bool atLeaf = (job >= 0) & (tv.node < 0);
        const unsigned long long lm = __ballot(atLeaf);
        if (lm != 0ull) {
          if (CNT) { waveSteps++; activeLaneSteps += (uint32_t)__popcll(lm); leafPasses++; leafLanes += (uint32_t)__popcll(lm); }
          if (atLeaf) trav_leaf_step<CNT>(sc, ray, tv, st, ct);
        }
      }
      PT_STAMP(tLeaf);
      const bool active = job >= 0;
      const bool fin = active && tv.node == kTravDone;
      const unsigned long long finMask = __ballot(fin);
      if (finMask != 0ull) {
        bool toShade = false;
        if (fin) {
          v4 h2;
          if (ray.kind == RK_SHADOW) { h2.x = tv.att.x; h2.y = tv.att.y; h2.z = tv.att.z; h2.w = i2f(0); toShade = true; }
          else {
            const int ref = tv.bestPrim < 0 ? kMiss : (tv.bestTri >= 0 ? tv.bestTri : ~tv.bestPrim);
            h2.x = tv.tbest; h2.y = tv.beta; h2.z = tv.gamma; h2.w = i2f(ref);
            toShade = tv.bestPrim >= 0;
          }
          hot2[job] = h2;
        }
        q_push(Q_SHADE, fin && toShade, job);
        q_push(Q_GEN, fin && !toShade, job);
        if (fin) { job = -1; tv.started = 0; }
      }
      const int idle = 64 - __popcll(__ballot(job >= 0));
      PT_STAMP(tFin);
      if (qCount[Q_SHADE] >= 64 || qCount[Q_GEN] >= 64) break;
      if (idle >= a.refillLanes && qCount[Q_TRAV] > 0) break;
      if (idle >= a.starveLanes && qCount[Q_TRAV] == 0 && (qCount[Q_SHADE] + qCount[Q_GEN]) > 0) break;
      if (idle == 64) break;
    }
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
      atomicAdd(&c[11], (unsigned long long)batches);
      atomicAdd(&c[12], (unsigned long long)batchLanes);
      atomicAdd(&c[13], (unsigned long long)dbgFull);
      atomicAdd(&c[14], (unsigned long long)dbgIdle);
      atomicAdd(&c[15], (unsigned long long)dbgWaiting);
      if (nDone != P) atomicAdd(&c[9], 1ull << 60);
      atomicAdd(&c[16], tBatch); atomicAdd(&c[17], tRefill); atomicAdd(&c[18], tNode); atomicAdd(&c[19], tLeaf); atomicAdd(&c[20], tFin);
      atomicAdd(&c[21], __builtin_amdgcn_s_memtime() - tStart);
      atomicAdd(&c[22], (unsigned long long)leafPasses); atomicAdd(&c[23], (unsigned long long)leafLanes);
    }
  }
  (void)nDone;
}

}  // namespace

int poolkernel_lds_stack_entries() { return kStackN; }
size_t poolkernel_cold_bytes(int nBlocks, int poolSlots) { return (size_t)nBlocks * kWaves * poolSlots * sizeof(SlotCold); }

hipError_t launch_poolkernel(hipStream_t stream, const LaunchArgs& a, int nBlocks, int poolSlots, bool counted) {
  dim3 grid(nBlocks), block(kBlockThreads);
  if (poolSlots == 128) {
    if (counted) pt_poolkernel<true, 128><<<grid, block, 0, stream>>>(a);
    else         pt_poolkernel<false, 128><<<grid, block, 0, stream>>>(a);
  } else if (poolSlots == 192) {
    if (counted) pt_poolkernel<true, 192><<<grid, block, 0, stream>>>(a);
    else         pt_poolkernel<false, 192><<<grid, block, 0, stream>>>(a);
  } else if (poolSlots == 256) {
    if (counted) pt_poolkernel<true, 256><<<grid, block, 0, stream>>>(a);
    else         pt_poolkernel<false, 256><<<grid, block, 0, stream>>>(a);
  } else {
    return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace pt
