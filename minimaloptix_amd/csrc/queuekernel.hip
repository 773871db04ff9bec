// queuekernel.hip -- megakernel variants 2 and 3: slot-resident traversal state + stage queues.
//
// Measured on variant 1 (coffee, 32 spp): a wave spends 42 % of its time in BVH node steps that
// run with 34 of 64 lanes, 22 % in leaf passes that run with 18 lanes, 11 % setting rays up for a
// quarter of the lanes.  The lanes are idle because a lane *owns* its ray: when the ray is parked
// at a leaf or finished, the lane waits.  Here nothing is owned by a lane.  Path slots hold the
// complete state in LDS (ray, 1/d, traversal registers, traversal stack) and in an HBM record
// (path payload); the lanes are workers that pick slots off four queues:
//
//   Q_NODE  slots standing at an internal BVH node   -> node loop (o, 1/d, tbest, node, sp only)
//   Q_LEAF  slots standing at a leaf                 -> leaf pass (triangle tests, any-hit)
//   Q_SHADE slots whose ray finished (hit / shadow)  -> closest-hit shading, NEE, next ray set-up
//   Q_GEN   slots needing a new (pixel,sample) item  -> miss, camera ray
//
// A lane in the node loop that reaches a leaf or finishes its ray swaps: it stores (node, sp),
// queues the slot and pops the next slot from Q_NODE.  Leaf passes, shading and regeneration run
// as 64-wide batches when a queue fills up (or when the node loop starves).  This is the
// wave-level ray compaction + sorting by stage of the north star; the per-path arithmetic is the
// same pt_path.h code as variants 0/1, so the images are bit-identical.
//
//   (SHARED = false, every wave with its own 128 slots and queues, was variant 2 of rounds 1-2; it is no longer instantiated)
//   variant 3 (SHARED = true):  the 4 waves of a workgroup share 512 slots and one set of queues
//       (a spin-lock in LDS guards one short queue transaction per pass), so that full batches
//       of every stage are available almost all the time.
#include <hip/hip_runtime.h>

#include "megakernel.h"
#include "pt_path.h"

namespace pt {

namespace {

constexpr int kBlockThreads = 256;
constexpr int kWaves = kBlockThreads / 64;
#ifndef PT_KP
#define PT_KP 128
#endif
#ifndef PT_STACKN
#define PT_STACKN 11
#endif
#ifndef PT_WAVES_PER_SIMD
#define PT_WAVES_PER_SIMD 3
#endif
constexpr int kP = PT_KP;               // slots per wave
constexpr int kStackN = PT_STACKN;      // LDS stack entries per slot; deeper levels spill to HBM
constexpr int kWavesPerSimd = PT_WAVES_PER_SIMD;   // occupancy target: 3 workgroups per CU (VGPR <= 168, LDS <= 53 KB)
constexpr int ring_capacity(int n) { int c = 1; while (c < n) c <<= 1; return c; }

// pool-wide queues first (they index PoolLds::queue); Q_NODE / Q_LEAF are per-wave rings (WavePriv)
enum { Q_SHADE = 0, Q_GEN = 1, kNumQ = 2, Q_NODE = 2, Q_LEAF = 3, DEST_DONE = 4, DEST_NONE = -1 };

// Path-slot record in HBM, private to the pool: eight 16-byte rows, grouped by WHO needs them and WHEN they change, so
// that a visit moves only the rows it uses (round 1 moved all 144 bytes in and out on every shading visit and 48-80
// bytes on every leaf visit: 474 B of scheduler state per ray, most of the kernel's fabric traffic):
//   leaf pass                      reads hit once a hit exists, writes hit only when the nearest hit changed
//   shading, radiance ray back     reads ctl thr rad (+ hit)
//   shading, shadow ray back       reads ctl thr rad nrm view pend (+ hit = attenuation, once a glass surface was crossed)
//   new shadow ray                 writes ctl rad pend, and nrm view only for the first light of a Disney hit
//   new radiance ray               writes ctl, and thr / rad only if they changed
// What the rows do not hold lives in LDS (origin, direction, tbest, node, stack; ray type and "hit row valid" in the
// slot's flag word), or is implied: the pixel follows from the work item, tmin is the scene's epsilon, a radiance ray's
// tmax is RT_DEFAULT_MAX and a shadow ray's tmax is the tbest the node loop carries (a shadow ray never shortens it).
struct alignas(16) i4 { int x, y, z, w; };
struct alignas(16) SlotCold {
  i4 ctl;     // item, depth, seed, mode | light << 3
  v4 thr;     // throughput, cdlin.y
  v4 rad;     // radiance so far, cdlin.z
  v4 spare;
  v4 hit;     // radiance ray: bestTri, bestPrim (int bits), beta, gamma | shadow ray: attenuation
  v4 nrm;     // Disney hit context while its lights are looped: N, mat (int bits)
  v4 view;    // V, cdlin.x
  v4 pend;    // weight of the shadow ray in flight: pendW, pendInv
};
static_assert(sizeof(SlotCold) == 128, "SlotCold layout");

// Slot records stream through the cache hierarchy once per visit; PT_SLOT_NT marks their loads/stores
// non-temporal so that they do not push BVH nodes out of the 4 MB L2 of the XCD.
#ifndef PT_SLOT_NT
#define PT_SLOT_NT 0
#endif
typedef float f4v __attribute__((ext_vector_type(4)));
template <class T> __device__ __forceinline__ T slot_load(const T* p) {
  static_assert(sizeof(T) % 16 == 0, "16-byte granules");
  if constexpr (PT_SLOT_NT) {
    T out;
    const f4v* src = reinterpret_cast<const f4v*>(p); f4v* dst = reinterpret_cast<f4v*>(&out);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 16; i++) dst[i] = __builtin_nontemporal_load(src + i);
    return out;
  } else {
    return *p;
  }
}
template <class T> __device__ __forceinline__ void slot_store(T* p, const T& v) {
  static_assert(sizeof(T) % 16 == 0, "16-byte granules");
  if constexpr (PT_SLOT_NT) {
    const f4v* src = reinterpret_cast<const f4v*>(&v); f4v* dst = reinterpret_cast<f4v*>(p);
#pragma unroll
    for (unsigned i = 0; i < sizeof(T) / 16; i++) __builtin_nontemporal_store(src[i], dst + i);
  } else {
    *p = v;
  }
}

// LDS image of NS slots
template <int NS>
struct PoolLds {
  // only what the node loop touches lives in LDS: 32 B + the stack per slot
  v4 nodeA[NS];           // o.xyz, tbest
  v4 nodeB[NS];           // d.xyz, node (int bits); a lane that picks the slot up for the node loop derives 1/d from it
  int stack[NS][kStackN + 1];   // [0] = sp | kShadeFlag, [1..] = entries
  unsigned short queue[kNumQ][ring_capacity(NS)];
  int qHead[kNumQ], qCount[kNumQ];   // SHARED only
  int done, lock;                    // SHARED only
};

template <int NS>
struct WavePriv {
  unsigned short qnode[ring_capacity(NS)];   // node-ready slots owned by this wave (ring)
  // One loop iteration pushes at most 128 slots (results of the last pass + lanes leaving the node loop).
  unsigned short qleaf[256];       // slots standing at a leaf, owned by this wave (ring; a pass runs at 64: < 64 + 128)
  unsigned short outbox[2][160];   // slots on their way to the pool's Q_SHADE / Q_GEN (flushed at 32: < 32 + 128)
};

typedef __attribute__((address_space(3))) int lds_int;

// The LDS part is addressed through an address_space(3) pointer so that push/pop compile to
// ds_write_b32/ds_read_b32 (a generic pointer makes the compiler merge the LDS and the HBM
// overflow path into one flat_load).
// flag bits next to the stack pointer in stack[slot][0]
constexpr int kShadeFlag = 1 << 30;   // a finished ray goes to Q_SHADE (hit or shadow ray), not Q_GEN
constexpr int kShadowRay = 1 << 29;   // the ray in flight is a shadow ray (MinimalOptiX.h:48 RAY_TYPE_SHADOW)
constexpr int kHitValid = 1 << 28;    // SlotCold::hit holds this ray's nearest hit / attenuation (else: none yet / (1,1,1))
constexpr int kSlotFlags = kShadeFlag | kShadowRay | kHitValid;

struct SlotStack {
  lds_int* lds;           // &stack[slot][1]
  int* ovf;               // this slot's overflow area in HBM (or nullptr)
  __device__ __forceinline__ void store(int sp, int v) {
    if (__builtin_expect(sp < kStackN, 1)) lds[sp] = v; else ovf[sp - kStackN] = v;
  }
  __device__ __forceinline__ int load(int sp) const {
    int v;
    if (__builtin_expect(sp < kStackN, 1)) v = lds[sp]; else v = ovf[sp - kStackN];
    return v;
  }
  __device__ __forceinline__ bool roomy(int sp) const { return sp + 3 <= kStackN; }     // three pushes stay in LDS
  __device__ __forceinline__ void store_fast(int sp, int v) { lds[sp] = v; }
  static constexpr bool kFlat = false;      // pt_path.h node_step_nearfar: this stack takes the branched tail
  __device__ __forceinline__ bool fits_fast(int, int) const { return false; }
  __device__ __forceinline__ int peek_fast(int) const { return 0; }
};

__device__ __forceinline__ float node_inv(float d) {      // slab_inv (pt_path.h) with the hardware reciprocal
  return __builtin_amdgcn_rcpf(__builtin_fabsf(d) < 1e-30f ? __builtin_copysignf(1e-30f, d) : d);
}
__device__ __forceinline__ int lane_rank(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// destination queue of a slot after a traversal step / a ray set-up
__device__ __forceinline__ int route(int node, int kind, int bestPrim) {
  if (node == kTravDone) return (kind == RK_SHADOW || bestPrim >= 0) ? Q_SHADE : Q_GEN;
  return node >= 0 ? Q_NODE : Q_LEAF;
}

// NEAR: the scene has a Disney GLASS material, shadow rays keep their nearest any-hit candidate (pt_path.h, rule D5).  A template
// parameter so that scenes without one -- the benchmark scene -- run code in which that logic does not exist.
// The launch arguments again, through a pointer the compiler cannot see through (packetkernel.hip fresh_args has the whole story): a pass that needs the
// scene takes its own copy -- s_load where used, dead when the pass ends -- instead of ~100 loop-invariant words living in scalar registers for the whole launch
// and the scheduler's own scalar state in lanes of spill registers (round 6: this kernel shipped with 118-120 spilled scalar registers).
__device__ __forceinline__ const LaunchArgs& fresh_args() {
  typedef __attribute__((address_space(4))) const LaunchArgs CA;
  unsigned long long p = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return *(const LaunchArgs*)(CA*)p;
}

template <bool CNT, bool SHARED, bool FAST = false, bool NEAR = false>
__global__ void __launch_bounds__(kBlockThreads, kWavesPerSimd) pt_queuekernel(const LaunchArgs a) {
  constexpr int NS = SHARED ? kP * kWaves : kP;          // slots per pool
  constexpr int RC = ring_capacity(NS);                  // ring capacity (power of two >= NS)
  __shared__ PoolLds<NS> sPool[SHARED ? 1 : kWaves];
  __shared__ WavePriv<NS> sPriv[kWaves];

  SceneView scv = a.scene;
  scv.shadowNearest = NEAR ? 1 : 0;                 // compile-time constant from here on
  const SceneView& sc = scv;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  PoolLds<NS>& W = sPool[SHARED ? 0 : wave];
  const int gpool = SHARED ? blockIdx.x : blockIdx.x * kWaves + wave;
  SlotCold* cold = reinterpret_cast<SlotCold*>(a.poolCold) + (size_t)gpool * NS;
  int* ovfBase = a.stackOverflow ? a.stackOverflow + (size_t)gpool * NS * a.ovfDepth : nullptr;

  // sub-phase clocks of the counting build
  unsigned long long tLocal = 0, tLock = 0, tTxn = 0, tIdle = 0, tBLoad = 0, tBRun = 0, tBStore = 0, nTxn = 0, nIter = 0;
  unsigned long long tSub = 0, nIterResult = 0, nIterLights = 0, nIterGen = 0, nodeRuns = 0, ringBacklog = 0, leafBacklog = 0;
#define PT_SUB0() do { if (CNT) tSub = __builtin_amdgcn_s_memtime(); } while (0)
#define PT_SUB(acc) do { if (CNT) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc += now_ - tSub; tSub = now_; } } while (0)
  // queue bookkeeping: registers (wave-uniform); with SHARED they mirror LDS inside a transaction
  int qHead[kNumQ] = { 0, 0 }, qCount[kNumQ] = { 0, 0 };
  int nDone = 0;

  [[maybe_unused]] auto q_push = [&](int q, bool pred, int slot) {
    const unsigned long long m = __ballot(pred);
    if (m == 0ull) return;
    if (pred) W.queue[q][(qHead[q] + qCount[q] + lane_rank(m)) & (RC - 1)] = (unsigned short)slot;
    qCount[q] += __popcll(m);
  };
  auto q_pop = [&](int q, bool want) -> int {
    const unsigned long long m = __ballot(want);
    const int n = min((int)__popcll(m), qCount[q]);
    int slot = -1;
    if (want) { const int r = lane_rank(m); if (r < n) slot = W.queue[q][(qHead[q] + r) & (RC - 1)]; }
    qHead[q] = (qHead[q] + n) & (RC - 1);
    qCount[q] -= n;
    return slot;
  };
  auto txn_begin = [&]() {
    if constexpr (SHARED) {
      if (lane == 0) { while (atomicCAS(&W.lock, 0, 1) != 0) __builtin_amdgcn_s_sleep(1); }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      if (CNT) { (void)__builtin_amdgcn_readfirstlane(W.lock); PT_SUB(tLock); nTxn++; }
      for (int q = 0; q < kNumQ; q++) {
        qHead[q] = __builtin_amdgcn_readfirstlane(W.qHead[q]);
        qCount[q] = __builtin_amdgcn_readfirstlane(W.qCount[q]);
      }
      nDone = __builtin_amdgcn_readfirstlane(W.done);
    }
  };
  auto txn_end = [&]() {
    if constexpr (SHARED) {
      if (lane == 0) {
        for (int q = 0; q < kNumQ; q++) { W.qHead[q] = qHead[q]; W.qCount[q] = qCount[q]; }
        W.done = nDone;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) __hip_atomic_store(&W.lock, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  };
  auto make_stack = [&](int slot) {
    SlotStack st;
    st.lds = (lds_int*)&W.stack[slot][1];
    st.ovf = ovfBase ? ovfBase + (size_t)slot * a.ovfDepth : nullptr;
    return st;
  };

  // ---- start-up: every slot in use needs a work item ----
  // Paths in flight = slots in use; by Little's law a ray spends (slots in use) / (rays per second) in the scheduler,
  // about 70 us with all 512 slots of every pool, so a path that bounces to the depth cap (about 1000 dependent rays)
  // takes 70-90 ms however short the launch is.  A short launch (one rank's share of a multi-GPU frame) therefore
  // uses fewer slots: a little less throughput, a much shorter critical path (LaunchArgs::slotsInUse, moptix_api.hip).
  const int nUse = (a.slotsInUse > 0 && a.slotsInUse < NS) ? (SHARED ? a.slotsInUse : max(64, a.slotsInUse / kWaves)) : NS;
  {
    const int first = SHARED ? threadIdx.x : lane, step = SHARED ? kBlockThreads : 64;
    for (int s = first; s < nUse; s += step) {
      i4 ctl; ctl.x = -1; ctl.y = 0; ctl.z = 0; ctl.w = M_NEW_PIXEL;     // item -1: whatever sample comes first is "new" (rows get written)
      cold[s].ctl = ctl;
      W.stack[s][0] = 0;
      W.queue[Q_GEN][s] = (unsigned short)s;
    }
    if constexpr (SHARED) {
      if (threadIdx.x == 0) {
        for (int q = 0; q < kNumQ; q++) { W.qHead[q] = 0; W.qCount[q] = 0; }
        W.qCount[Q_GEN] = nUse; W.done = NS - nUse; W.lock = 0;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __syncthreads();
    } else {
      qCount[Q_GEN] = nUse; nDone = NS - nUse;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
  }

  Counters ct = {};
  uint32_t nodeSteps = 0, nodeLanes = 0, leafPasses = 0, leafLanes = 0, batches = 0, batchLanes = 0, idleSpins = 0;
  unsigned long long tBatch = 0, tSwap = 0, tNode = 0, tLeaf = 0, tStamp = 0;
  const unsigned long long tStart = CNT ? __builtin_amdgcn_s_memtime() : 0ull;
  [[maybe_unused]] const unsigned long long rtStart = CNT ? __builtin_amdgcn_s_memrealtime() : 0ull;
#define PT_STAMP(acc) do { if (CNT) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc += now_ - tStamp; tStamp = now_; } } while (0)
  tStamp = tStart;

  // ---- node-loop worker context: the only per-lane state that survives between passes ----
  int ns = -1;                 // slot this lane is walking, -1 = none
  int nsFlag = 0;              // flag bits (kSlotFlags) of that slot
  PathState nray;              // o, tmin used
  nray.o = mk3(0, 0, 0); nray.tmin = sc.epsT; nray.d = mk3(0, 0, 1); nray.tmax = 0; nray.kind = RK_RADIANCE; nray.mode = M_TRACE;
  Trav ntv;                    // inv, tbest, node, sp used
  ntv.node = kTravDone; ntv.sp = 0; ntv.tbest = 0; ntv.inv = mk3(0, 0, 0); ntv.noi = mk3(0, 0, 0); ntv.bestPrim = -1; ntv.bestTri = -1;
  ntv.beta = 0; ntv.gamma = 0; ntv.att = mk3(1, 1, 1); ntv.started = 1;

  // result of the last pass, queued inside the next transaction
  int pendSlot = -1, pendDest = DEST_NONE;

  // ---- leaf pass: the slots popped from Q_LEAF stand at a leaf ----
  auto leaf_pass = [&](int slot) {
    const LaunchArgs& a = fresh_args();                 // this pass's own view of the arguments
    SceneView scl = a.scene; scl.shadowNearest = NEAR ? 1 : 0;
    const SceneView& sc = scl;
    const bool have = slot >= 0;
    if (CNT) { leafPasses++; leafLanes += (uint32_t)__popcll(__ballot(have)); }
    pendSlot = slot; pendDest = DEST_NONE;
    // slot records written by earlier passes of this wave (other lanes) or published by other waves: the stores
    // of the previous pass were left in flight, so they are ordered here, where their latency has already elapsed
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (have) {
      PathState ps; Trav tv;
      ps.tmin = sc.epsT; ps.mode = M_TRACE;
      const v4 na = W.nodeA[slot], nb = W.nodeB[slot];
      const int fl = W.stack[slot][0];
      const bool shadow = (fl & kShadowRay) != 0, hitValid = (fl & kHitValid) != 0;
      // the leaf is known from LDS: its triangles are requested together with the slot's rows (one round trip)
      LeafChunk ch;
      leaf_fetch4(sc, f2i(nb.w), 0, ch);
      const SlotCold* cs = cold + slot;
      v4 wh = mk4(0.f, 0.f, 0.f, 0.f);
      if (hitValid) wh = slot_load(&cs->hit);
      ps.o = mk3(na.x, na.y, na.z); ps.d = mk3(nb.x, nb.y, nb.z);
      ps.kind = shadow ? RK_SHADOW : RK_RADIANCE;
      ps.tmax = shadow ? na.w : kRtDefaultMax;
      tv.inv = mk3(0.f, 0.f, 0.f); tv.tbest = na.w;      // the leaf step does not use 1/d
      tv.node = f2i(nb.w); tv.sp = fl & ~kSlotFlags;
      tv.bestTri = -1; tv.bestPrim = -1; tv.beta = 0.f; tv.gamma = 0.f; tv.att = mk3(1.f, 1.f, 1.f);
      if (hitValid) {
        if (shadow) { tv.att = mk3(wh.x, wh.y, wh.z); if (sc.shadowNearest) tv.bestPrim = f2i(wh.w); }     // verdict + id of the nearest any-hit surface so far
        else { tv.bestTri = f2i(wh.x); tv.bestPrim = f2i(wh.y); tv.beta = wh.z; tv.gamma = wh.w; }
      }
      const int oldTri = tv.bestTri, oldPrim = tv.bestPrim;
      const v3 oldAtt = tv.att;
      SlotStack st = make_stack(slot);
      trav_leaf_step_fetched<CNT>(sc, ps, tv, st, ct, ch);
      // most leaf visits find nothing nearer: the hit row is only written when it changed (beta / gamma change with bestTri)
      const bool changed = shadow ? ((tv.att.x != oldAtt.x || tv.att.y != oldAtt.y || tv.att.z != oldAtt.z) || (sc.shadowNearest && tv.bestPrim != oldPrim))
                                  : (tv.bestTri != oldTri || tv.bestPrim != oldPrim);
      if (changed) {
        v4 o2;
        if (shadow) o2 = mk4(tv.att.x, tv.att.y, tv.att.z, i2f(tv.bestPrim));
        else o2 = mk4(i2f(tv.bestTri), i2f(tv.bestPrim), tv.beta, tv.gamma);
        slot_store(&cold[slot].hit, o2);
      }
      W.nodeA[slot].w = tv.tbest;
      W.nodeB[slot].w = i2f(tv.node);
      W.stack[slot][0] = tv.sp | (fl & kShadowRay) | ((hitValid || changed) ? kHitValid : 0) |
                         ((shadow || tv.bestPrim >= 0) ? kShadeFlag : 0);
      pendDest = route(tv.node, ps.kind, tv.bestPrim);
    }
    // Slot records in HBM are re-read by other lanes / waves.  With a shared pool a slot only reaches another wave
    // through a queue transaction, whose release fence (txn_end) covers these stores: no wait here, the store
    // latency overlaps with the bookkeeping that follows.
    if constexpr (!SHARED) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  };

  // ---- shading / regeneration batch: run the path state machine for the popped slots ----
  auto run_batch = [&](int slot, bool shadeBatch) {
    const LaunchArgs& a = fresh_args();                 // this pass's own view of the arguments
    SceneView scl = a.scene; scl.shadowNearest = NEAR ? 1 : 0;
    const SceneView& sc = scl;
    const bool have = slot >= 0;
    if (CNT) { batches++; batchLanes += (uint32_t)__popcll(__ballot(have)); }
    PT_SUB0();
    pendSlot = slot; pendDest = DEST_NONE;
    PathState ps; Trav res;
    ps.mode = M_DONE; ps.kind = RK_RADIANCE;
    res.node = kTravDone; res.bestPrim = -1;
    // slot records written by earlier passes of this wave (other lanes) or published by other waves: the stores
    // of the previous pass were left in flight, so they are ordered here, where their latency has already elapsed
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    bool shadowIn = false;                       // this visit started with a shadow ray coming back
    // which rows this visit has to write back.  Throughput changes with a bounce (depth) or a new sample (item) only;
    // radiance and the texture colour are compared around on_result (short live ranges instead of two rows of registers
    // held across the whole batch)
    int depthIn = 0, itemIn = 0;
    bool radDirty = false, texDirty = false;
    if (have) {
      const SlotCold* cs = cold + slot;
      const int fl = W.stack[slot][0];
      const bool shadow = (fl & kShadowRay) != 0, hitValid = (fl & kHitValid) != 0;
      const i4 ctl = slot_load(&cs->ctl);
      const v4 thrIn = slot_load(&cs->thr), radIn = slot_load(&cs->rad);
      depthIn = ctl.y; itemIn = ctl.x;
      v4 wh = mk4(0.f, 0.f, 0.f, 0.f);
      v4 wn = mk4(0.f, 0.f, 1.f, 0.f), wv = mk4(0.f, 0.f, 1.f, 0.f), wp = mk4(0.f, 0.f, 0.f, 0.f);
      if (hitValid) wh = slot_load(&cs->hit);
      if (shadow) { wn = slot_load(&cs->nrm); wv = slot_load(&cs->view); wp = slot_load(&cs->pend); }
      const v4 na = W.nodeA[slot], wd = W.nodeB[slot];
      ps.mode = ctl.w & 7; ps.light = ctl.w >> 3; ps.item = ctl.x; ps.depth = ctl.y; ps.seed = (uint32_t)ctl.z; ps.pixel = 0;
      ps.thr = mk3(thrIn.x, thrIn.y, thrIn.z); ps.rad = mk3(radIn.x, radIn.y, radIn.z);
      ps.N = mk3(wn.x, wn.y, wn.z); ps.mat = f2i(wn.w); ps.V = mk3(wv.x, wv.y, wv.z);
      ps.pendW = mk3(wp.x, wp.y, wp.z); ps.pendInv = wp.w; ps.accum = mk3(0, 0, 0);
      ps.cdlin = mk3(wv.w, thrIn.w, radIn.w);
      ps.o = mk3(na.x, na.y, na.z); ps.d = mk3(wd.x, wd.y, wd.z); ps.tmin = sc.epsT; ps.tmax = kRtDefaultMax;
      ps.kind = shadow ? RK_SHADOW : RK_RADIANCE;
      res.tbest = na.w; res.bestTri = -1; res.bestPrim = -1; res.beta = 0.f; res.gamma = 0.f; res.att = mk3(1.f, 1.f, 1.f);
      if (hitValid) {
        if (shadow) res.att = mk3(wh.x, wh.y, wh.z);
        else { res.bestTri = f2i(wh.x); res.bestPrim = f2i(wh.y); res.beta = wh.z; res.gamma = wh.w; }
      }
      if (ps.mode == M_TRACE) { ps.mode = M_RESULT; shadowIn = shadow; }
    }
    if (CNT) { __builtin_amdgcn_s_waitcnt(0); PT_SUB(tBLoad); }
    for (;;) {
      if (have && ps.mode == M_NEW_SAMPLE) {
        if (CNT) {      // finish-time histogram (1 ms buckets): how many samples end when, and how deep they were
          const unsigned long long b = min(255ull, (__builtin_amdgcn_s_memrealtime() - rtStart) / 100000ull);
          atomicAdd(a.counters + 40 + b, 1ull); atomicMax(a.counters + 296 + b, (unsigned long long)ps.depth); atomicAdd(a.counters + 552 + b, (unsigned long long)ps.depth);
        }
        store_sample(a, ps.item, ps.accum);
        if (a.tileCost != nullptr && ps.depth >= kDeepPath) atomicMax(a.tileCost + ((ps.item % a.nItems) >> a.unitShift), (unsigned int)ps.depth);
        ps.mode = M_NEW_PIXEL;
      }
      // lanes that arrive with a finished ray go first; the lanes that only need a new work item wait for them, so
      // that begin_sample + the ray set-up run once, for all of them together
      const bool resultFirst = __ballot(have && ps.mode == M_RESULT) != 0ull;
      const bool run = have && ps.mode != M_TRACE && ps.mode != M_DONE &&
                       !((shadeBatch || resultFirst) && ps.mode == M_NEW_PIXEL);
      if (__ballot(run) == 0ull) break;
      if (CNT) {   // which state-machine stages this iteration executes (wave level)
        if (__ballot(run && ps.mode == M_RESULT)) nIterResult++;
        if (__ballot(run && ps.mode == M_LIGHTS)) nIterLights++;
        if (__ballot(run && ps.mode == M_NEW_PIXEL)) nIterGen++;
      }
      if (run) {
        if (ps.mode == M_RESULT) {
          const v3 r0 = ps.rad, c0 = ps.cdlin;
          on_result<CNT>(sc, ps, res, ct);
          radDirty = radDirty || f2i(r0.x) != f2i(ps.rad.x) || f2i(r0.y) != f2i(ps.rad.y) || f2i(r0.z) != f2i(ps.rad.z);
          texDirty = texDirty || f2i(c0.x) != f2i(ps.cdlin.x) || f2i(c0.y) != f2i(ps.cdlin.y) || f2i(c0.z) != f2i(ps.cdlin.z);
        } else if (ps.mode == M_LIGHTS) {
          on_lights<CNT, FAST>(sc, ps, ct);
        } else {  // M_NEW_PIXEL: next (pixel, sample) work item
          int k = atomicAdd(a.workCounter, 1);
          k = (k >= a.nWork) ? -1 : handout_to_item(a, k);
          if (CNT && k < 0) atomicMin(a.counters + 37, (unsigned long long)__builtin_amdgcn_s_memrealtime());   // first time the items ran out
          int s;
          if (k < 0) { ps.mode = M_DONE; }
          else if (item_to_pixel(a, k, s, ps.pixel)) { ps.item = k; begin_sample<CNT>(sc, ps, a.seeds[s], ct); }
        }
      }
    }
    if (CNT) { __builtin_amdgcn_s_waitcnt(0); PT_SUB(tBRun); }
    if (have) {
      SlotCold* cw = cold + slot;
      i4 ctl; ctl.x = ps.item; ctl.y = ps.depth; ctl.z = (int)ps.seed; ctl.w = ps.mode | (ps.light << 3);
      slot_store(&cw->ctl, ctl);
      // throughput and radiance rows: only when their bits changed (a shadow ray leaves thr alone, a glass bounce leaves rad alone)
      const bool newPath = ps.item != itemIn || ps.depth != depthIn;      // bounce or new sample
      if (newPath || texDirty) slot_store(&cw->thr, mk4(ps.thr.x, ps.thr.y, ps.thr.z, ps.cdlin.y));
      if (ps.item != itemIn || radDirty || texDirty) slot_store(&cw->rad, mk4(ps.rad.x, ps.rad.y, ps.rad.z, ps.cdlin.z));
      if (ps.mode == M_TRACE) {
        // new ray: analytic primitives + traversal set-up happen here, on the full batch
        Trav tv;
        trav_begin<CNT>(sc, ps, tv, ct);
        v4 na, nb;
        na.x = ps.o.x; na.y = ps.o.y; na.z = ps.o.z; na.w = tv.tbest;
        nb.x = ps.d.x; nb.y = ps.d.y; nb.z = ps.d.z; nb.w = i2f(tv.node);
        W.nodeA[slot] = na; W.nodeB[slot] = nb;
        const bool shadow = ps.kind == RK_SHADOW;
        const bool hitNow = shadow ? ((tv.att.x != 1.f || tv.att.y != 1.f || tv.att.z != 1.f) || (sc.shadowNearest && tv.bestPrim >= 0)) : (tv.bestPrim >= 0);
        W.stack[slot][0] = ((shadow || tv.bestPrim >= 0) ? kShadeFlag : 0) | (shadow ? kShadowRay : 0) | (hitNow ? kHitValid : 0);
        if (hitNow) slot_store(&cw->hit, shadow ? mk4(tv.att.x, tv.att.y, tv.att.z, i2f(tv.bestPrim)) : mk4(i2f(tv.bestTri), i2f(tv.bestPrim), tv.beta, tv.gamma));
        if (shadow) {
          slot_store(&cw->pend, mk4(ps.pendW.x, ps.pendW.y, ps.pendW.z, ps.pendInv));
          if (!shadowIn) {      // first light of this Disney hit: N, V, material (and the texture colour) are new
            slot_store(&cw->nrm, mk4(ps.N.x, ps.N.y, ps.N.z, i2f(ps.mat)));
            slot_store(&cw->view, mk4(ps.V.x, ps.V.y, ps.V.z, ps.cdlin.x));
          }
        }
        pendDest = route(tv.node, ps.kind, tv.bestPrim);
      } else {
        W.stack[slot][0] = 0;
        pendDest = (ps.mode == M_NEW_PIXEL) ? Q_GEN : DEST_DONE;
      }
    }
    if constexpr (!SHARED) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    PT_SUB(tBStore);
  };

  // per-wave private structures: the node-ready ring (slots produced by this wave's own passes
  // stay with the wave) and the out-boxes that collect slots for the pool-wide queues between
  // two transactions
  unsigned short* myNodeQ = sPriv[wave].qnode;
  int nqHead = 0, nqCount = 0;
  unsigned short* myLeafQ = sPriv[wave].qleaf;
  int lqHead = 0, lqCount = 0;             // leaf-ready ring: leaf passes need no queue transaction
  int obCount[2] = { 0, 0 };               // out-boxes for Q_SHADE, Q_GEN
  int localDone = 0;
  auto local_push = [&](int dest, int slot) {   // wave-collective; dest per lane
    {
      const unsigned long long m = __ballot(dest == Q_NODE);
      if (m != 0ull) {
        if (dest == Q_NODE) myNodeQ[(nqHead + nqCount + lane_rank(m)) & (RC - 1)] = (unsigned short)slot;
        nqCount += __popcll(m);
      }
    }
    {
      const unsigned long long m = __ballot(dest == Q_LEAF);
      if (m != 0ull) {
        if (dest == Q_LEAF) myLeafQ[(lqHead + lqCount + lane_rank(m)) & 255] = (unsigned short)slot;
        lqCount += __popcll(m);
      }
    }
    for (int d = 0; d < 2; d++) {
      const unsigned long long m = __ballot(dest == Q_SHADE + d);
      if (m != 0ull) {
        if (dest == Q_SHADE + d) sPriv[wave].outbox[d][obCount[d] + lane_rank(m)] = (unsigned short)slot;
        obCount[d] += __popcll(m);
      }
    }
    localDone += __popcll(__ballot(dest == DEST_DONE));
  };

  unsigned int guard = 0;
  const unsigned long long wdStart = __builtin_amdgcn_s_memrealtime();
  for (;;) {
    // bounded in wall-clock time (never hang the GPU): checked every 4096 iterations
    if ((++guard & 4095u) == 0u && __builtin_amdgcn_s_memrealtime() - wdStart > a.watchdogTicks) { if (lane == 0) atomicOr(a.workCounter + 1, 1); break; }
    PT_SUB0(); if (CNT) nIter++;
    // ---- local bookkeeping (no lock): results of the last pass, swap, refill ----
    if (__ballot(pendDest != DEST_NONE) != 0ull) { local_push(pendDest, pendSlot); pendDest = DEST_NONE; }
    {
      const bool leave = ns >= 0 && !(ntv.node >= 0 && ntv.node != kTravDone);
      if (__ballot(leave) != 0ull) {
        int dest = DEST_NONE;
        if (leave) {
          W.nodeB[ns].w = i2f(ntv.node); W.stack[ns][0] = ntv.sp | nsFlag;
          dest = (ntv.node == kTravDone) ? ((nsFlag & kShadeFlag) ? Q_SHADE : Q_GEN) : Q_LEAF;   // a lane leaves at a leaf or finished
        }
        local_push(dest, ns);
        if (leave) ns = -1;
      }
      if (nqCount > 0 && __ballot(ns < 0) != 0ull) {
        const unsigned long long m = __ballot(ns < 0);
        const int n = min((int)__popcll(m), nqCount);
        if (ns < 0) {
          const int r = lane_rank(m);
          if (r < n) {
            ns = myNodeQ[(nqHead + r) & (RC - 1)];
            const v4 na = W.nodeA[ns], nb = W.nodeB[ns];
            const int spw = W.stack[ns][0];
            nray.o = mk3(na.x, na.y, na.z); ntv.tbest = na.w;
            // 1/d by v_rcp_f32 (1 ulp): the slab test is conservative by far more than that (boxes are padded by 1e-5 of
            // the scene, pt_lbvh.h pad_lo/pad_hi), and which boxes are entered never changes the result (rule D5)
            ntv.inv = mk3(node_inv(nb.x), node_inv(nb.y), node_inv(nb.z)); ntv.node = f2i(nb.w);
            ntv.noi = neg_o_inv(nray.o, ntv.inv);
            ntv.sp = spw & ~kSlotFlags; nsFlag = spw & kSlotFlags;
          }
        }
        nqHead = (nqHead + n) & (RC - 1); nqCount -= n;
      }
    }
    const int nActive = __popcll(__ballot(ns >= 0));
    const bool starving = nActive <= 64 - a.starveLanes;
    int pass = -1;      // -1 node loop, 0 leaf, 1 shade, 2 gen, 3 idle, 4 exit
    int mySlot = -1;
    auto leaf_pop = [&]() -> int {              // up to 64 slots from this wave's own leaf ring
      const int n = min(64, lqCount);
      const int slot = lane < n ? (int)myLeafQ[(lqHead + lane) & 255] : -1;
      lqHead = (lqHead + n) & 255; lqCount -= n;
      return slot;
    };
    const bool flush = obCount[0] >= 32 || obCount[1] >= 32;      // out-boxes are bounded: flushing comes first
    if (!flush && (lqCount >= 64 || (starving && lqCount >= 32))) {
      pass = 0; mySlot = leaf_pop();
      PT_SUB(tLocal);
    } else if (starving || flush) {
      // =========================== queue transaction ===========================
      PT_SUB(tLocal);
      txn_begin();
      for (int d = 0; d < 2; d++) {
        const int q = Q_SHADE + d;
        for (int base = 0; base < obCount[d]; base += 64) {
          const int i = base + lane;
          if (i < obCount[d]) W.queue[q][(qHead[q] + qCount[q] + i) & (RC - 1)] = sPriv[wave].outbox[d][i];
        }
        qCount[q] += obCount[d]; obCount[d] = 0;
      }
      nDone += localDone; localDone = 0;
      if (qCount[Q_SHADE] >= 64) pass = 1;
      else if (qCount[Q_GEN] >= 64) pass = 2;
      else if (lqCount >= 64) pass = 0;
      else if (starving) {
        const int l = lqCount, sh = qCount[Q_SHADE], g = qCount[Q_GEN];
        if (l + sh + g == 0) { if (nActive == 0) pass = (nDone == NS) ? 4 : 3; }
        else pass = (l >= sh && l >= g) ? 0 : (sh >= g ? 1 : 2);
      }
      if (pass == 1 || pass == 2) mySlot = q_pop(pass == 1 ? Q_SHADE : Q_GEN, true);
      txn_end();
      if (pass == 0) mySlot = leaf_pop();
      PT_SUB(tTxn);
      // =========================================================================
    } else PT_SUB(tLocal);
    PT_STAMP(tSwap);
    if (pass == 4) break;
    if (pass == 3) { if (CNT) idleSpins++; __builtin_amdgcn_s_sleep(32); PT_SUB(tIdle); PT_STAMP(tSwap); continue; }
    if (pass == 0) { leaf_pass(mySlot); PT_STAMP(tLeaf); continue; }
    if (pass > 0) { run_batch(mySlot, pass == 1); PT_STAMP(tBatch); continue; }

    // ---- node loop: until swapLanes lanes have left the node set ----
    if (CNT) { nodeRuns++; ringBacklog += (unsigned long long)nqCount; leafBacklog += (unsigned long long)lqCount; }
    {
      SlotStack st = make_stack(ns >= 0 ? ns : 0);
      for (;;) {
        const bool atNode = ns >= 0 && ntv.node >= 0 && ntv.node != kTravDone;
        const unsigned long long m = __ballot(atNode);
        const int n = __popcll(m);
        if (n == 0 || nActive - n >= a.swapLanes) break;
        if (CNT) { nodeSteps++; nodeLanes += (uint32_t)n; }
        if (atNode) trav_node_step<CNT>(sc, nray, ntv, st, ct);
      }
    }
    PT_STAMP(tNode);
  }

  if constexpr (CNT) {
    unsigned long long* c = a.counters;
    const uint32_t v[9] = { wave_sum(ct.samples), wave_sum(ct.primaryRays), wave_sum(ct.bounceRays), wave_sum(ct.shadowRays),
                            wave_sum(ct.nodeFetches), wave_sum(ct.triTests), wave_sum(ct.closestHits), wave_sum(ct.lightLoads),
                            wave_sum(ct.analyticTests) };
    if (lane == 0) {
      for (int i = 0; i < 9; i++) atomicAdd(&c[i], (unsigned long long)v[i]);
      atomicAdd(&c[9], (unsigned long long)nodeSteps + leafPasses);
      atomicAdd(&c[10], (unsigned long long)nodeLanes + leafLanes);
      atomicAdd(&c[11], (unsigned long long)batches);
      atomicAdd(&c[12], (unsigned long long)batchLanes);
      atomicAdd(&c[14], (unsigned long long)idleSpins);
      atomicAdd(&c[16], tBatch); atomicAdd(&c[17], tSwap); atomicAdd(&c[18], tNode); atomicAdd(&c[19], tLeaf);
      atomicAdd(&c[21], __builtin_amdgcn_s_memtime() - tStart);
      atomicAdd(&c[22], (unsigned long long)leafPasses); atomicAdd(&c[23], (unsigned long long)leafLanes);
      atomicMax(&c[38], (unsigned long long)__builtin_amdgcn_s_memrealtime());   // last wave out
      atomicMin(&c[36], rtStart);
      atomicAdd(&c[24], tLocal); atomicAdd(&c[25], tLock); atomicAdd(&c[26], tTxn); atomicAdd(&c[27], tIdle);
      atomicAdd(&c[39], nodeRuns); atomicAdd(&c[15], ringBacklog); atomicAdd(&c[13], leafBacklog);
      atomicAdd(&c[33], nIterResult); atomicAdd(&c[34], nIterLights); atomicAdd(&c[35], nIterGen);
      atomicAdd(&c[28], tBLoad); atomicAdd(&c[29], tBRun); atomicAdd(&c[30], tBStore); atomicAdd(&c[31], nTxn); atomicAdd(&c[32], nIter);
    }
  }
}

}  // namespace

// queuekernel_lean.hip compiles this file a second time (other slot count, stack depth and occupancy target) under other names
#ifndef PT_QK_EXPORT
#define PT_QK_EXPORT(name) name
#endif
int PT_QK_EXPORT(queuekernel_lds_stack_entries)() { return kStackN; }
int PT_QK_EXPORT(queuekernel_slots)() { return kP; }
size_t PT_QK_EXPORT(queuekernel_cold_bytes)(int nBlocks) { return (size_t)nBlocks * kWaves * kP * sizeof(SlotCold); }
size_t PT_QK_EXPORT(queuekernel_overflow_ints)(int nBlocks, int ovfDepth) { return (size_t)nBlocks * kWaves * kP * (size_t)ovfDepth; }

template <bool CNT, bool FAST>
static void launch_qk(dim3 grid, dim3 block, hipStream_t stream, const LaunchArgs& a) {
  if (a.scene.shadowNearest) pt_queuekernel<CNT, true, FAST, true><<<grid, block, 0, stream>>>(a);
  else                       pt_queuekernel<CNT, true, FAST, false><<<grid, block, 0, stream>>>(a);
}
hipError_t PT_QK_EXPORT(launch_queuekernel)(hipStream_t stream, const LaunchArgs& a, int nBlocks, bool counted, bool fastShading) {
  dim3 grid(nBlocks), block(kBlockThreads);
  // fastShading: opt-in approximate BRDF arithmetic (pt_disney.h ShadeMath)
  if (fastShading) { if (counted) launch_qk<true, true>(grid, block, stream, a); else launch_qk<false, true>(grid, block, stream, a); }
  else             { if (counted) launch_qk<true, false>(grid, block, stream, a); else launch_qk<false, false>(grid, block, stream, a); }
  return hipGetLastError();
}

}  // namespace pt
