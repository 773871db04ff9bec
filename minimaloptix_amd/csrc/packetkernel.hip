// packetkernel.hip -- megakernel variant 4: the scheduler of variant 3 (queuekernel.hip: path slots with their traversal
// state in LDS, node / leaf / shade / gen queues shared by the workgroup, lanes as workers) around the PER-BOUNCE state
// machine of pt_packet.h.
//
// Variant 3 gives every ray its own shading visit: a Disney hit with three facing lights is four trips through the
// shade queue, four slot-record round trips and four traversal set-ups, one after the other.  Here a visit does all of a
// hit's work that does not depend on a trace (every light draw, the facing tests, disneyPdf / disneyEval per light, the
// BRDF sample, the seed fork) and leaves a PACKET: up to three shadow rays and the continuation ray, all from the hit
// point.  The slot then traces the packet's rays one after the other WITHOUT going back to shading: when a ray ends and
// another one is pending, the next leaf pass loads that ray (16 bytes of the slot record) and restarts the slot at the
// root.  The visit that follows folds the shadow results into the radiance in light order and shades the continuation's
// hit.  Same random draws, same decisions, same floating-point operations on the path's values as variants 0-3 (the
// images are bit-identical); a path of depth k is a chain of k visits instead of up to 4k.
//
// Deep paths (LaunchArgs::auxDepth) go one step further: the shadow rays of a hit are traced in slots BORROWED from
// finished paths, at the same time as the continuation, and joined through three bits of the path slot's LDS flag word
// (see PoolLds::freeMask, SlotSink, store_flags / arrive_parks / aux_done).  A launch's tail is a handful of paths
// walking the 256-bounce cap; this makes each of their bounces one dependent ray instead of up to four.
//
// What a pass touches (round 4):
//   * a finished packet is routed by what its continuation hit (kShadeFlag): Disney triangle hits to Q_SHADE, whose batches run the
//     Disney program on full waves, everything else to Q_GEN, whose visit also takes the next work item;
//   * the leaf pass makes ONE memory round trip: triangle records only -- no hit row (a tie at exactly tbest reads it), no material
//     (Tri48::shadow says what a triangle is to a shadow ray), the packet's next ray only where this ray can end;
//   * shade batches and leaf passes read the launch arguments through fresh_args(): nothing of the scene sits in scalar
//     registers while the node loop runs;
//   * 576 path slots x 9 LDS stack entries per workgroup: rings sized exactly, 53,104 of the 53,760 bytes three workgroups per CU get.
//
//   * round 6: a workgroup that is down to its last paths (the launch's drain) stops shading partial batches and, once every path it has left waits in
//     Q_SHADE / Q_GEN, hands them to drainkernel.hip BEHIND the main loop and leaves (the block after the loop says why it is there and not in it);
//   * this file is compiled twice (Makefile): as it stands for the 64-byte-node instantiations, with LLVM's max-ilp scheduling, and through
//     packetkernel_n128.hip (PT_PK_N128) for the 128-byte-node ones with the default strategy -- each is a few per cent slower under the other's.
//
// Limits (moptix_api.hip falls back to variant 3 otherwise): at most kPacketShadows lights, no Disney material on an
// analytic primitive (shadow rays then need the brute-force lists at every ray start).
#include <hip/hip_runtime.h>

#include "megakernel.h"
#include "pt_path.h"
#include "pt_packet.h"
#include "pt_slot.h"

namespace pt {

namespace {

constexpr int kBlockThreads = 256;
constexpr int kWaves = kBlockThreads / 64;
// Path slots per wave and LDS stack entries per slot: what a workgroup's third of the CU's LDS (53,760 bytes for three workgroups:
// 42 blocks of 1,280; a 54,128-byte build ran two per CU) is spent on.  A slot costs 32 B of ray + 4 (entries + 1) B of stack + 12 B of ring space.  Measured on
// coffee at 64 spp, same bits (profiles/r04_slots.txt): 512 slots x 11 entries 95.3 ms, 544 x 10 93.8, 576 x 9 93.3, 608 x 8 93.9,
// 640 x 7 94.5 -- more paths in flight against more stack entries spilled to HBM.  (queuekernel.hip keeps its own 512 x 11.)
#ifndef PT_PK_KP
#define PT_PK_KP 144
#endif
#ifndef PT_PK_STACKN
#define PT_PK_STACKN 9
#endif
#define PT_KP PT_PK_KP
#define PT_STACKN PT_PK_STACKN
#ifndef PT_WAVES_PER_SIMD
#define PT_WAVES_PER_SIMD 3
#endif
constexpr int kP = PT_KP;               // slots per wave
constexpr int kStackN = PT_STACKN;      // LDS stack entries per slot; deeper levels spill to HBM
constexpr int kWavesPerSimd = PT_WAVES_PER_SIMD;   // occupancy target: 3 workgroups per CU (VGPR <= 168, LDS <= 53 KB)
// Rings hold exactly NS entries (a slot sits in at most one ring): sized to the next power of two they would cost 640 -> 1024
// entries each and the workgroup its third of the CU's LDS.  An index is head + count + rank < 2 NS: one conditional subtract.
constexpr int ring_capacity(int n) { return n; }
template <int NS> __device__ __forceinline__ int ring_wrap(int i) {
  if constexpr ((NS & (NS - 1)) == 0) return i & (NS - 1); else return i >= NS ? i - NS : i;
}
static_assert(PT_KP * kWaves <= (1 << kSlotBits), "slot ids are 10 bits");


// pool-wide queues first (they index PoolLds::queue); Q_NODE / Q_LEAF are per-wave rings (WavePriv)
enum { Q_SHADE = 0, Q_GEN = 1, kNumQ = 2, Q_NODE = 2, Q_LEAF = 3, DEST_DONE = 4, DEST_NONE = -1 };

// LDS image of NS slots
template <int NS>
struct PoolLds {
  // only what the node loop touches lives in LDS: 32 B + the stack per slot
  v4 nodeA[NS];           // o.xyz, tbest
  v4 nodeB[NS];           // d.xyz, node (int bits); a lane that picks the slot up for the node loop derives 1/d from it
  int stack[NS][kStackN + 1];   // [0] = sp | packet description (see the flag bits below), [1..] = entries
  unsigned short queue[kNumQ][ring_capacity(NS)];
  int qHead[kNumQ], qCount[kNumQ];   // SHARED only
  int done, lock;                    // SHARED only
  int drainAt;                       // once this many slots are done the pool stops shading and hands its last paths to the drain kernel (set at start-up)
  // Concurrent shadow rays for deep paths (LaunchArgs::auxDepth): a slot whose path is done (the work items ran out, or
  // LaunchArgs::slotsInUse left it without one) can be borrowed by a deep path, one per shadow ray of a hit, so that
  // the shadow rays and the continuation are traced at the same time: the launch's tail is a few capped paths walking
  // their 257 bounces, and this divides the time each bounce takes.
  unsigned int freeMask[NS / 32];    // bit set: the slot carries no path and nobody has borrowed it
  int nFree;                         // set bits in freeMask (approximate: read without the lock)
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
  // pt_path.h node_step_nearfar's branch-free tail: m pushes on top of sp entries stay in LDS and entry sp exists (the dead
  // store of a step that enters nothing lands there); the entry a pop would take (sp - 1; for sp = 0 the slot's flag word, unused)
  static constexpr bool kFlat = true;
  __device__ __forceinline__ bool fits_fast(int sp, int m) const { return sp + m <= kStackN && sp < kStackN; }
  __device__ __forceinline__ int peek_fast(int sp) const { return lds[min(sp, kStackN) - 1]; }
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

// The launch arguments again, read from the kernel-argument segment through a pointer the compiler cannot see through.
// The kernel is one loop; whatever it reads from its arguments is loop-invariant, so the compiler loads ALL of it in front
// of the loop and keeps it in scalar registers for the whole launch: camera (19 words), background, a dozen table pointers,
// the hand-out parameters -- about 100 words against 102 registers, and the rest of the kernel's scalar state (queue heads,
// exec masks of the scheduler) then lives in lanes of a spill VGPR, one v_readlane / v_writelane (4.4 clocks of the vector
// pipe each, tools/micro/valu_issue.hip) per access: 111 spilled scalar registers in round 3.  A pass that needs the scene
// (shading batch, leaf pass) therefore takes its own copy here: s_load where the value is used, dead when the pass ends,
// and only what the node loop and the scheduler need stays resident.
__device__ __forceinline__ const LaunchArgs& fresh_args() {
  typedef __attribute__((address_space(4))) const LaunchArgs CA;
  unsigned long long p = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return *(const LaunchArgs*)(CA*)p;
}

// destination queue of a slot after a traversal step / a ray set-up
__device__ __forceinline__ int route(int node, int kind, int bestPrim) {
  if (node == kTravDone) return (kind == RK_SHADOW || bestPrim >= 0) ? Q_SHADE : Q_GEN;
  return node >= 0 ? Q_NODE : Q_LEAF;
}

// pt_packet.h's sink for the shadow rays of a new packet: each goes to the slot as soon as it is known (weight row; the ray
// itself to LDS if it is the first of the list, else to its ray row) -- or, for a deep path that has borrowed slots, each
// shadow ray goes to its own borrowed slot, to be traced at the same time as the continuation.
struct SlotSink {
  SlotCold* cold; int slot; v4* nodeA; v4* nodeB; int (*stack)[kStackN + 1]; const PathState* ps; int root;
  int axp;                              // this path's three borrowed slots (10 bits each) or -1
  __device__ __forceinline__ void shadow(int j, v3 d, float tmax, v3 w, float inv) const {
    SlotCold* cw = at32(cold, slot);
    slot_store(&cw->pend[j], mk4(w.x, w.y, w.z, inv));
    if (axp >= 0) {
      const int ax = (axp >> (kSlotBits * j)) & kSlotMask;
      nodeA[ax] = mk4(ps->o.x, ps->o.y, ps->o.z, tmax); nodeB[ax] = mk4(d.x, d.y, d.z, i2f(root));
      stack[ax][0] = kAuxSlot | kShadowRay | (1 << kNShShift) | aux_parent_bits(slot);      // a packet of one shadow ray
    } else if (j == 0) { nodeA[slot] = mk4(ps->o.x, ps->o.y, ps->o.z, tmax); nodeB[slot] = mk4(d.x, d.y, d.z, i2f(root)); }   // ps->o: the hit point
    else slot_store(&cw->ray[j - 1], mk4(d.x, d.y, d.z, tmax));
  }
};

// NEAR: the scene has a Disney GLASS material, shadow rays keep their nearest any-hit candidate (pt_path.h, rule D5).  A template
// parameter so that scenes without one -- the benchmark scene -- run code in which that logic does not exist.
template <bool CNT, bool SHARED, bool FAST = false, bool NEAR = false, bool N64 = false>
__global__ void __launch_bounds__(kBlockThreads, kWavesPerSimd) pt_packetkernel(const LaunchArgs a) {
  constexpr int NS = SHARED ? kP * kWaves : kP;          // slots per pool
  __shared__ PoolLds<NS> sPool[SHARED ? 1 : kWaves];
  __shared__ WavePriv<NS> sPriv[kWaves];
  // -DPT_EVLOG (tools/gpu_lone_path.py EVLOG=1; not part of the product build): once a visit has seen a path deeper than 100 bounces, every
  // wave of that workgroup logs its passes with 100 MHz stamps -- the timeline of one capped path walking in an idle machine
#ifdef PT_EVLOG
  __shared__ int sEvOn;
  if (threadIdx.x == 0) sEvOn = 0;
#define PT_EV(code, val) do { if (sEvOn && (threadIdx.x & 63) == 0) { const unsigned long long i_ = atomicAdd(a.evLog, 1ull); \
    if (i_ < 65000ull) a.evLog[1 + i_] = (__builtin_amdgcn_s_memrealtime() << 24) | (((unsigned long long)(val) & 0xffffull) << 8) | ((unsigned long long)(threadIdx.x >> 6) << 4) | (unsigned long long)(code); } } while (0)
#else
#define PT_EV(code, val) do { } while (0)
#endif

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
    if (pred) W.queue[q][ring_wrap<NS>(qHead[q] + qCount[q] + lane_rank(m))] = (unsigned short)slot;
    qCount[q] += __popcll(m);
  };
  auto q_pop = [&](int q, bool want) -> int {
    const unsigned long long m = __ballot(want);
    const int n = min((int)__popcll(m), qCount[q]);      // (int): min(long long, int) resolves to the double overload
    int slot = -1;
    if (want) { const int r = lane_rank(m); if (r < n) slot = W.queue[q][ring_wrap<NS>(qHead[q] + r)]; }
    qHead[q] = ring_wrap<NS>(qHead[q] + n);
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
    st.ovf = ovfBase ? reinterpret_cast<int*>(reinterpret_cast<char*>(ovfBase) + (uint32_t)((uint32_t)slot * (uint32_t)a.ovfDepth * 4u)) : nullptr;      // 32-bit offset: a pool's overflow area is far below 4 GB
    return st;
  };

  // Flag word of a slot after a traversal step.  While the slot's shadow rays are out in borrowed slots, other waves
  // count down in the join bits of the word: the owner then leaves those bits alone.
  auto store_flags = [&](int slot, int value) {
    if (__builtin_expect((value & kHasAux) != 0, 0)) { atomicAnd(&W.stack[slot][0], kJoinMask); atomicOr(&W.stack[slot][0], value & ~kJoinMask); }
    else W.stack[slot][0] = value;
  };
  // The path slot's own ray is done.  True = some shadow rays are still out: the slot parks (in no queue) and the last
  // of them pushes it to Q_SHADE.
  auto arrive_parks = [&](int slot) -> bool {
    const int old = atomicOr(&W.stack[slot][0], kArrived);
    return ((old >> kJoinShift) & 3) != 0;
  };
  // A borrowed slot's shadow ray is done (its attenuation is in its flag word / att row, read by the path slot's next
  // visit).  Returns the path slot if this was the last one out and the path slot's own ray is done as well (the caller
  // pushes it to Q_SHADE), else -1.
  auto aux_done = [&](int fl, int& dest) -> int {
    const int parent = aux_parent(fl);
    const int old = atomicSub(&W.stack[parent][0], 1 << kJoinShift);
    const bool last = (old & kJoinMask) == ((1 << kJoinShift) | kArrived);
    dest = last ? ((old & kShadeFlag) ? Q_SHADE : Q_GEN) : DEST_NONE;      // kArrived in `old`: the path slot's own flags are final
    return last ? parent : -1;
  };

  // ---- start-up: every slot in use needs a work item ----
  // Paths in flight = slots in use; by Little's law a ray spends (slots in use) / (rays per second) in the scheduler,
  // about 70 us with all 512 slots of every pool, so a path that bounces to the depth cap (about 1000 dependent rays)
  // takes 70-90 ms however short the launch is.  A short launch (one rank's share of a multi-GPU frame) therefore
  // uses fewer slots: a little less throughput, a much shorter critical path (LaunchArgs::slotsInUse, moptix_api.hip).
  const bool auxOn = SHARED && a.auxDepth > 0 && sc.rootRef >= 0;
  const int nUse = (a.slotsInUse > 0 && a.slotsInUse < NS) ? (SHARED ? a.slotsInUse : max(64, a.slotsInUse / kWaves)) : NS;
  {
    if (SHARED && threadIdx.x < NS / 32) {      // slots without a path can be borrowed from the start
      const int lo = (int)threadIdx.x * 32;
      W.freeMask[threadIdx.x] = !auxOn || nUse >= lo + 32 ? 0u : (nUse <= lo ? ~0u : ~0u << (nUse - lo));
    }
    if (threadIdx.x == 0) W.nFree = auxOn ? NS - nUse : 0;
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
        W.qCount[Q_GEN] = nUse; const int below = a.workCounter[kDrainList + kDrainBelow];
        // drainAt: "slots without a path" from which on the pool hands over.  Slots that never carried a path (NS - nUse) are in that count from the
        // start, so the threshold is at least one above it: a slot only joins the count when it has found the work counter exhausted, and a pool
        // must never stop shading while work items are left (slots_in_use <= drain_below would otherwise drain from the first transaction on,
        // with most of the launch's items unrendered -- found by the option fuzzer, seed 103 case 252)
        W.done = NS - nUse; W.drainAt = below > 0 ? max(NS - below, NS - nUse + 1) : 0x7fff; W.lock = 0;
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __syncthreads();
    } else {
      qCount[Q_GEN] = nUse; nDone = NS - nUse;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
  }

  Counters ct = {};
  // counting build: 16-byte rows of the slot record moved, by who moves them (DESIGN.md "Slot rows"):
  // [0] shading visit loads, [1] shading visit stores (incl. the packet's ray / weight rows), [2] leaf pass loads, [3] leaf pass stores
  uint32_t rows[4] = { 0, 0, 0, 0 };
#define PT_ROWS(i, n) do { if (CNT) rows[i] += (uint32_t)(n); } while (0)
  uint32_t nodeSteps = 0, nodeLanes = 0, leafPasses = 0, leafLanes = 0, batches = 0, batchLanes = 0, idleSpins = 0;
  unsigned long long tBatch = 0, tSwap = 0, tNode = 0, tLeaf = 0, tStamp = 0;
  const unsigned long long tStart = CNT ? __builtin_amdgcn_s_memtime() : 0ull;
  [[maybe_unused]] const unsigned long long rtStart = CNT ? __builtin_amdgcn_s_memrealtime() : 0ull;
#define PT_STAMP(acc) do { if (CNT) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc += now_ - tStamp; tStamp = now_; } } while (0)
  tStamp = tStart;

  // ---- node-loop worker context: the only per-lane state that survives between passes ----
  int ns = -1;                 // slot this lane is walking, -1 = none
  int nsFlag = 0;              // packet description bits of that slot (everything above the stack pointer)
  PathState nray;              // o, tmin used
  nray.o = mk3(0, 0, 0); nray.tmin = sc.epsT; nray.d = mk3(0, 0, 1); nray.tmax = 0; nray.kind = RK_RADIANCE; nray.mode = M_TRACE;
  Trav ntv;                    // inv, tbest, node, sp used
  ntv.node = kTravDone; ntv.sp = 0; ntv.tbest = 0; ntv.inv = mk3(0, 0, 0); ntv.noi = mk3(0, 0, 0); ntv.bestPrim = -1; ntv.bestTri = -1;
  ntv.beta = 0; ntv.gamma = 0; ntv.att = mk3(1, 1, 1); ntv.started = 1;

  // result of the last pass, queued inside the next transaction
  int pendSlot = -1, pendDest = DEST_NONE;

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
        if (dest == Q_NODE) myNodeQ[ring_wrap<NS>(nqHead + nqCount + lane_rank(m))] = (unsigned short)slot;
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

  // ---- leaf pass: the slots popped from Q_LEAF stand at a leaf ----
  auto leaf_pass = [&](int slot) {
    const LaunchArgs& a = fresh_args();                 // this pass's own view of the arguments (see fresh_args)
    SceneView scl = a.scene; scl.shadowNearest = NEAR ? 1 : 0;
    const SceneView& sc = scl;
    const bool have = slot >= 0;
    if (CNT) { leafPasses++; leafLanes += (uint32_t)__popcll(__ballot(have)); }
    PT_EV(7, __popcll(__ballot(have)));
    pendSlot = slot; pendDest = DEST_NONE;
    // slot records written by earlier passes of this wave (other lanes) or published by other waves: the stores
    // of the previous pass were left in flight, so they are ordered here, where their latency has already elapsed
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (have) {
      const v4 na = W.nodeA[slot], nb = W.nodeB[slot];
      const int fl = W.stack[slot][0];
      const int node0 = f2i(nb.w);
      const bool isSwitch = node0 == kSwitchRef;
      const bool shadow = (fl & kShadowRay) != 0, hitValid = (fl & kHitValid) != 0;
      const int cur = fl_cur(fl);
      const SlotCold* cs = at32(cold, slot);
      // one round trip: the leaf's triangles and whichever row of the slot record this visit needs
      LeafChunk ch;
      leaf_fetch4(sc, isSwitch ? make_leaf_ref(0, 1) : node0, 0, ch);
      v4 wr = mk4(0.f, 0.f, 1.f, 0.f), wn = mk4(0.f, 0.f, 1.f, 0.f);
      const bool more = fl_more(fl);
      // ray cur+1 of the packet sits in ray[cur]: needed if this ray ends here -- a ray switch, a shadow ray (its any-hit program may
      // end it at any leaf) or a leaf reached with an empty stack; a radiance ray with entries on its stack goes on after this leaf
      if (more && (isSwitch || shadow || (fl & kSpMask) == 0)) { wn = slot_load(&cs->ray[cur]); PT_ROWS(2, 1); }
      if (!isSwitch) {
        if (shadow) { if (fl_stat(fl, cur) == 2) { wr = slot_load(&cs->att[cur]); PT_ROWS(2, 1); } }
        // the continuation's hit row stays where it is: tbest (LDS) decides every candidate except one at EXACTLY tbest, and
        // only then the primitive id of the hit the slot holds is needed (below).  The row is in a 100 MB pool, the slowest
        // fetch of this pass by far.
      }
      // the packet's next ray: same origin, restart at the root
      auto next_ray = [&](int flags) {
        const int nxt = cur + 1;
        const bool nshadow = nxt < fl_nsh(flags);
        W.nodeB[slot] = mk4(wn.x, wn.y, wn.z, i2f(sc.rootRef));
        W.nodeA[slot].w = wn.w;
        W.stack[slot][0] = (flags & ~(kSpMask | (3 << kCurShift) | kShadowRay)) | (nxt << kCurShift) | (nshadow ? kShadowRay : 0);
        pendDest = sc.rootRef >= 0 ? Q_NODE : Q_LEAF;
      };
      if (isSwitch) {
        next_ray(fl);
      } else {
        PathState ps; Trav tv;
        ps.tmin = sc.epsT; ps.mode = M_TRACE;
        ps.o = mk3(na.x, na.y, na.z); ps.d = mk3(nb.x, nb.y, nb.z);
        ps.kind = shadow ? RK_SHADOW : RK_RADIANCE;
        ps.tmax = shadow ? na.w : kRtDefaultMax;
        tv.inv = mk3(0.f, 0.f, 0.f); tv.tbest = na.w;      // the leaf step does not use 1/d
        tv.node = node0; tv.sp = fl & kSpMask;
        tv.bestTri = -1; tv.bestPrim = -1; tv.beta = 0.f; tv.gamma = 0.f; tv.att = mk3(1.f, 1.f, 1.f); tv.bestCls = SHADOW_NONE;
        if (shadow) { if (fl_stat(fl, cur) == 2) { tv.att = mk3(wr.x, wr.y, wr.z); if (sc.shadowNearest) tv.bestPrim = f2i(wr.w); } }
        else if (hitValid) tv.bestPrim = kPrimUnknown;       // "some primitive at tbest": any candidate at exactly tbest passes potential() for now
        const int oldPrim = tv.bestPrim;
        const v3 oldAtt = tv.att;
        SlotStack st = make_stack(slot);
        trav_leaf_step_fetched<CNT>(sc, ps, tv, st, ct, ch);
        int nfl = fl & ~kSpMask;
        if (shadow) {
          if (sc.shadowNearest) {
            // the verdict of the nearest any-hit surface so far and that surface's id (equal-t rule) travel in the att row
            if (tv.bestPrim != oldPrim) {
              nfl = (nfl & ~(3 << (kStatShift + 2 * cur))) | (2 << (kStatShift + 2 * cur));
              slot_store(&at32(cold, slot)->att[cur], mk4(tv.att.x, tv.att.y, tv.att.z, i2f(tv.bestPrim))); PT_ROWS(3, 1);
            }
          } else if (tv.att.x != oldAtt.x || tv.att.y != oldAtt.y || tv.att.z != oldAtt.z) {
            const bool zero = tv.att.x == 0.f && tv.att.y == 0.f && tv.att.z == 0.f;      // disneyAnyHit on an opaque surface
            nfl = (nfl & ~(3 << (kStatShift + 2 * cur))) | ((zero ? 1 : 2) << (kStatShift + 2 * cur));
            if (!zero) { slot_store(&at32(cold, slot)->att[cur], mk4(tv.att.x, tv.att.y, tv.att.z, 0.f)); PT_ROWS(3, 1); }
          }
        } else if (tv.bestPrim != oldPrim) {
          // most leaf visits find nothing nearer: the hit row is only written when it changed (primitive ids are unique, so a new
          // hit is a new bestPrim).  A candidate at exactly the old tbest ties with the hit the row holds: the lower primitive id
          // wins (rule D5), and this is the one case that reads the row.
          bool keep = true;
          if (hitValid && tv.tbest == na.w) { const v4 held = slot_load(&cs->hit); PT_ROWS(2, 1); keep = !(f2i(held.y) < tv.bestPrim); }
          if (keep) {
            slot_store(&at32(cold, slot)->hit, mk4(i2f(tv.bestTri), i2f(tv.bestPrim), tv.beta, tv.gamma)); PT_ROWS(3, 1);
            nfl = (nfl & ~kShadeFlag) | kHitValid | (tv.bestCls == SHADOW_OPAQUE ? kShadeFlag : 0);
          }
        }
        if (tv.node == kTravDone && more) {
          next_ray(nfl);                                                    // this ray is done, the packet is not: on to its next ray
        } else {
          W.nodeA[slot].w = tv.tbest;
          W.nodeB[slot].w = i2f(tv.node);
          store_flags(slot, tv.sp | nfl);
          pendDest = tv.node == kTravDone ? ((nfl & kShadeFlag) ? Q_SHADE : Q_GEN) : (tv.node >= 0 ? Q_NODE : Q_LEAF);
          if (__builtin_expect(tv.node == kTravDone && (nfl & (kHasAux | kAuxSlot)) != 0, 0)) {      // rare: deep paths only
            if (nfl & kAuxSlot) {
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");        // the att row written above is read by the path slot's visit
              pendSlot = aux_done(nfl, pendDest);
            } else if (arrive_parks(slot)) pendDest = DEST_NONE;
          }
        }
      }
    }
    // Slot records in HBM are re-read by other lanes / waves.  With a shared pool a slot only reaches another wave
    // through a queue transaction, whose release fence (txn_end) covers these stores: no wait here, the store
    // latency overlaps with the bookkeeping that follows.
    if constexpr (!SHARED) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    PT_EV(8, 0);
  };

  // ---- shading / regeneration batch: run the path state machine for the popped slots ----
  auto run_batch = [&](int slot, bool shadeBatch) {
    const LaunchArgs& a = fresh_args();                 // this pass's own view of the arguments (see fresh_args)
    SceneView scl = a.scene; scl.shadowNearest = NEAR ? 1 : 0;
    const SceneView& sc = scl;
    const bool have = slot >= 0;
    if (CNT) { batches++; batchLanes += (uint32_t)__popcll(__ballot(have)); }
    PT_EV(1, __popcll(__ballot(have)));
    PT_SUB0();
    pendSlot = slot; pendDest = DEST_NONE;
    PathState ps; Trav res;
    ps.mode = M_DONE; ps.kind = RK_RADIANCE;
    res.node = kTravDone; res.bestPrim = -1;
    // slot records written by earlier passes of this wave (other lanes) or published by other waves: the stores
    // of the previous pass were left in flight, so they are ordered here, where their latency has already elapsed
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    Packet pk; packet_clear(pk);
    v3 att[kPacketShadows] = { mk3(1.f, 1.f, 1.f), mk3(1.f, 1.f, 1.f), mk3(1.f, 1.f, 1.f) };
    int axp = -1;                                      // the three slots this path has borrowed for its shadow rays (10 bits each), -1 = none
    if (have) {
      const SlotCold* cs = at32(cold, slot);
      const int fl = W.stack[slot][0];
      const bool hitValid = (fl & kHitValid) != 0;
      const int nSh = (fl >> kPendShift) & 3;          // 0 for a slot that waits for a work item (flags are cleared then)
      const bool hadAux = (fl & kHasAux) != 0;         // its shadow rays were traced by the borrowed slots
      const i4 ctl = slot_load(&cs->ctl);
#ifdef PT_EVLOG
      if (ctl.y >= 100) sEvOn = 1;
#endif
      const v4 thrIn = slot_load(&cs->thr), radIn = slot_load(&cs->rad);
      v4 wh = mk4(0.f, 0.f, 0.f, 0.f), wb = mk4(1.f, 1.f, 1.f, 1.f);
      v4 wp[kPacketShadows], wa[kPacketShadows];
      if (ctl.w & kCtlHoldsAux) axp = f2i(thrIn.w);
      PT_ROWS(0, 3);
      if (hitValid) { wh = slot_load(&cs->hit); PT_ROWS(0, 1); }
      if (fl & kHasScale) { wb = slot_load(&cs->bsc); PT_ROWS(0, 1); }
#pragma unroll
      for (int i = 0; i < kPacketShadows; i++) {
        wp[i] = mk4(0.f, 0.f, 0.f, 0.f); wa[i] = mk4(1.f, 1.f, 1.f, 0.f);
        if (i < nSh) {
          wp[i] = slot_load(&cs->pend[i]); PT_ROWS(0, 1);
          const int ax = (axp >> (kSlotBits * i)) & kSlotMask;                 // only used with hadAux
          const int stt = hadAux ? fl_stat(W.stack[ax][0], 0) : fl_stat(fl, i);
          if (stt == 2) { wa[i] = hadAux ? slot_load(&at32(cold, ax)->att[0]) : slot_load(&cs->att[i]); PT_ROWS(0, 1); }
          else if (stt == 1) wa[i] = mk4(0.f, 0.f, 0.f, 0.f);
        }
      }
      const v4 na = W.nodeA[slot], wd = W.nodeB[slot];
      ps.mode = ctl.w & 7; ps.light = 0; ps.item = ctl.x; ps.depth = ctl.y; ps.seed = (uint32_t)ctl.z; ps.pixel = 0;
      pk.nShadow = (ctl.w >> 3) & 3; pk.hasBounce = (ctl.w >> 5) & 1; pk.hasScale = (ctl.w >> 6) & 1;
      pk.bscale = mk3(wb.x, wb.y, wb.z); pk.binv = wb.w;
#pragma unroll
      for (int i = 0; i < kPacketShadows; i++) { pk.pendW[i] = mk3(wp[i].x, wp[i].y, wp[i].z); pk.pendInv[i] = wp[i].w; att[i] = mk3(wa[i].x, wa[i].y, wa[i].z); }
      ps.thr = mk3(thrIn.x, thrIn.y, thrIn.z); ps.rad = mk3(radIn.x, radIn.y, radIn.z);
      ps.N = mk3(0.f, 0.f, 1.f); ps.mat = 0; ps.V = mk3(0.f, 0.f, 1.f);
      ps.pendW = mk3(0.f, 0.f, 0.f); ps.pendInv = 0.f; ps.accum = mk3(0, 0, 0); ps.cdlin = mk3(0.f, 0.f, 0.f);
      // the continuation is the packet's last ray: its origin, direction and tbest are what LDS holds now
      ps.o = mk3(na.x, na.y, na.z); ps.d = mk3(wd.x, wd.y, wd.z); ps.tmin = sc.epsT; ps.tmax = kRtDefaultMax;
      ps.kind = RK_RADIANCE;
      res.tbest = na.w; res.bestTri = -1; res.bestPrim = -1; res.beta = 0.f; res.gamma = 0.f; res.att = mk3(1.f, 1.f, 1.f);
      if (hitValid) { res.bestTri = f2i(wh.x); res.bestPrim = f2i(wh.y); res.beta = wh.z; res.gamma = wh.w; }
      if (ps.mode == M_TRACE) ps.mode = M_RESULT;
    }
    const SlotSink sink{ cold, slot >= 0 ? slot : 0, W.nodeA, W.nodeB, W.stack, &ps, sc.rootRef, axp };
    if (CNT) { __builtin_amdgcn_s_waitcnt(0); PT_SUB(tBLoad); }
#ifdef PT_EVLOG
    __builtin_amdgcn_s_waitcnt(0); PT_EV(2, 0);
#endif
    for (;;) {
      if (have && ps.mode == M_NEW_SAMPLE) {
        if (CNT) {      // finish-time histogram (1 ms buckets): how many samples end when, and how deep they were
          const unsigned long long b = min(255ull, (__builtin_amdgcn_s_memrealtime() - rtStart) / 100000ull);
          atomicAdd(a.counters + 40 + b, 1ull); atomicMax(a.counters + 296 + b, (unsigned long long)ps.depth); atomicAdd(a.counters + 552 + b, (unsigned long long)ps.depth);
        }
        store_sample(a, ps.item, ps.accum);
        if (CNT) atomicAdd(a.counters + 815, 1ull);
        if (a.tileCost != nullptr && ps.depth >= kDeepPath) atomicMax(a.tileCost + ((ps.item % a.nItems) >> a.unitShift), (unsigned int)ps.depth);
        ps.mode = M_NEW_PIXEL;
      }
      // lanes that arrive with a finished packet go first; the lanes that only need a new work item wait for them, so
      // that begin_sample + the ray set-up run once, for all of them together
      const bool resultFirst = __ballot(have && ps.mode == M_RESULT) != 0ull;
      const bool run = have && ps.mode != M_TRACE && ps.mode != M_DONE &&
                       !((shadeBatch || resultFirst) && ps.mode == M_NEW_PIXEL);
      if (__ballot(run) == 0ull) break;
      if (CNT) {   // which state-machine stages this iteration executes (wave level)
        if (__ballot(run && ps.mode == M_RESULT)) nIterResult++;
        if (__ballot(run && ps.mode == M_NEW_PIXEL)) nIterGen++;
      }
      if (run) {
        if (ps.mode == M_RESULT) {
          on_result_packet<CNT, FAST, SlotSink>(sc, ps, pk, res, att, ct, sink);
        } else {  // M_NEW_PIXEL: next (pixel, sample) work item
          int k = atomicAdd(a.workCounter, 1);
          k = (k >= a.nWork) ? -1 : handout_to_item(a, k);
          if (CNT && k < 0) atomicMin(a.counters + 37, (unsigned long long)__builtin_amdgcn_s_memrealtime());   // first time the items ran out
          int s;
          if (k < 0) { ps.mode = M_DONE; }
          else if (item_to_pixel(a, k, s, ps.pixel)) { ps.item = k; begin_sample<CNT>(sc, ps, a.seeds[s], ct); packet_primary(pk); }
        }
      }
    }
    if (CNT) { __builtin_amdgcn_s_waitcnt(0); PT_SUB(tBRun); }
#ifdef PT_EVLOG
    __builtin_amdgcn_s_waitcnt(0); PT_EV(3, 0);
#endif
    bool useAux = false;
    if (have) {
      SlotCold* cw = at32(cold, slot);
      // borrowed slots go back when the path has ended (the lane may hold a new path's camera ray by now)
      if (axp >= 0 && !(ps.mode == M_TRACE && ps.depth >= a.auxDepth)) {
        for (int j = 0; j < kPacketShadows; j++) { const int ax = (axp >> (kSlotBits * j)) & kSlotMask; atomicOr(&W.freeMask[ax >> 5], 1u << (ax & 31)); }
        atomicAdd(&W.nFree, kPacketShadows);
        axp = -1;
      }
      // a path that gets deep borrows three slots (one per possible shadow ray of a hit; used from its next hit on)
      int axpNext = axp;
      if (auxOn) {
        const bool wantAux = axp < 0 && ps.mode == M_TRACE && ps.depth >= a.auxDepth;
        if (wantAux && W.nFree >= kPacketShadows) {                // rare: lanes take their turn at the free list (LDS atomics)
          int got = 0, pack = 0;
          for (int wi = 0; wi < NS / 32 && got < kPacketShadows; wi++) {
            unsigned int cur = W.freeMask[wi];
            while (cur != 0u && got < kPacketShadows) {
              const int b = __builtin_ctz(cur);
              const unsigned int prev = atomicAnd(&W.freeMask[wi], ~(1u << b));       // claims the bit if it is still there
              if (prev & (1u << b)) { pack |= (wi * 32 + b) << (kSlotBits * got); got++; }
              cur = prev & ~(1u << b);
            }
          }
          if (got == kPacketShadows) { axpNext = pack; atomicSub(&W.nFree, kPacketShadows); }
          else for (int j = 0; j < got; j++) { const int ax = (pack >> (kSlotBits * j)) & kSlotMask; atomicOr(&W.freeMask[ax >> 5], 1u << (ax & 31)); }
        }
      }
      i4 ctl; ctl.x = ps.item; ctl.y = ps.depth; ctl.z = (int)ps.seed;
      ctl.w = ps.mode | (pk.nShadow << 3) | (pk.hasBounce << 5) | (pk.hasScale << 6);
      if (axpNext >= 0) ctl.w |= kCtlHoldsAux;
      slot_store(&cw->ctl, ctl);
      slot_store(&cw->thr, mk4(ps.thr.x, ps.thr.y, ps.thr.z, i2f(axpNext)));
      slot_store(&cw->rad, mk4(ps.rad.x, ps.rad.y, ps.rad.z, 0.f));
      PT_ROWS(1, 3);
      if (ps.mode == M_TRACE) {
        // new packet: the brute-force lists for the continuation (radiance) ray, then the rays in trace order: shadow rays
        // in light order, the continuation last; the first goes to LDS, the others to the ray rows
        float tb0 = kRtDefaultMax; int bp0 = -1;
        if (pk.hasBounce) {
          Trav tv;
          ps.kind = RK_RADIANCE;
          trav_begin<CNT, false>(sc, ps, tv, ct);
          tb0 = tv.tbest; bp0 = tv.bestPrim;
        }
        const bool hitNow = bp0 >= 0;
        if (hitNow) { slot_store(&cw->hit, mk4(i2f(-1), i2f(bp0), 0.f, 0.f)); PT_ROWS(1, 1); }
        // rows the packet's sink wrote during the visit (SlotSink::shadow): one weight row per shadow ray, one ray row for each after the first
        PT_ROWS(1, pk.nShadow + ((axp >= 0 || pk.nShadow == 0) ? 0 : pk.nShadow - 1));
        useAux = axp >= 0 && pk.nShadow > 0;               // the shadow rays sit in the borrowed slots (SlotSink)
        const int nOwnSh = useAux ? 0 : pk.nShadow;
        const int nRays = nOwnSh + pk.hasBounce;
        const v4 rb = mk4(ps.d.x, ps.d.y, ps.d.z, tb0);                                     // the continuation: last ray of the list
        if (nOwnSh == 0) {                                                                  // ... and the first one here
          W.nodeA[slot] = mk4(ps.o.x, ps.o.y, ps.o.z, rb.w);
          W.nodeB[slot] = mk4(rb.x, rb.y, rb.z, i2f(sc.rootRef));
        } else if (pk.hasBounce) {
          // the shadow rays are in place already (SlotSink); ray j >= 1 of the list lives in ray[j-1]
          if (pk.nShadow == 1) slot_store(&cw->ray[0], rb);
          else if (pk.nShadow == 2) slot_store(&cw->ray[1], rb);
          else slot_store(&cw->ray[2], rb);
          PT_ROWS(1, 1);
        }
        if (pk.hasScale) { slot_store(&cw->bsc, mk4(pk.bscale.x, pk.bscale.y, pk.bscale.z, pk.binv)); PT_ROWS(1, 1); }
        W.stack[slot][0] = (nOwnSh << kNShShift) | ((max(nRays, 1) - 1) << kNRayShift) | (nOwnSh > 0 ? kShadowRay : 0) | (pk.nShadow << kPendShift) |
                           (hitNow ? kHitValid : 0) | (pk.hasScale ? kHasScale : 0) |
                           (useAux ? kHasAux : 0);
        pendDest = sc.rootRef >= 0 ? Q_NODE : Q_LEAF;
        if (useAux) {
          // join bits: shadow rays out; a packet without a continuation has nothing of its own to trace and parks at once
          W.stack[slot][0] |= (pk.nShadow << kJoinShift) | (pk.hasBounce ? 0 : kArrived);
          if (!pk.hasBounce) pendDest = DEST_NONE;
        }
      } else {
        W.stack[slot][0] = 0;
        pendDest = (ps.mode == M_NEW_PIXEL) ? Q_GEN : DEST_DONE;
        if (auxOn && ps.mode != M_NEW_PIXEL) {               // no work item left for this slot: deep paths may borrow it
          atomicOr(&W.freeMask[slot >> 5], 1u << (slot & 31)); atomicAdd(&W.nFree, 1);
        }
      }
    }
    if (auxOn && __ballot(useAux) != 0ull) {                // the borrowed slots start at the root, with this wave
      for (int j = 0; j < kPacketShadows; j++) local_push((useAux && j < pk.nShadow) ? Q_NODE : DEST_NONE, (axp >> (kSlotBits * j)) & kSlotMask);
    }
    if constexpr (!SHARED) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    PT_SUB(tBStore);
    PT_EV(4, __popcll(__ballot(useAux)));
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
        int dest = DEST_NONE, pushSlot = ns;
        if (leave) {
          const int nodeOut = (ntv.node == kTravDone && fl_more(nsFlag)) ? kSwitchRef : ntv.node;   // ray done, packet not: the leaf pass switches rays
          W.nodeB[ns].w = i2f(nodeOut); store_flags(ns, ntv.sp | nsFlag);
          dest = (nodeOut == kTravDone) ? ((nsFlag & kShadeFlag) ? Q_SHADE : Q_GEN) : Q_LEAF;   // a lane leaves at a leaf, at a ray switch or finished
          if (__builtin_expect(nodeOut == kTravDone && (nsFlag & (kHasAux | kAuxSlot)) != 0, 0)) {     // rare: deep paths only
            if (nsFlag & kAuxSlot) {
              __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");    // att rows written by this wave's last leaf pass
              pushSlot = aux_done(nsFlag, dest);
            } else if (arrive_parks(ns)) dest = DEST_NONE;
          }
        }
        local_push(dest, pushSlot);
        if (leave) ns = -1;
      }
      if (nqCount > 0 && __ballot(ns < 0) != 0ull) {
        const unsigned long long m = __ballot(ns < 0);
        const int n = min((int)__popcll(m), nqCount);
        if (ns < 0) {
          const int r = lane_rank(m);
          if (r < n) {
            ns = myNodeQ[ring_wrap<NS>(nqHead + r)];
            const v4 na = W.nodeA[ns], nb = W.nodeB[ns];
            const int spw = W.stack[ns][0];
            nray.o = mk3(na.x, na.y, na.z); ntv.tbest = na.w;
            // 1/d by v_rcp_f32 (1 ulp): the slab test is conservative by far more than that (boxes are padded by 1e-5 of
            // the scene, pt_lbvh.h pad_lo/pad_hi), and which boxes are entered never changes the result (rule D5)
            ntv.inv = mk3(node_inv(nb.x), node_inv(nb.y), node_inv(nb.z)); ntv.node = f2i(nb.w);
            ntv.noi = neg_o_inv(nray.o, ntv.inv);
            ntv.sp = spw & kSpMask; nsFlag = spw & ~kSpMask;
          }
        }
        nqHead = ring_wrap<NS>(nqHead + n); nqCount -= n;
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
      PT_EV(9, nActive);
      txn_begin();
      for (int d = 0; d < 2; d++) {
        const int q = Q_SHADE + d;
        for (int base = 0; base < obCount[d]; base += 64) {
          const int i = base + lane;
          if (i < obCount[d]) W.queue[q][ring_wrap<NS>(qHead[q] + qCount[q] + i)] = sPriv[wave].outbox[d][i];
        }
        qCount[q] += obCount[d]; obCount[d] = 0;
      }
      nDone += localDone; localDone = 0;
      if (qCount[Q_SHADE] >= 64) pass = 1;
      else if (qCount[Q_GEN] >= 64) pass = 2;
      else if (lqCount >= 64) pass = 0;
      else if (starving) {
        const int l = lqCount;
        int sh = qCount[Q_SHADE], g = qCount[Q_GEN];
        // The pool is down to its last paths (LaunchArgs drain list, megakernel.h): no more partial shading batches.  Rays in flight are
        // walked to the end of their packets; once every path that is left waits in one of the two queues the workgroup leaves the loop and
        // hands them to the drain kernel (after the loop: hand-over).  Only this branch knows about it -- the pass selection of a full pool
        // is what it was.
        if (nDone >= W.drainAt) { if (nDone + sh + g == NS && nActive == 0 && l == 0) pass = 4; sh = 0; g = 0; }
        if (pass != 4) {
          if (l + sh + g == 0) { if (nActive == 0) pass = (nDone == NS) ? 4 : 3; }
          else pass = (l >= sh && l >= g) ? 0 : (sh >= g ? 1 : 2);
        }
      }
      if (pass == 1 || pass == 2) mySlot = q_pop(pass == 1 ? Q_SHADE : Q_GEN, true);
      txn_end();
      if (pass == 0) mySlot = leaf_pop();
      PT_EV(10, pass + 1);
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
      PT_EV(5, nActive);
#ifdef PT_EVLOG
      int evSteps = 0;
#endif
      for (;;) {
#ifdef PT_EVLOG
        evSteps++;
#endif
        const bool atNode = ns >= 0 && ntv.node >= 0 && ntv.node != kTravDone;
        const unsigned long long m = __ballot(atNode);
        const int n = __popcll(m);
        if (n == 0 || nActive - n >= a.swapLanes) break;
        if (CNT) { nodeSteps++; nodeLanes += (uint32_t)n; }
        if (atNode) trav_node_step<CNT, N64>(sc, nray, ntv, st, ct);
      }
#ifdef PT_EVLOG
      PT_EV(6, evSteps - 1);
#endif
    }
    PT_STAMP(tNode);
  }

  // ---- hand-over: the paths that are left go on in the drain kernel (drainkernel.hip) ----
  // The launch's drain is a few paths walking to the depth cap, one dependent pass after the other through this scheduler: 36 us per bounce
  // in an idle machine (profiles/r06_tail_anatomy.txt).  A workgroup that is down to LaunchArgs' drain threshold has stopped shading
  // (transaction above); every path it still holds has come back from its packet and sits in Q_SHADE or Q_GEN.  Its slot record already
  // holds the path (ctl thr rad hit bsc pend att); the 36 bytes that live in LDS go into the record's two spare rows, a borrowed slot's
  // verdict is copied to where an unborrowed packet would have left it, and the record's index goes on the drain list.  Outside the loop
  // on purpose: inside, as one more pass, the same code cost the benchmark frame 1 % through the loop's register allocation.
  if constexpr (SHARED) {
    __syncthreads();
    // (everything this block needs is taken afresh from the arguments: nothing stays live through the loop for its sake)
    const LaunchArgs& a = fresh_args();
    SlotCold* cold = reinterpret_cast<SlotCold*>(a.poolCold) + (size_t)blockIdx.x * NS;
    const int gpool = blockIdx.x;
    PoolLds<NS>& W = sPool[0];
    if (W.done != NS && __hip_atomic_load(a.workCounter + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {      // (not after the watchdog: the queues are in no defined state then)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const int nSh = W.qCount[Q_SHADE], nGe = W.qCount[Q_GEN];
      for (int e = threadIdx.x; e < nSh + nGe; e += kBlockThreads) {
        const int q = e < nSh ? Q_SHADE : Q_GEN, r = e < nSh ? e : e - nSh;
        const int slot = W.queue[q][ring_wrap<NS>(W.qHead[q] + r)];
        SlotCold* cw = at32(cold, slot);
        int fl = W.stack[slot][0];
        const i4 ctl = slot_load(&cw->ctl);
        if ((ctl.w & 7) != M_TRACE) continue;           // a slot that only waits for a work item (there is none)
        if (fl & kHasAux) {
          const int axp = f2i(slot_load(&cw->thr).w);
          const int nS = (fl >> kPendShift) & 3;
#pragma unroll
          for (int i = 0; i < kPacketShadows; i++) {
            if (i < nS) {
              const int ax = (axp >> (kSlotBits * i)) & kSlotMask;
              const int stt = fl_stat(W.stack[ax][0], 0);
              fl = (fl & ~(3 << (kStatShift + 2 * i))) | (stt << (kStatShift + 2 * i));
              if (stt == 2) slot_store(&cw->att[i], slot_load(&at32(cold, ax)->att[0]));
            }
          }
          fl &= ~(kHasAux | kJoinMask);
        }
        const v4 na = W.nodeA[slot], nb = W.nodeB[slot];
        slot_store(&cw->spare[0], na);
        slot_store(&cw->spare[1], mk4(nb.x, nb.y, nb.z, i2f(fl)));
        // the list has two ends: deep paths from the front, the others from the back (the drain kernel starts with the deep ones)
        const bool deep = ctl.y >= kDrainDeep;
        int* dl = a.workCounter + kDrainList;
        const int idx = atomicAdd(dl + (deep ? kDrainDeepN : kDrainOtherN), 1);
        dl[deep ? kDrainEntries + idx : kDrainEntries + dl[kDrainCap] - 1 - idx] = gpool * NS + slot;
        if (CNT) atomicAdd(a.counters + 812, 1ull);
      }
    }
  }

  if constexpr (CNT) {
    unsigned long long* c = a.counters;
    const uint32_t v[9] = { wave_sum(ct.samples), wave_sum(ct.primaryRays), wave_sum(ct.bounceRays), wave_sum(ct.shadowRays),
                            wave_sum(ct.nodeFetches), wave_sum(ct.triTests), wave_sum(ct.closestHits), wave_sum(ct.lightLoads),
                            wave_sum(ct.analyticTests) };
    const uint32_t rw[4] = { wave_sum(rows[0]), wave_sum(rows[1]), wave_sum(rows[2]), wave_sum(rows[3]) };
    for (int i = 0; i < kCensusRegions; i++) {
      const uint32_t cl = wave_sum(ct.censusLanes[i]), cw = wave_sum(ct.censusWaves[i]);
      if (lane == 0 && cw != 0u) { atomicAdd(&c[816 + i], (unsigned long long)cl); atomicAdd(&c[816 + kCensusRegions + i], (unsigned long long)cw); }
    }
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
      atomicAdd(&c[808], (unsigned long long)rw[0]); atomicAdd(&c[809], (unsigned long long)rw[1]);
      atomicAdd(&c[810], (unsigned long long)rw[2]); atomicAdd(&c[811], (unsigned long long)rw[3]);
      atomicAdd(&c[28], tBLoad); atomicAdd(&c[29], tBRun); atomicAdd(&c[30], tBStore); atomicAdd(&c[31], nTxn); atomicAdd(&c[32], nIter);
    }
  }
}

}  // namespace

// The instantiations are compiled in two translation units, by node format: this file holds the 64-byte-node kernels (the benchmark scene's), and
// packetkernel_n128.hip includes it with PT_PK_N128 for the 128-byte-node ones.  The reason is the compiler, not the code: LLVM's instruction
// scheduling strategy is a per-file flag (Makefile), "max-ilp" is 0.85 % faster than the default on the 64-byte kernels (coffee 312.5 -> 309.9 ms,
// glass knot -1.5 %) and 2-4 % SLOWER on the 128-byte ones (dining room) and on the queue kernels (random spheres) -- measured, round 6.
#ifndef PT_PK_N128
int packetkernel_lds_stack_entries() { return kStackN; }
int packetkernel_slots() { return kP * kWaves; }
size_t packetkernel_cold_bytes(int nBlocks) { return (size_t)nBlocks * kWaves * kP * sizeof(SlotCold); }
size_t packetkernel_overflow_ints(int nBlocks, int ovfDepth) { return (size_t)nBlocks * kWaves * kP * (size_t)ovfDepth; }
constexpr bool kThisFileN64 = true;
#define PT_PK_LAUNCH launch_packetkernel_n64
#else
constexpr bool kThisFileN64 = false;
#define PT_PK_LAUNCH launch_packetkernel_n128
#endif

template <bool CNT, bool FAST>
static void launch_pk(dim3 grid, dim3 block, hipStream_t stream, const LaunchArgs& a) {
  if (a.scene.shadowNearest) pt_packetkernel<CNT, true, FAST, true, kThisFileN64><<<grid, block, 0, stream>>>(a);
  else                       pt_packetkernel<CNT, true, FAST, false, kThisFileN64><<<grid, block, 0, stream>>>(a);
}
hipError_t PT_PK_LAUNCH(hipStream_t stream, const LaunchArgs& a, int nBlocks, bool counted, bool fastShading) {
  dim3 grid(nBlocks), block(kBlockThreads);
  if (fastShading) { if (counted) launch_pk<true, true>(grid, block, stream, a); else launch_pk<false, true>(grid, block, stream, a); }
  else             { if (counted) launch_pk<true, false>(grid, block, stream, a); else launch_pk<false, false>(grid, block, stream, a); }
  return hipGetLastError();
}
#ifndef PT_PK_N128
hipError_t launch_packetkernel_n128(hipStream_t stream, const LaunchArgs& a, int nBlocks, bool counted, bool fastShading);
hipError_t launch_packetkernel(hipStream_t stream, const LaunchArgs& a, int nBlocks, bool counted, bool fastShading) {
  // moptix_api.hip fill_view: the option node_format decides whether the scene view carries the 64-byte nodes
  return a.scene.nodes64 != nullptr ? launch_packetkernel_n64(stream, a, nBlocks, counted, fastShading) : launch_packetkernel_n128(stream, a, nBlocks, counted, fastShading);
}
#endif

}  // namespace pt
