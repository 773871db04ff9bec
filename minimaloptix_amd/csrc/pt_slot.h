// pt_slot.h -- the path-slot record of kernel variant 4 in HBM (SlotCold) and the bits of a slot's LDS flag word: shared by
// packetkernel.hip, which owns the slots, and drainkernel.hip, which takes the last paths of a launch over at a packet boundary.
#pragma once
#include "pt_types.h"

namespace pt {
namespace {

constexpr int kSlotBits = 10;           // slot ids in packed words (borrowed slots, parent of a borrowed slot)
constexpr int kSlotMask = (1 << kSlotBits) - 1;



// Path-slot record in HBM, private to the pool: two 128-byte lines of 16-byte rows.
//   line 0, the shading visit's rows: ctl thr rad hit bsc pend[3]
//   line 1, the traversal's rows:     ray[3] (the packet's rays after the first, in trace order) att[3] (shadow
//                                     attenuations, only once a glass surface was crossed)
// A shading visit reads line 0 (the pend rows only for the packet's shadow rays) and writes it back; a ray switch reads
// one ray row; the leaf pass reads / writes hit (continuation) or att (shadow ray, tinted only).
struct alignas(16) i4 { int x, y, z, w; };
struct alignas(128) SlotCold {
  i4 ctl;       // item, depth, seed, mode | nShadow << 3 | hasBounce << 5 | hasScale << 6 | holds borrowed slots << 7
  v4 thr;       // throughput (of the hit the packet left from: the shadow results are folded with it); .w = the three borrowed slots (int bits)
  v4 rad;       // radiance so far
  v4 hit;       // continuation ray: bestTri, bestPrim (int bits), beta, gamma
  v4 bsc;       // weight of the continuation still to be applied at the next visit: brdf, 1/pdf
  v4 pend[3];   // shadow ray i: pendW, pendInv
  v4 ray[3];    // rays still to trace: d, t (shadow: tmax; continuation: tbest after the brute-force lists)
  v4 att[3];    // shadow ray i: attenuation when tinted
  v4 spare[2];  // (the texture colour of a hit never outlives its visit here)
};
static_assert(sizeof(SlotCold) == 256, "SlotCold layout");

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

// stack[slot][0]: bits 0-7 stack pointer, the rest describes the packet in flight
constexpr int kSpMask = 0xff;
// The continuation's nearest hit is a triangle with a Disney material that is not glass (Tri48::shadow == SHADOW_OPAQUE; kept up to
// date by the leaf pass that writes a nearer hit; the brute-force lists at a ray's start leave it clear): the finished packet goes
// to Q_SHADE, whose batches run the Disney program -- ~3,000 vector instructions.  Every other finished packet (continuation missed,
// hit a light quad or another analytic primitive, hit glass / a non-Disney mesh material, or there was no continuation) goes to
// Q_GEN: its visit folds the shadow results, runs miss / the cheap material program and, where the sample ended, takes the next
// work item in the same visit.  Until round 4 every packet with shadow rays went to Q_SHADE: its batches then ran the Disney code
// with 34 of 64 lanes on the benchmark scene (12 of 64 on the glass knot: profiles/r04_lane_census.txt), and a sample that ended
// on a miss needed a second visit for its new work item.
constexpr int kShadeFlag = 1 << 30;
constexpr int kShadowRay = 1 << 29;   // the ray in flight is a shadow ray (MinimalOptiX.h:48 RAY_TYPE_SHADOW)
constexpr int kHitValid = 1 << 28;    // SlotCold::hit holds the continuation's nearest hit so far
constexpr int kCurShift = 8;          // bits 8-9: index of the ray in flight within the packet (shadow rays first)
constexpr int kNShShift = 10;         // bits 10-11: shadow rays in the packet
constexpr int kNRayShift = 12;        // bits 12-13: rays in the packet - 1
constexpr int kStatShift = 14;        // bits 14-19: per shadow ray 0 = attenuation (1,1,1), 1 = (0,0,0), 2 = tinted (att row)
constexpr int kHasAux = 1 << 21;      // path slot: the packet's shadow rays are being traced by borrowed slots (join before shading)
constexpr int kAuxSlot = 1 << 22;     // borrowed slot: one shadow ray of the path slot named in bits 16-20 and 23-27 (aux_parent)
constexpr int kPendShift = 23;        // bits 23-24: pend rows the next visit has to read (= the packet's shadow rays, wherever they are traced)
// Join of a path slot with its borrowed slots, in the path slot's flag word: only these bits are touched by other
// waves (LDS atomics), so the owner updates the rest of the word with atomics too while kHasAux is set (store_flags).
constexpr int kJoinShift = 25;        // bits 25-26: borrowed slots whose shadow ray is still out
constexpr int kArrived = 1 << 27;     // the path slot's own ray is done (or it had none)
constexpr int kJoinMask = (3 << kJoinShift) | kArrived;
__device__ __forceinline__ int aux_parent_bits(int parent) { return ((parent & 31) << 16) | ((parent >> 5) << 23); }      // a borrowed slot uses neither its pend / join bits nor kArrived
__device__ __forceinline__ int aux_parent(int fl) { return ((fl >> 16) & 31) | (((fl >> 23) & 31) << 5); }
// SlotCold::ctl.w bit 7: the path holds three borrowed slots (10 bits each in thr.w), kept until the path ends
constexpr int kCtlHoldsAux = 1 << 7;
constexpr int kHasScale = 1 << 20;    // SlotCold::bsc holds the continuation's weight (a Disney bounce)
constexpr int kPrimUnknown = 0x7fffffff;      // leaf pass: the slot holds a hit whose primitive id has not been read (above every real id)
constexpr int kSwitchRef = (int)0x80000000;   // "node" of a slot whose ray ended while another ray of the packet is pending
__device__ __forceinline__ int fl_cur(int fl) { return (fl >> kCurShift) & 3; }
__device__ __forceinline__ int fl_nsh(int fl) { return (fl >> kNShShift) & 3; }
__device__ __forceinline__ int fl_nray(int fl) { return ((fl >> kNRayShift) & 3) + 1; }
__device__ __forceinline__ int fl_stat(int fl, int i) { return (fl >> (kStatShift + 2 * i)) & 3; }
__device__ __forceinline__ bool fl_more(int fl) { return fl_cur(fl) + 1 < fl_nray(fl); }     // another ray of the packet is pending


}  // namespace
}  // namespace pt
