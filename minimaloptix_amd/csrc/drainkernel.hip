// drainkernel.hip -- the last paths of a launch, a wave per (up to sixteen) path(s).
//
// The packet kernel (packetkernel.hip) is a throughput machine: a path's bounce is a chain of passes -- one shading visit, and per ray of
// the packet node runs and leaf passes that each wait for their turn at the wave -- and in an idle GPU that chain is 36 us per bounce
// (profiles/r06_tail_anatomy.txt: visit 9.6 us; ~30 node steps of 0.5 us, one dependent gather each; 4.5 leaf passes of 1.9 us; the scheduler
// between them).  The drain of a launch is a few paths walking that chain to the depth cap (Material.cu:29, MinimalOptiX.h:85: 256 bounces)
// while 99 % of the lanes idle.  So a workgroup of the packet kernel that is down to LaunchArgs::drainBelow paths hands each of them over
// at its next packet boundary (hand_over) and leaves, and this kernel finishes them with the machine turned the other way round:
//
//   * a wave owns up to kDP paths, one per lane (lanes 0 .. kDP-1): the shading visit is pt_packet.h's on_result_packet, the same code on
//     the same values as the packet kernel's visit, with the path state in registers -- no slot record, no queue, no transaction;
//   * the packet's rays -- up to three shadow rays and the continuation, of every path of the wave -- are traced TOGETHER by all 64 lanes:
//     a shared frontier in LDS holds (node or leaf, ray) entries, every round pops up to 64 of them, one per lane, tests the node's four
//     child boxes or the leaf's triangles, and pushes the entered children.  A ray's ~30 dependent node steps become ~10 rounds of the
//     tree's depth, whatever the number of rays.
//
// Same results, bit for bit: what a ray reports is defined without reference to the order of the traversal (rule D5) --
// the nearest accepted hit by (t, primitive id) for a radiance ray and for a shadow ray of a scene with glass, "some opaque surface in
// (eps, tmax)" for the others -- so the frontier keeps, per ray, a 64-bit key (t's bits above the primitive id) that candidates lower with
// an LDS atomic minimum: the minimum over the triangles of every leaf whose box the ray enters before its interval ends there is what the
// depth-first walk finds.  Entries are culled against the key as it stands when they are popped (conservatively, like node_step_nearfar).
// tri_test, hit attributes, materials, random draws: the shared functions of pt_geom.h / pt_path.h / pt_packet.h.
#include <hip/hip_runtime.h>

#include "megakernel.h"
#include "pt_path.h"
#include "pt_packet.h"
#include "pt_slot.h"

namespace pt {

namespace {

// Paths per wave (lanes 0 .. kDP-1 shade; all 64 lanes trace).  A lane takes the next path of the list when its own has ended, so a wave
// is full while the list lasts -- most of what a cold launch hands over ends within a few bounces, and sixteen of them share a visit's
// ~8 us -- and walks the stragglers alone or in twos at the end, when a bounce is one visit and ~10 rounds.
constexpr int kDP = 16;
constexpr int kDR = kDP * 4;            // rays in flight per wave: ray id = path * 4 + index in the packet (shadow rays first)
constexpr int kRidMask = kDR - 1;       // a frontier entry keeps its ray id in the lowest mantissa bits of its entry distance
constexpr int kCap = 1536;              // frontier entries
constexpr int kMargin = 160;            // room kept for a depth-first descent when the frontier is nearly full (3 pushes per level)
constexpr unsigned int kNoPrim = 0x7fffffffu;
// A wave that walks one or two paths -- the launch's very end, where every microsecond of a bounce is exposed 257 times -- pays two rounds for the root and its
// children with one to four lanes busy.  The root's grandchildren and their boxes (<= 16 entries, from the 128-byte nodes) sit in LDS instead, and while the wave
// holds at most kTop rays each ray starts there: (ray, entry) pairs tested in one round without a fetch.  Order and starting level change the work, not a result.
constexpr int kTop = 16;

struct DrainLds {
  int ref[kCap];                        // node index (>= 0) or leaf reference (< 0)
  float tn[kCap];                       // entry distance, the ray id in its six lowest mantissa bits (rounded down: conservative)
  float o[kDP][3];                      // origin of a path's packet (all its rays leave from one point)
  float d[kDR][3], inv[kDR][3], noi[kDR][3];
  float tmax[kDR];
  int shadow[kDR];
  unsigned long long best[kDR];         // (bits of t) << 32 | primitive id: nearest accepted hit so far; kNoPrim = none
  int bestTri[kDR], bestCls[kDR], bestMat[kDR];
  float beta[kDR], gamma[kDR];
  int dead[kDR];                        // shadow ray terminated by an opaque surface (scenes without glass)
  // the tree's second level (the root's children's children, <= kTop of them) with their boxes: where a ray starts while the wave holds few rays
  float topBox[kTop][6]; int topRef[kTop]; int nTop;
  int rayList[kTop];                    // the ids of the rays in flight, compact
};

__device__ __forceinline__ int lane_prefix(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}
__device__ __forceinline__ uint32_t wave_sum32(uint32_t v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float inv_dir(float d) {
  return __builtin_amdgcn_rcpf(__builtin_fabsf(d) < 1e-30f ? __builtin_copysignf(1e-30f, d) : d);
}
__device__ __forceinline__ unsigned long long hit_key(float t, unsigned int prim) { return ((unsigned long long)(uint32_t)f2i(t) << 32) | prim; }

// the four child boxes of a node against one ray: which are entered (pt_path.h trav_node_step's planes, without its sort and stack)
template <bool N64>
__device__ __forceinline__ void child_tests(const SceneView& sc, int node, v3 inv, v3 noi, float tmin, float tbest, int r[4], float tn[4], bool in[4]) {
  float nx[4], fx[4], ny[4], fy[4], nz[4], fz[4];
  if constexpr (N64) {
    const Node64 n = load_const(at32(sc.nodes64, node));
    const float sx = n.sx * inv.x, sy = n.sy * inv.y, sz = n.sz * inv.z;
    const float cx = fma_(n.ox, inv.x, noi.x), cy = fma_(n.oy, inv.y, noi.y), cz = fma_(n.oz, inv.z, noi.z);
    const bool bx = inv.x < 0.f, by = inv.y < 0.f, bz = inv.z < 0.f;
    planes4q(bx ? n.q[3] : n.q[0], sx, cx, nx); planes4q(bx ? n.q[0] : n.q[3], sx, cx, fx);
    planes4q(by ? n.q[4] : n.q[1], sy, cy, ny); planes4q(by ? n.q[1] : n.q[4], sy, cy, fy);
    planes4q(bz ? n.q[5] : n.q[2], sz, cz, nz); planes4q(bz ? n.q[2] : n.q[5], sz, cz, fz);
    r[0] = n.ref[0]; r[1] = n.ref[1]; r[2] = n.ref[2]; r[3] = n.ref[3];
  } else {
    const Node128 n = load_const(at32(sc.nodes, node));
    float a[4], b[4];
    planes4(n.lox, inv.x, noi.x, a); planes4(n.hix, inv.x, noi.x, b);
#pragma unroll
    for (int c = 0; c < 4; c++) { nx[c] = fminf_(a[c], b[c]); fx[c] = fmaxf_(a[c], b[c]); }
    planes4(n.loy, inv.y, noi.y, a); planes4(n.hiy, inv.y, noi.y, b);
#pragma unroll
    for (int c = 0; c < 4; c++) { ny[c] = fminf_(a[c], b[c]); fy[c] = fmaxf_(a[c], b[c]); }
    planes4(n.loz, inv.z, noi.z, a); planes4(n.hiz, inv.z, noi.z, b);
#pragma unroll
    for (int c = 0; c < 4; c++) { nz[c] = fminf_(a[c], b[c]); fz[c] = fmaxf_(a[c], b[c]); }
    r[0] = n.ref[0]; r[1] = n.ref[1]; r[2] = n.ref[2]; r[3] = n.ref[3];
  }
#pragma unroll
  for (int c = 0; c < 4; c++) {
    tn[c] = fmaxf_(fmaxf_(nx[c], ny[c]), fmaxf_(nz[c], tmin));
    const float tf = fminf_(fminf_(fx[c], fy[c]), fminf_(fz[c], tbest));
    in[c] = (tn[c] <= tf * 1.0000005f) && (c < 2 || r[c] != kEmptyRef);
  }
}

template <bool CNT, bool FAST, bool NEAR, bool N64>
__global__ void __launch_bounds__(64) pt_drainkernel(const LaunchArgs a) {
  __shared__ DrainLds S;
  const int lane = threadIdx.x;
  SceneView scv = a.scene;
  scv.shadowNearest = NEAR ? 1 : 0;
  const SceneView& sc = scv;
  int* dl = a.workCounter + kDrainList;
  const int nDeep = dl[kDrainDeepN], total = nDeep + dl[kDrainOtherN], cap = dl[kDrainCap];
  if (total == 0) return;
  const SlotCold* cold = reinterpret_cast<const SlotCold*>(a.poolCold);
  const int triBase = sc.nSpheres + sc.nQuads;

  if (lane == 0) {
    int n = 0;
    if (sc.rootRef >= 0 && sc.rootRef != kEmptyRef) {
      const Node128 r = load_const(at32(sc.nodes, sc.rootRef));
      auto put = [&](const Node128& nd, int c) {
        const float* f = reinterpret_cast<const float*>(&nd);      // lox loy loz hix hiy hiz, four children each
        for (int k = 0; k < 6; k++) S.topBox[n][k] = f[4 * k + c];
        S.topRef[n] = nd.ref[c]; n++;
      };
      for (int c1 = 0; c1 < r.count; c1++) {
        if (r.ref[c1] == kEmptyRef) continue;
        if (r.ref[c1] >= 0) {
          const Node128 m = load_const(at32(sc.nodes, r.ref[c1]));
          for (int c2 = 0; c2 < m.count; c2++) if (m.ref[c2] != kEmptyRef) put(m, c2);
        } else put(r, c1);                                          // a leaf right under the root
      }
    }
    S.nTop = n;
  }
  __syncthreads();

  Counters ct = {};
  uint32_t rounds = 0, roundLanes = 0;
  PathState ps; Packet pk; Trav res;
  v3 att[kPacketShadows] = { mk3(1.f, 1.f, 1.f), mk3(1.f, 1.f, 1.f), mk3(1.f, 1.f, 1.f) };
  packet_clear(pk);
  ps.mode = M_DONE; ps.kind = RK_RADIANCE; ps.item = 0; ps.depth = 0; ps.seed = 0; ps.pixel = 0; ps.light = 0; ps.mat = 0;
  ps.thr = mk3(1.f, 1.f, 1.f); ps.rad = mk3(0.f, 0.f, 0.f); ps.accum = mk3(0.f, 0.f, 0.f); ps.cdlin = mk3(0.f, 0.f, 0.f);
  ps.o = mk3(0.f, 0.f, 0.f); ps.d = mk3(0.f, 0.f, 1.f); ps.tmin = sc.epsT; ps.tmax = kRtDefaultMax;
  ps.N = mk3(0.f, 0.f, 1.f); ps.V = mk3(0.f, 0.f, 1.f); ps.pendW = mk3(0.f, 0.f, 0.f); ps.pendInv = 0.f;
  res.node = kTravDone; res.sp = 0; res.started = 1; res.tbest = kRtDefaultMax; res.bestPrim = -1; res.bestTri = -1; res.bestCls = SHADOW_NONE;
  res.beta = 0.f; res.gamma = 0.f; res.att = mk3(1.f, 1.f, 1.f); res.inv = mk3(0.f, 0.f, 0.f); res.noi = mk3(0.f, 0.f, 0.f);
  bool listEmpty = false;
  unsigned int guard = 0;
  const unsigned long long wdStart = __builtin_amdgcn_s_memrealtime();

  for (;;) {
    // ---- a lane without a path takes the next one from the list: the state run_batch (packetkernel.hip) would have loaded ----
    if (lane < kDP && ps.mode == M_DONE && !listEmpty) {
      const int k = atomicAdd(dl + kDrainNext, 1);
      if (k >= total) listEmpty = true;
      else {
        // deep paths first (they are the ones that may walk to the cap: the launch ends when the last of them does)
        const SlotCold* cs = cold + (k < nDeep ? dl[kDrainEntries + k] : dl[kDrainEntries + cap - 1 - (k - nDeep)]);
        const i4 ctl = slot_load(&cs->ctl);
        const v4 thrIn = slot_load(&cs->thr), radIn = slot_load(&cs->rad);
        const v4 na = slot_load(&cs->spare[0]), nb = slot_load(&cs->spare[1]);
        const int fl = f2i(nb.w);
        if (CNT) atomicAdd(a.counters + 813, 1ull);
        const int nSh = (fl >> kPendShift) & 3;
        v4 wh = mk4(0.f, 0.f, 0.f, 0.f), wb = mk4(1.f, 1.f, 1.f, 1.f);
        if (fl & kHitValid) wh = slot_load(&cs->hit);
        if (fl & kHasScale) wb = slot_load(&cs->bsc);
        packet_clear(pk);
#pragma unroll
        for (int i = 0; i < kPacketShadows; i++) {
          att[i] = mk3(1.f, 1.f, 1.f);
          if (i < nSh) {
            const v4 wp = slot_load(&cs->pend[i]);
            pk.pendW[i] = mk3(wp.x, wp.y, wp.z); pk.pendInv[i] = wp.w;
            const int stt = fl_stat(fl, i);
            if (stt == 2) { const v4 wa = slot_load(&cs->att[i]); att[i] = mk3(wa.x, wa.y, wa.z); }
            else if (stt == 1) att[i] = mk3(0.f, 0.f, 0.f);
          }
        }
        ps.item = ctl.x; ps.depth = ctl.y; ps.seed = (uint32_t)ctl.z; ps.pixel = 0; ps.light = 0;
        pk.nShadow = (ctl.w >> 3) & 3; pk.hasBounce = (ctl.w >> 5) & 1; pk.hasScale = (ctl.w >> 6) & 1;
        pk.bscale = mk3(wb.x, wb.y, wb.z); pk.binv = wb.w;
        ps.thr = mk3(thrIn.x, thrIn.y, thrIn.z); ps.rad = mk3(radIn.x, radIn.y, radIn.z);
        ps.N = mk3(0.f, 0.f, 1.f); ps.mat = 0; ps.V = mk3(0.f, 0.f, 1.f);
        ps.pendW = mk3(0.f, 0.f, 0.f); ps.pendInv = 0.f; ps.accum = mk3(0.f, 0.f, 0.f); ps.cdlin = mk3(0.f, 0.f, 0.f);
        ps.o = mk3(na.x, na.y, na.z); ps.d = mk3(nb.x, nb.y, nb.z); ps.tmin = sc.epsT; ps.tmax = kRtDefaultMax; ps.kind = RK_RADIANCE;
        res.tbest = na.w; res.bestTri = -1; res.bestPrim = -1; res.beta = 0.f; res.gamma = 0.f; res.att = mk3(1.f, 1.f, 1.f);
        if (fl & kHitValid) { res.bestTri = f2i(wh.x); res.bestPrim = f2i(wh.y); res.beta = wh.z; res.gamma = wh.w; }
        ps.mode = M_RESULT;
      }
    }
    // ---- the visit: fold the shadow results, shade the continuation's hit, leave the next packet (pt_packet.h) ----
    if (lane < kDP && ps.mode == M_RESULT) {
      const PacketSink sink{ pk };
      on_result_packet<CNT, FAST, PacketSink>(sc, ps, pk, res, att, ct, sink);
      if (ps.mode == M_NEW_SAMPLE) {
        if (CNT) {      // finish-time histogram of the counting build, on the packet kernel's clock (its first wave's start)
          const unsigned long long b = min(255ull, (__builtin_amdgcn_s_memrealtime() - a.counters[36]) / 100000ull);
          atomicAdd(a.counters + 40 + b, 1ull); atomicMax(a.counters + 296 + b, (unsigned long long)ps.depth); atomicAdd(a.counters + 552 + b, (unsigned long long)ps.depth);
        }
        store_sample(a, ps.item, ps.accum);
        if (CNT) atomicAdd(a.counters + 814, 1ull);
        if (a.tileCost != nullptr && ps.depth >= kDeepPath) atomicMax(a.tileCost + ((ps.item % a.nItems) >> a.unitShift), (unsigned int)ps.depth);
        ps.mode = M_DONE;
      }
    }
    const bool tracing = lane < kDP && ps.mode == M_TRACE;
    if (__ballot(tracing) == 0ull) {
      if (__ballot(lane < kDP && !listEmpty) == 0ull) break;      // no path in the wave and none left to take
      continue;
    }
    if ((++guard & 255u) == 0u && __builtin_amdgcn_s_memrealtime() - wdStart > a.watchdogTicks) { if (lane == 0) atomicOr(a.workCounter + 1, 1); break; }

    // ---- the packet's rays into LDS; the brute-force lists for the continuation (radiance) ray ----
    int nR = 0;
    if (tracing) {
      nR = pk.nShadow + pk.hasBounce;
      S.o[lane][0] = ps.o.x; S.o[lane][1] = ps.o.y; S.o[lane][2] = ps.o.z;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (j < nR) {
          const bool sh = j < pk.nShadow;
          v3 d = ps.d; float tmax = kRtDefaultMax;
#pragma unroll
          for (int i = 0; i < kPacketShadows; i++) if (sh && i == j) { d = pk.sd[i]; tmax = pk.stmax[i]; }
          unsigned int prim0 = kNoPrim;
          if (!sh) {
            Trav tv;
            ps.kind = RK_RADIANCE;
            trav_begin<CNT, false>(sc, ps, tv, ct);
            tmax = tv.tbest; if (tv.bestPrim >= 0) prim0 = (unsigned int)tv.bestPrim;
          }
          const int rid = lane * 4 + j;
          const v3 inv = mk3(inv_dir(d.x), inv_dir(d.y), inv_dir(d.z));
          const v3 noi = neg_o_inv(ps.o, inv);
          S.d[rid][0] = d.x; S.d[rid][1] = d.y; S.d[rid][2] = d.z;
          S.inv[rid][0] = inv.x; S.inv[rid][1] = inv.y; S.inv[rid][2] = inv.z;
          S.noi[rid][0] = noi.x; S.noi[rid][1] = noi.y; S.noi[rid][2] = noi.z;
          S.tmax[rid] = tmax; S.shadow[rid] = sh ? 1 : 0;
          S.best[rid] = hit_key(tmax, prim0);
          S.bestTri[rid] = -1; S.bestCls[rid] = SHADOW_NONE; S.bestMat[rid] = 0; S.beta[rid] = 0.f; S.gamma[rid] = 0.f; S.dead[rid] = 0;
        }
      }
    }
    int sp = 0;
    if (sc.rootRef != kEmptyRef) {
      const unsigned long long m0 = __ballot(nR > 0), m1 = __ballot(nR > 1), m2 = __ballot(nR > 2), m3 = __ballot(nR > 3);
      const int off = lane_prefix(m0) + lane_prefix(m1) + lane_prefix(m2) + lane_prefix(m3);
      const int nRays = __popcll(m0) + __popcll(m1) + __popcll(m2) + __popcll(m3);
      const int nT = S.nTop;
      if (nT > 0 && nRays <= kTop) {         // few rays: each starts at the tree's second level (kTop above)
        for (int j = 0; j < nR; j++) S.rayList[off + j] = lane * 4 + j;
        __syncthreads();
        const int pairs = nRays * nT;
        for (int base = 0; base < pairs; base += 64) {
          const int idx = base + lane;
          bool in = false; int rid = 0, e = 0; float tn = 0.f;
          if (idx < pairs) {
            const int r = idx / nT; e = idx - r * nT; rid = S.rayList[r];
            const float tb = i2f((int32_t)(S.best[rid] >> 32));
            float tf = tb; tn = sc.epsT;
#pragma unroll
            for (int ax = 0; ax < 3; ax++) {
              const float a = fma_(S.topBox[e][ax], S.inv[rid][ax], S.noi[rid][ax]), b = fma_(S.topBox[e][3 + ax], S.inv[rid][ax], S.noi[rid][ax]);
              tn = fmaxf_(tn, fminf_(a, b)); tf = fminf_(tf, fmaxf_(a, b));
            }
            in = tn <= tf * 1.0000005f;
          }
          const unsigned long long m = __ballot(in);
          if (in) { const int w = sp + lane_prefix(m); S.ref[w] = S.topRef[e]; S.tn[w] = i2f((f2i(tn) & ~kRidMask) | rid); }
          sp += __popcll(m);
        }
      } else {                               // one frontier entry per ray, at the root
        for (int j = 0; j < nR; j++) { S.ref[off + j] = sc.rootRef; S.tn[off + j] = i2f((f2i(sc.epsT) & ~kRidMask) | (lane * 4 + j)); }
        sp = nRays;
      }
    }
    __syncthreads();

    // ---- all 64 lanes: pop up to 64 entries, test, push the entered children ----
    while (sp > 0) {
      int n = min(64, sp);
      n = min(n, max(1, (kCap - kMargin - sp) / 3));
      if (sp + 3 > kCap) { if (lane == 0) atomicOr(a.workCounter + 1, 1); sp = 0; break; }      // cannot happen with a tree of depth < kMargin / 3
      const bool have = lane < n;
      int ref = 0, rid = 0; float tnf = 0.f;
      if (have) { const int idx = sp - 1 - lane; ref = S.ref[idx]; tnf = S.tn[idx]; rid = f2i(tnf) & kRidMask; }
      sp -= n;
      if (CNT) { rounds++; roundLanes += (uint32_t)n; }
      bool live = have;
      float tb = 0.f; v3 o = mk3(0.f, 0.f, 0.f), d = mk3(0.f, 0.f, 1.f);
      int isShadow = 0;
      if (have) {
        tb = i2f((int32_t)(S.best[rid] >> 32));
        isShadow = S.shadow[rid];
        live = i2f(f2i(tnf) & ~kRidMask) <= tb * 1.0000005f && !(isShadow && !NEAR && S.dead[rid]);
        const int p = rid >> 2;
        o = mk3(S.o[p][0], S.o[p][1], S.o[p][2]);
      }
      int cr[4] = { 0, 0, 0, 0 }; float ctn[4] = { 0.f, 0.f, 0.f, 0.f }; bool cin[4] = { false, false, false, false };
      unsigned long long myKey = ~0ull; int myTri = -1, myCls = SHADOW_NONE, myMat = 0; float myBe = 0.f, myGa = 0.f;
      if (live && ref >= 0) {
        cnt<CNT>(ct.nodeFetches);
        const v3 inv = mk3(S.inv[rid][0], S.inv[rid][1], S.inv[rid][2]), noi = mk3(S.noi[rid][0], S.noi[rid][1], S.noi[rid][2]);
        child_tests<N64>(sc, ref, inv, noi, sc.epsT, tb, cr, ctn, cin);
      } else if (live) {
        d = mk3(S.d[rid][0], S.d[rid][1], S.d[rid][2]);
        const float tmaxTest = (isShadow && !NEAR) ? S.tmax[rid] : kRtDefaultMax;      // as trav_leaf_step_fetched: potential() range-tests the others
        const float tmax0 = S.tmax[rid];
        const int first = leaf_first(ref), count = leaf_count(ref);
        bool term = false;
        for (int base = 0; base < count && !term; base += 4) {
          LeafChunk ch;
          leaf_fetch4(sc, ref, base, ch);
#pragma unroll
          for (int j = 0; j < 4; j++) {
            if (base + j < count && !term) {
              cnt<CNT>(ct.triTests);
              v3 nn; float t, be, ga;
              if (tri_test(o, d, sc.epsT, tmaxTest, ch.p0[j], ch.e0[j], ch.e1[j], nn, t, be, ga)) {
                if (isShadow && !NEAR) {
                  if (ch.shadow[j] == SHADOW_OPAQUE) { S.dead[rid] = 1; term = true; }      // disneyAnyHit on an opaque surface: rtTerminateRay
                } else if (!isShadow || (ch.shadow[j] != SHADOW_NONE && t < tmax0)) {
                  // radiance: every accepted triangle is a candidate (the key's minimum is potential()'s chain, ties included); a shadow ray that
                  // keeps its nearest candidate: surfaces with the any-hit program inside the ray's own interval (shadow_candidate_tri)
                  const unsigned long long key = hit_key(t, (unsigned int)(triBase + ch.prim[j]));
                  if (key < myKey) { myKey = key; myTri = first + base + j; myCls = ch.shadow[j]; myMat = ch.mat[j]; myBe = be; myGa = ga; }
                }
              }
            }
          }
        }
        if (myKey != ~0ull) atomicMin(&S.best[rid], myKey);
      }
      // children entered -> frontier
      const bool isNode = live && ref >= 0;
      const int m = isNode ? (int)cin[0] + (int)cin[1] + (int)cin[2] + (int)cin[3] : 0;
      const unsigned long long b0 = __ballot((m & 1) != 0), b1 = __ballot((m & 2) != 0), b2 = __ballot((m & 4) != 0);
      int w = sp + lane_prefix(b0) + 2 * lane_prefix(b1) + 4 * lane_prefix(b2);
      if (isNode) {
#pragma unroll
        for (int c = 3; c >= 0; c--) {        // children 3 .. 0: the lower ones end up nearer the top
          if (cin[c]) { S.ref[w] = cr[c]; S.tn[w] = i2f((f2i(ctn[c]) & ~kRidMask) | rid); w++; }
        }
      }
      sp += __popcll(b0) + 2 * __popcll(b1) + 4 * __popcll(b2);
      __syncthreads();
      // the lane that holds a ray's new nearest hit leaves its attributes
      if (myKey != ~0ull && S.best[rid] == myKey) { S.bestTri[rid] = myTri; S.bestCls[rid] = myCls; S.bestMat[rid] = myMat; S.beta[rid] = myBe; S.gamma[rid] = myGa; }
    }
    __syncthreads();

    // ---- results back to the paths' lanes ----
    if (tracing) {
#pragma unroll
      for (int i = 0; i < kPacketShadows; i++) {
        att[i] = mk3(1.f, 1.f, 1.f);
        if (i < pk.nShadow) {
          const int rid = lane * 4 + i;
          if (NEAR) {
            const unsigned int prim = (unsigned int)(S.best[rid] & 0xffffffffull);
            if (prim != kNoPrim) att[i] = (S.bestCls[rid] == SHADOW_GLASS) ? load_const(&at32(sc.mats, S.bestMat[rid])->color) : mk3(0.f, 0.f, 0.f);
          } else if (S.dead[rid]) att[i] = mk3(0.f, 0.f, 0.f);
        }
      }
      if (pk.hasBounce) {
        const int rid = lane * 4 + pk.nShadow;
        const unsigned long long key = S.best[rid];
        const unsigned int prim = (unsigned int)(key & 0xffffffffull);
        res.tbest = i2f((int32_t)(key >> 32));
        res.bestPrim = prim == kNoPrim ? -1 : (int)prim;
        res.bestTri = S.bestTri[rid]; res.bestCls = S.bestCls[rid]; res.beta = S.beta[rid]; res.gamma = S.gamma[rid];
      } else { res.tbest = kRtDefaultMax; res.bestPrim = -1; res.bestTri = -1; }
      ps.mode = M_RESULT;
    }
    __syncthreads();
  }

  if constexpr (CNT) {
    unsigned long long* c = a.counters;
    const uint32_t v[9] = { wave_sum32(ct.samples), wave_sum32(ct.primaryRays), wave_sum32(ct.bounceRays), wave_sum32(ct.shadowRays),
                            wave_sum32(ct.nodeFetches), wave_sum32(ct.triTests), wave_sum32(ct.closestHits), wave_sum32(ct.lightLoads),
                            wave_sum32(ct.analyticTests) };
    for (int i = 0; i < kCensusRegions; i++) {
      const uint32_t cl = wave_sum32(ct.censusLanes[i]), cw = wave_sum32(ct.censusWaves[i]);
      if (lane == 0 && cw != 0u) { atomicAdd(&c[816 + i], (unsigned long long)cl); atomicAdd(&c[816 + kCensusRegions + i], (unsigned long long)cw); }
    }
    if (lane == 0) {
      for (int i = 0; i < 9; i++) atomicAdd(&c[i], (unsigned long long)v[i]);
      atomicAdd(&c[9], (unsigned long long)rounds); atomicAdd(&c[10], (unsigned long long)roundLanes);
      atomicMax(&c[38], (unsigned long long)__builtin_amdgcn_s_memrealtime());   // last wave out
    }
  }
}

}  // namespace

size_t drain_list_ints(int nBlocks, int drainBelow) { return kDrainEntries + (size_t)nBlocks * (size_t)(drainBelow > 0 ? drainBelow : 0); }

template <bool CNT, bool FAST>
static void launch_dk(dim3 grid, hipStream_t stream, const LaunchArgs& a) {
  const bool n64 = a.scene.nodes64 != nullptr;
  if (a.scene.shadowNearest) { if (n64) pt_drainkernel<CNT, FAST, true, true><<<grid, dim3(64), 0, stream>>>(a); else pt_drainkernel<CNT, FAST, true, false><<<grid, dim3(64), 0, stream>>>(a); }
  else                       { if (n64) pt_drainkernel<CNT, FAST, false, true><<<grid, dim3(64), 0, stream>>>(a); else pt_drainkernel<CNT, FAST, false, false><<<grid, dim3(64), 0, stream>>>(a); }
}
// one wave per workgroup, eight per CU: more waves than the drain ever has paths / kDP for long
hipError_t launch_drainkernel(hipStream_t stream, const LaunchArgs& a, int nCUs, bool counted, bool fastShading) {
  dim3 grid(nCUs * 8);
  if (fastShading) { if (counted) launch_dk<true, true>(grid, stream, a); else launch_dk<false, true>(grid, stream, a); }
  else             { if (counted) launch_dk<true, false>(grid, stream, a); else launch_dk<false, false>(grid, stream, a); }
  return hipGetLastError();
}

}  // namespace pt
