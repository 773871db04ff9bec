// pt_path.h -- the per-lane path state machine and the resumable BVH traversal that the
// megakernel runs.  The reference's recursion
//     camera -> rtTrace -> closest-hit -> rtTrace -> ...      
// is flattened into   L = sum_k (prod_{j<k} w_j) * e_k   with the RNG draw order and the
// seed forks of SURVEY A2 preserved, so a lane always owns exactly one ray in flight:
//
//   M_NEW_PIXEL (fetch a (pixel,sample) work item) -> [M_TRACE -> M_RESULT (-> M_LIGHTS -> M_TRACE ...)]*
//       -> M_NEW_SAMPLE (sample finished, its clamped value is in ps.accum) -> M_NEW_PIXEL
//
// Traversal state (Trav) lives in registers + an LDS stack and is resumable at any step,
// which is what lets the kernel leave the traversal loop when too few lanes of the wave
// are still traversing (wave-level re-fill) and come back later.
#pragma once
#include "pt_types.h"
#include "pt_rng.h"
#include "pt_geom.h"
#include "pt_disney.h"
#include "pt_texture.h"

namespace pt {

enum { M_NEW_PIXEL = 0, M_NEW_SAMPLE = 1, M_TRACE = 2, M_RESULT = 3, M_LIGHTS = 4, M_DONE = 5 };
enum { RK_RADIANCE = 0, RK_SHADOW = 1 };   // MinimalOptiX.h:48 RayType

struct Counters {
  uint32_t samples, primaryRays, bounceRays, shadowRays;
  uint32_t nodeFetches, triTests, closestHits, lightLoads, analyticTests;
  // counting build only: lanes / waves that entered each divergent region of the passes (census<> below; the table is
  // printed under MOPTIX_DEBUG and committed as profiles/r04_lane_census.txt)
  uint32_t censusLanes[kCensusRegions], censusWaves[kCensusRegions];
};
template <bool CNT> PT_HD void cnt(uint32_t& c, uint32_t n = 1) { if (CNT) c += n; }
enum { CR_RESULT = 0, CR_MISS, CR_HIT, CR_LIGHT, CR_DEPTHCAP, CR_LAMBERT, CR_METAL, CR_GLASS, CR_DISNEY_GLASS, CR_DISNEY, CR_LIGHT0, CR_LIGHT1,
       CR_LIGHT2, CR_BOUNCE_EVAL, CR_GEN, CR_TRI0, CR_TRI1, CR_TRI2, CR_TRI3, CR_TRI_HIT, CR_SHADOW_FOLD, CR_LIGHT_DRAW, CR_BOUNCE_SAMPLE, CR_NODE_BRANCHED };
// one call at the top of a region: every lane that is active here counts itself, the lowest of them counts the wave
template <bool CNT> PT_HD void census(Counters& ct, int region) {
#if defined(__HIP_DEVICE_COMPILE__)
  if constexpr (CNT) {
    const unsigned long long m = __ballot(1);
    ct.censusLanes[region]++;
    if (__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)) == 0) ct.censusWaves[region]++;
  }
#endif
}

struct Trav {
  int node;        // current node/leaf reference; kTravDone when the ray is finished
  int sp;          // stack entries in use
  int started;     // analytic lists done, BVH walk in progress
  float tbest;     // radiance: nearest accepted t so far; shadow: the ray's tmax
  int bestPrim;    // primitive id of the nearest hit (spheres, quads, triangles) or -1
  int bestTri;     // record index of the nearest triangle, or -1
  int bestCls;     // Tri48::shadow of that triangle (SHADOW_OPAQUE = a Disney surface that is not glass): what program its hit will run
  float beta, gamma;
  v3 att;          // shadow attenuation (disneyAnyHit)
  v3 inv;          // 1/d
  v3 noi;          // -(o * 1/d): the node step evaluates a slab plane as fma(plane, inv, noi)
};

struct PathState {
  int mode;
  int pixel;           // y*W + x
  int item;            // work-item index = slot of this sample in the per-sample buffer
  v3 accum;            // the finished sample's clamped colour (valid in M_NEW_SAMPLE)
  v3 thr, rad;         // path throughput and accumulated radiance of the current sample
  int depth; uint32_t seed;
  v3 o, d; float tmin, tmax; int kind;    // the ray in flight
  v3 N, V; int mat; int light;            // Disney hit context while its lights are looped
  v3 pendW; float pendInv;                // weight of the shadow ray in flight
  v3 cdlin;                               // srgb2lin(texture colour) of the hit (textured Disney materials only)
};

// ---------------------------------------------------------------------------------------
// Traversal
// ---------------------------------------------------------------------------------------

// Shadow rays and disneyAnyHit (Material.cu:225-232), rule D5 of DESIGN.md 2.  In OptiX an any-hit program that neither ignores
// the intersection nor terminates the ray ACCEPTS it: the ray's interval shrinks to that hit and nothing farther along the ray is
// looked at any more.  disneyAnyHit terminates on an opaque surface (attenuation 0) and does neither on a GLASS surface
// (attenuation *= color), so with a front-to-back traversal a shadow ray is decided by the NEAREST surface that has the program:
// opaque -> (0,0,0); glass -> its colour, and whatever lies behind it -- opaque or not -- never gets to block the ray.  That is
// the deterministic definition used here (nearest by (t, primitive id)); round 3 found it in the reference's demo/coffee.png,
// where the floor round the machine is lit through the glass pot.  Primitives without the program (lights,
// non-Disney materials) are not there for a shadow ray.
//   * scene without a Disney GLASS material (sc.shadowNearest == 0): "some opaque surface on the segment" is the same answer and
//     the ray stops at the first one found -- shadow_any_hit, returns true when the ray is terminated;
//   * otherwise the traversal keeps the nearest candidate like a radiance ray does (tbest shrinks, bestPrim for ties) and
//     tv.att is the verdict of the nearest one so far -- shadow_candidate.
PT_HD bool shadow_any_hit(const SceneView& sc, int mat, v3& att) {
  const DevMaterial m = load_const(at32(sc.mats, mat));
  if (m.kind != MAT_DISNEY) return false;                 // no any-hit program: does not occlude
  if (m.brdfType == BRDF_GLASS) { att = att * m.color; return false; }
  att = mk3(0.f, 0.f, 0.f);
  return true;                                            // rtTerminateRay
}
PT_HD void shadow_candidate(const SceneView& sc, int mat, float t, int prim, float tmin, float& tbest, int& bestPrim, v3& att) {
  if (!potential(t, prim, tmin, tbest, bestPrim)) return;      // most candidates are not nearer: no material fetch for them
  const DevMaterial m = load_const(at32(sc.mats, mat));
  if (m.kind != MAT_DISNEY) return;
  tbest = t; bestPrim = prim;
  att = (m.brdfType == BRDF_GLASS) ? m.color : mk3(0.f, 0.f, 0.f);
}

// The same two for a triangle, whose record says what it is to a shadow ray (Tri48::shadow): only a glass surface's colour is
// still read from the material table.
PT_HD bool shadow_any_hit_tri(const SceneView& sc, int cls, int mat, v3& att) {
  if (cls == SHADOW_NONE) return false;
  if (cls == SHADOW_GLASS) { att = att * load_const(&at32(sc.mats, mat)->color); return false; }
  att = mk3(0.f, 0.f, 0.f);
  return true;
}
PT_HD void shadow_candidate_tri(const SceneView& sc, int cls, int mat, float t, int prim, float tmin, float& tbest, int& bestPrim, v3& att) {
  if (cls == SHADOW_NONE || !potential(t, prim, tmin, tbest, bestPrim)) return;
  tbest = t; bestPrim = prim;
  att = (cls == SHADOW_GLASS) ? load_const(&at32(sc.mats, mat)->color) : mk3(0.f, 0.f, 0.f);
}

// 1/d for the slab planes.  A direction component below 1e-30 in magnitude is treated as +-1e-30 so that
// plane*inv and o*inv stay finite (inf - inf would poison the planes); over any t the scene allows, that moves
// the ray by less than 1e-29, far inside the padding of the boxes.
PT_HD float slab_inv(float d) { return 1.0f / (__builtin_fabsf(d) < 1e-30f ? __builtin_copysignf(1e-30f, d) : d); }
PT_HD v3 neg_o_inv(v3 o, v3 inv) { return mk3(-(o.x * inv.x), -(o.y * inv.y), -(o.z * inv.z)); }

// Brute-force lists ("NoAccel" groups and the light geometry) + set-up of the BVH walk.
// WINDOW: the chunked sphere loop looks at a sphere's roots with the hardware square root first (below).  The packet kernel (variant
// 4) passes false: its scenes are triangle meshes with a handful of analytic primitives at most, the chunk never runs there, and
// with the window test compiled in its register allocation came out with 69 spilled vector registers instead of 5 (+7 % on the
// benchmark frame); without the chunk altogether with 37.  The allocation of that kernel is that fragile (NOTEBOOK.md).
template <bool CNT, bool WINDOW = true>
PT_HD void trav_begin(const SceneView& sc, const PathState& ps, Trav& tv, Counters& ct) {
  tv.tbest = ps.tmax; tv.bestPrim = -1; tv.bestTri = -1; tv.beta = 0.f; tv.gamma = 0.f;
  tv.att = mk3(1.f, 1.f, 1.f);
  tv.inv = mk3(slab_inv(ps.d.x), slab_inv(ps.d.y), slab_inv(ps.d.z));
  tv.noi = neg_o_inv(ps.o, tv.inv);
  tv.sp = 0; tv.started = 1;
  bool terminated = false;
  if (ps.kind == RK_RADIANCE) {
    int i0 = 0;
#ifndef PT_SPHERE_CHUNK
#define PT_SPHERE_CHUNK 4
#endif
#if defined(__HIP_DEVICE_COMPILE__)
    // Four spheres per trip: their discriminants are four independent chains (one sphere alone is a chain of ~16 dependent
    // instructions behind a scalar load), the roots are taken only where a discriminant is not negative.  Same operations
    // per sphere as sphere_roots, and the equal-t rule of potential() makes the outcome independent of the order anyway.
    for (; i0 + PT_SPHERE_CHUNK <= sc.nSpheres; i0 += PT_SPHERE_CHUNK) {
      float bq[PT_SPHERE_CHUNK], dq[PT_SPHERE_CHUNK];
#pragma unroll
      for (int k = 0; k < PT_SPHERE_CHUNK; k++) {
        const DevSphere s = load_uniform(sc.spheres + i0 + k);
        const v3 oc = ps.o - s.center;
        bq[k] = dot(ps.d, oc);
        const float c = dot(oc, oc) - s.radius * s.radius;
        dq[k] = bq[k] * bq[k] - c;
      }
#pragma unroll
      for (int k = 0; k < PT_SPHERE_CHUNK; k++) {
        // The correctly rounded roots (~35 instructions) only for spheres whose roots can fall into the window (tmin, tbest]: a first
        // look with the hardware square root (1 ulp) and a margin of 1e-4 of the magnitudes involved, a thousand times its error, so a
        // sphere the exact code would accept always passes (the exact code then decides, with the same operations as before).  A
        // wave runs the exact code when ANY of its 64 rays' lines meets the sphere -- most spheres, before this test.
        bool look = !(dq[k] < 0);
        if constexpr (WINDOW) {
          const float sa = __builtin_amdgcn_sqrtf(fmaxf_(dq[k], 0.f));
          const float margin = 1e-4f * (1.f + __builtin_fabsf(bq[k]) + sa);
          look = look && (sa - bq[k]) > ps.tmin - margin && (-bq[k] - sa) < tv.tbest + margin;
        }
        if (look) {
          const float sq = __builtin_sqrtf(dq[k]);
          const float t1 = -bq[k] - sq, t2 = -bq[k] + sq;
          if (potential(t1, i0 + k, ps.tmin, tv.tbest, tv.bestPrim)) { tv.tbest = t1; tv.bestPrim = i0 + k; }
          else if (potential(t2, i0 + k, ps.tmin, tv.tbest, tv.bestPrim)) { tv.tbest = t2; tv.bestPrim = i0 + k; }
        }
      }
    }
#endif
    for (int i = i0; i < sc.nSpheres; i++) {              // sphereIntersect, Geometry.cu:18-55
      const DevSphere s = sc.spheres[i];
      float t1, t2;
      if (sphere_roots(s.center, s.radius, ps.o, ps.d, t1, t2)) {
        if (potential(t1, i, ps.tmin, tv.tbest, tv.bestPrim)) { tv.tbest = t1; tv.bestPrim = i; }
        else if (potential(t2, i, ps.tmin, tv.tbest, tv.bestPrim)) { tv.tbest = t2; tv.bestPrim = i; }
      }
    }
    for (int i = 0; i < sc.nQuads; i++) {                 // quadIntersect, Geometry.cu:70-91
      const DevQuad q = load_uniform(sc.quads + i);
      float t; const int id = sc.nSpheres + i;
      if (quad_test(q.plane, q.v1, q.v2, q.anchor, ps.o, ps.d, ps.tmin, ps.tmax, t) &&
          potential(t, id, ps.tmin, tv.tbest, tv.bestPrim)) { tv.tbest = t; tv.bestPrim = id; }
    }
    cnt<CNT>(ct.analyticTests, (uint32_t)(sc.nSpheres + sc.nQuads));
  } else if (sc.anyDisneyAnalytic) {
    for (int i = 0; i < sc.nSpheres && !terminated; i++) {
      const int mat = load_uniform(sc.sphereMat + i);
      if (sc.mats[mat].kind != MAT_DISNEY) continue;
      const DevSphere s = load_uniform(sc.spheres + i);
      float t1, t2;
      if (!sphere_roots(s.center, s.radius, ps.o, ps.d, t1, t2)) continue;
      if (sc.shadowNearest) {                                // the roots in the order sphereIntersect reports them
        shadow_candidate(sc, mat, t1, i, ps.tmin, tv.tbest, tv.bestPrim, tv.att);
        shadow_candidate(sc, mat, t2, i, ps.tmin, tv.tbest, tv.bestPrim, tv.att);
      } else if ((t1 > ps.tmin && t1 < ps.tmax) || (t2 > ps.tmin && t2 < ps.tmax))
        terminated = shadow_any_hit(sc, mat, tv.att);
    }
    for (int i = 0; i < sc.nQuads && !terminated; i++) {
      const DevQuad q = load_uniform(sc.quads + i);
      if (sc.mats[q.mat].kind != MAT_DISNEY) continue;
      float t;
      if (quad_test(q.plane, q.v1, q.v2, q.anchor, ps.o, ps.d, ps.tmin, ps.tmax, t)) {
        if (sc.shadowNearest) shadow_candidate(sc, q.mat, t, sc.nSpheres + i, ps.tmin, tv.tbest, tv.bestPrim, tv.att);
        else terminated = shadow_any_hit(sc, q.mat, tv.att);
      }
    }
    cnt<CNT>(ct.analyticTests, (uint32_t)(sc.nSpheres + sc.nQuads));
  }
  tv.node = (terminated || sc.rootRef == kEmptyRef) ? kTravDone : sc.rootRef;
}

// Conservative slab test of one child box.  Never culls a box whose triangles the exact
// test could accept: node boxes are padded at build time and the far bound gets one ulp of
// slack; "<=" keeps boxes that start exactly at tbest (needed for the equal-t rule).
PT_HD bool slab(v3 lo, v3 hi, v3 o, v3 inv, float tmin, float tmax, float& tn) {
  float t0x = (lo.x - o.x) * inv.x, t1x = (hi.x - o.x) * inv.x;
  float t0y = (lo.y - o.y) * inv.y, t1y = (hi.y - o.y) * inv.y;
  float t0z = (lo.z - o.z) * inv.z, t1z = (hi.z - o.z) * inv.z;
  tn = fmaxf_(fmaxf_(fminf_(t0x, t1x), fminf_(t0y, t1y)), fmaxf_(fminf_(t0z, t1z), tmin));
  float tf = fminf_(fminf_(fmaxf_(t0x, t1x), fmaxf_(t0y, t1y)), fminf_(fmaxf_(t0z, t1z), tmax));
  return tn <= tf * 1.0000005f;
}

template <class Stack>
PT_HD void trav_pop(Trav& tv, Stack& st) {
  if (tv.sp == 0) tv.node = kTravDone;
  else { tv.sp--; tv.node = st.load(tv.sp); }
}

// One four-child node for a lane with tv.node >= 0: the children the ray enters are visited nearest first; the order
// affects only the amount of work, never the result (equal-t rule D5 is order independent).
//
// Slab planes are evaluated as t = fma(plane, 1/d, -(o/d)): one instruction per plane.  Compared with (plane - o) * (1/d) the
// rounding error moves by about one ulp of o in space, far inside the padding of the boxes (pt_lbvh.h pad_lo/pad_hi); 1/d is
// kept finite (slab_inv).  The test stays conservative, and nothing downstream depends on which boxes were entered.
//
// What the node step costs is set by WHICH vector instructions it issues (tools/micro/valu_issue.hip,
// profiles/r04_valu_ceiling.txt): on gfx950 v_fma / v_mul / v_add / v_mov / v_and / v_or / v_lshrrev issue at ~2.35 clocks per
// wave64 instruction per SIMD, everything else this code needs -- v_min / v_max / v_max3, v_cvt_f32_ubyteN, v_cmp, v_cndmask,
// the f64 min / max -- at ~4.3.  Hence
//   * near / far plane per axis by the SIGN of the ray's direction, chosen on the packed plane words of the 64-byte node
//     before they are unpacked (6 selects per node) instead of ordering each pair of plane distances (24 min / max);
//   * the entered children are sorted as 64-bit keys (entry distance's bits in the high word, child reference in the low
//     word) with v_min_f64 / v_max_f64: 10 instructions for the 5-exchange network instead of 5 compares + 20 selects -- a
//     positive float's bits order like the float, and a double whose high word they are orders like its high word;
//   * children 0 and 1 of a node always exist (pt_lbvh.h emits 2..4 children, unused slots last): two empty tests, not four.
struct ChildKey {
#if defined(__HIP_DEVICE_COMPILE__)
  double v;
  PT_HD static ChildKey make(float t, int ref) {
    ChildKey k; k.v = __builtin_bit_cast(double, ((unsigned long long)(uint32_t)f2i(t) << 32) | (uint32_t)ref); return k;
  }
  PT_HD float t() const { return i2f((int32_t)(__builtin_bit_cast(unsigned long long, v) >> 32)); }
  PT_HD int ref() const { return (int32_t)(uint32_t)__builtin_bit_cast(unsigned long long, v); }
  // inline assembly: llvm.minnum on a value that came out of a bit cast gets a canonicalising v_max_f64 x, x in front
  PT_HD static void order(ChildKey& a, ChildKey& b) {
    double lo, hi;
    asm("v_min_f64 %0, %1, %2" : "=v"(lo) : "v"(a.v), "v"(b.v));
    asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(a.v), "v"(b.v));
    a.v = lo; b.v = hi;
  }
#else
  unsigned long long v;      // same order: the keys are positive normal doubles (t >= tmin > 0, below the NaN patterns)
  PT_HD static ChildKey make(float t, int ref) { ChildKey k; k.v = ((unsigned long long)(uint32_t)f2i(t) << 32) | (uint32_t)ref; return k; }
  PT_HD float t() const { return i2f((int32_t)(v >> 32)); }
  PT_HD int ref() const { return (int32_t)(uint32_t)v; }
  PT_HD static void order(ChildKey& a, ChildKey& b) { if (b.v < a.v) { const unsigned long long x = a.v; a.v = b.v; b.v = x; } }
#endif
};
#ifndef PT_STACK_ROOMY
#define PT_STACK_ROOMY 1
#endif
// The slab tests, the sort and the pushes of one four-child node whose plane distances are known: tnr / tfr = distance to
// the plane of child k's box the ray meets first / last, per axis.  needTest[k]: child k may be an unused slot.
template <bool CNT, class Stack>
PT_HD void node_step_nearfar(const PathState& ps, Trav& tv, Stack& st, Counters& ct, const float nx[4], const float fx[4], const float ny[4],
                             const float fy[4], const float nz[4], const float fz[4], int r0, int r1, int r2, int r3, int top) {
  const int r[4] = { r0, r1, r2, r3 };
  cnt<CNT>(ct.nodeFetches);
  const float kFar = 3.0e38f;
  ChildKey k[4];
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int c = 0; c < 4; c++) {
    const float tn = fmaxf_(fmaxf_(nx[c], ny[c]), fmaxf_(nz[c], ps.tmin));
    const float tf = fminf_(fminf_(fx[c], fy[c]), fminf_(fz[c], tv.tbest));
    const bool in = (c < 2) ? (tn <= tf * 1.0000005f) : (tn <= tf * 1.0000005f && r[c] != kEmptyRef);
    k[c] = ChildKey::make(in ? tn : kFar, r[c]);
  }
  ChildKey::order(k[0], k[1]); ChildKey::order(k[2], k[3]); ChildKey::order(k[0], k[2]);
  ChildKey::order(k[1], k[3]); ChildKey::order(k[1], k[2]);
  if constexpr (Stack::kFlat) {
    // The step's tail without a branch (the common case): sorted keys are hits first, so with h_j = "child j of the order is
    // entered" the m = h1 + h2 + h3 pushes are k[m] .. k[1], farthest first.  All three stores are issued: store i goes to entry
    // sp + min(i, m - 1) and carries what belongs there (the stores past the last real one repeat it; with m = 0 they put
    // a dead value just above the top), and the entry below the top was read before the node's record arrived (`top`): a lane
    // that enters no child takes it.  ~60 scalar instructions (exec masks, skips) per step in the branched form.
    const bool h0 = k[0].t() < kFar, h1 = k[1].t() < kFar, h2 = k[2].t() < kFar, h3 = k[3].t() < kFar;
    const int m = (int)h1 + (int)h2 + (int)h3;
    if (__builtin_expect(st.fits_fast(tv.sp, m), 1)) {
      const int r1 = k[1].ref(), r2 = k[2].ref(), r3 = k[3].ref();
      const int p1 = tv.sp + (int)h2, p2 = p1 + (int)h3;
      st.store_fast(tv.sp, h3 ? r3 : (h2 ? r2 : r1)); st.store_fast(p1, h3 ? r2 : r1); st.store_fast(p2, r1);
      tv.node = h0 ? k[0].ref() : (tv.sp > 0 ? top : kTravDone);
      tv.sp = h0 ? p2 + (int)h1 : (tv.sp > 0 ? tv.sp - 1 : 0);
      return;
    }
    census<CNT>(ct, CR_NODE_BRANCHED);
  }
  if (k[0].t() < kFar) {
    // one test for the step instead of one per push: do three more entries fit the stack's fast part?
    if (PT_STACK_ROOMY && st.roomy(tv.sp)) {
      if (k[3].t() < kFar) { st.store_fast(tv.sp, k[3].ref()); tv.sp++; }
      if (k[2].t() < kFar) { st.store_fast(tv.sp, k[2].ref()); tv.sp++; }
      if (k[1].t() < kFar) { st.store_fast(tv.sp, k[1].ref()); tv.sp++; }
    } else {
      if (k[3].t() < kFar) { st.store(tv.sp, k[3].ref()); tv.sp++; }
      if (k[2].t() < kFar) { st.store(tv.sp, k[2].ref()); tv.sp++; }
      if (k[1].t() < kFar) { st.store(tv.sp, k[1].ref()); tv.sp++; }
    }
    tv.node = k[0].ref();
  } else {
    trav_pop(tv, st);
  }
}
PT_HD void planes4(const v4& p, float inv, float noi, float out[4]) {
  out[0] = fma_(p.x, inv, noi); out[1] = fma_(p.y, inv, noi); out[2] = fma_(p.z, inv, noi); out[3] = fma_(p.w, inv, noi);
}
// N64: the node loop fetches the 64-byte form of the nodes (pt_types.h Node64): four look-ups per lane instead of seven.
// The decode is folded into the slab test: plane = corner + q * step, so
//   t = (plane - o) / d = q * (step / d) + (corner - o) / d = fma(q, step * (1/d), fma(corner, 1/d, -o/d))
// -- per node three multiplications and three fma, per plane one v_cvt_f32_ubyteN and the fma the uncompressed form needs as
// well.  With d > 0 the lower plane of a box is the one the ray meets first, with d < 0 the upper one: fma is monotone in q,
// so picking the word by the sign gives exactly min / max of the two distances.  A template parameter, not a flag of the
// scene: with both decodes in one kernel the packet kernel ran 7-10 % slower, for either format.
PT_HD void planes4q(uint32_t w, float step, float base, float out[4]) {
  out[0] = fma_((float)(w & 0xffu), step, base); out[1] = fma_((float)((w >> 8) & 0xffu), step, base);
  out[2] = fma_((float)((w >> 16) & 0xffu), step, base); out[3] = fma_((float)(w >> 24), step, base);
}
// the step for a 64-byte record
template <bool CNT, class Stack>
PT_HD void trav_node_step64(const Node64& n, const PathState& ps, Trav& tv, Stack& st, Counters& ct, int top) {
  float nx[4], fx[4], ny[4], fy[4], nz[4], fz[4];
  const float sx = n.sx * tv.inv.x, sy = n.sy * tv.inv.y, sz = n.sz * tv.inv.z;
  const float cx = fma_(n.ox, tv.inv.x, tv.noi.x), cy = fma_(n.oy, tv.inv.y, tv.noi.y), cz = fma_(n.oz, tv.inv.z, tv.noi.z);
  const bool bx = tv.inv.x < 0.f, by = tv.inv.y < 0.f, bz = tv.inv.z < 0.f;      // the same for every node of a ray
  planes4q(bx ? n.q[3] : n.q[0], sx, cx, nx); planes4q(bx ? n.q[0] : n.q[3], sx, cx, fx);
  planes4q(by ? n.q[4] : n.q[1], sy, cy, ny); planes4q(by ? n.q[1] : n.q[4], sy, cy, fy);
  planes4q(bz ? n.q[5] : n.q[2], sz, cz, nz); planes4q(bz ? n.q[2] : n.q[5], sz, cz, fz);
  node_step_nearfar<CNT>(ps, tv, st, ct, nx, fx, ny, fy, nz, fz, n.ref[0], n.ref[1], n.ref[2], n.ref[3], top);
}
template <bool CNT, bool N64 = false, class Stack>
PT_HD void trav_node_step(const SceneView& sc, const PathState& ps, Trav& tv, Stack& st, Counters& ct) {
  float nx[4], fx[4], ny[4], fy[4], nz[4], fz[4];
  int top = kTravDone;
  if constexpr (Stack::kFlat) top = st.peek_fast(tv.sp);      // the entry below the top, requested together with the node
  if constexpr (N64) {
    const Node64 n = load_const(at32(sc.nodes64, tv.node));
    trav_node_step64<CNT>(n, ps, tv, st, ct, top);
  } else {
#if defined(__HIP_DEVICE_COMPILE__)
    // Near / far plane per axis chosen by the ADDRESS the plane vector is fetched from: with d > 0 the lower planes of the four
    // boxes are the ones the ray meets first (record offset 0 / 16 / 32), with d < 0 the upper ones (+48); fma is monotone in the
    // plane, so this is exactly min / max of the two distances.  Six address additions (2.35-clock class; the six offsets are
    // loop-invariant and live in registers) instead of 24 v_min / v_max (4.3): coffee 81.5 -> 78.9 ms at 64 spp on the packet
    // kernel, same bits (round 5).  The 64-byte form keeps its lead (76.8 ms) although it issues 36 instructions more per step:
    // a wave's seven gathers are served one after the other by the CU's address unit (64 lanes each), and that time is on the
    // step's critical path (NOTEBOOK.md, round 5).  The host build keeps the plain form below (same values).
    const char* nb = reinterpret_cast<const char*>(sc.nodes);
    const uint32_t no = (uint32_t)tv.node * (uint32_t)sizeof(Node128);
    const uint32_t sx = (uint32_t)(f2i(tv.inv.x) >> 31) & 48u, sy = (uint32_t)(f2i(tv.inv.y) >> 31) & 48u, sz = (uint32_t)(f2i(tv.inv.z) >> 31) & 48u;
    const v4 pnx = load_const(reinterpret_cast<const v4*>(nb + (no + sx)));
    const v4 pfx = load_const(reinterpret_cast<const v4*>(nb + (no + (48u - sx))));
    const v4 pny = load_const(reinterpret_cast<const v4*>(nb + (no + (16u + sy))));
    const v4 pfy = load_const(reinterpret_cast<const v4*>(nb + (no + (64u - sy))));
    const v4 pnz = load_const(reinterpret_cast<const v4*>(nb + (no + (32u + sz))));
    const v4 pfz = load_const(reinterpret_cast<const v4*>(nb + (no + (80u - sz))));
    const i4r rf = load_const(reinterpret_cast<const i4r*>(nb + (no + 96u)));
    planes4(pnx, tv.inv.x, tv.noi.x, nx); planes4(pfx, tv.inv.x, tv.noi.x, fx);
    planes4(pny, tv.inv.y, tv.noi.y, ny); planes4(pfy, tv.inv.y, tv.noi.y, fy);
    planes4(pnz, tv.inv.z, tv.noi.z, nz); planes4(pfz, tv.inv.z, tv.noi.z, fz);
    node_step_nearfar<CNT>(ps, tv, st, ct, nx, fx, ny, fy, nz, fz, rf.x, rf.y, rf.z, rf.w, top);
    return;
#endif
    const Node128 n = load_const(at32(sc.nodes, tv.node));
    float a[4], b[4];
    planes4(n.lox, tv.inv.x, tv.noi.x, a); planes4(n.hix, tv.inv.x, tv.noi.x, b);
    for (int c = 0; c < 4; c++) { nx[c] = fminf_(a[c], b[c]); fx[c] = fmaxf_(a[c], b[c]); }
    planes4(n.loy, tv.inv.y, tv.noi.y, a); planes4(n.hiy, tv.inv.y, tv.noi.y, b);
    for (int c = 0; c < 4; c++) { ny[c] = fminf_(a[c], b[c]); fy[c] = fmaxf_(a[c], b[c]); }
    planes4(n.loz, tv.inv.z, tv.noi.z, a); planes4(n.hiz, tv.inv.z, tv.noi.z, b);
    for (int c = 0; c < 4; c++) { nz[c] = fminf_(a[c], b[c]); fz[c] = fmaxf_(a[c], b[c]); }
    node_step_nearfar<CNT>(ps, tv, st, ct, nx, fx, ny, fy, nz, fz, n.ref[0], n.ref[1], n.ref[2], n.ref[3], top);
  }
}

// One leaf (count x 48-byte triangle records) for a lane with tv.node < 0.  The records of up to four
// triangles are fetched together (one memory round trip per chunk instead of one per triangle; slots past the
// end of the leaf re-read its last record, which costs no extra line) and then tested in order.  A caller that
// knows the leaf early (queuekernel.hip: from LDS) issues leaf_fetch4 for the first chunk itself, together with
// its other loads.
struct LeafChunk { v3 p0[4], e0[4], e1[4]; int mat[4], prim[4], shadow[4]; };
PT_HD void leaf_fetch4(const SceneView& sc, int leafRef, int base, LeafChunk& ch) {
  const int first = leaf_first(leafRef), count = leaf_count(leafRef);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int j = 0; j < 4; j++) {
    const int k = base + j < count ? base + j : count - 1;
    const Tri48 tpv = load_const(at32(sc.tris, first + k));
    const Tri48* tp = &tpv;
    ch.p0[j] = tp->p0; ch.e0[j] = tp->e0; ch.e1[j] = tp->e1; ch.mat[j] = tp->mat; ch.prim[j] = tp->prim; ch.shadow[j] = tp->shadow;
  }
}
template <bool CNT, class Stack>
PT_HD void trav_leaf_step_fetched(const SceneView& sc, const PathState& ps, Trav& tv, Stack& st, Counters& ct, LeafChunk& ch) {
  {
    const int first = leaf_first(tv.node), count = leaf_count(tv.node);
    const int triBase = sc.nSpheres + sc.nQuads;
    bool terminated = false;
    // a shadow ray that keeps its nearest candidate is range-tested by potential() against tbest (callers that resume a ray
    // hand tbest over as ps.tmax, and a candidate at exactly tbest must still reach the equal-t rule)
    const float tmaxTest = (ps.kind == RK_SHADOW && sc.shadowNearest) ? kRtDefaultMax : ps.tmax;
    for (int base = 0; base < count && !terminated; base += 4) {
      if (base > 0) leaf_fetch4(sc, tv.node, base, ch);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (base + j < count && !terminated) {
          cnt<CNT>(ct.triTests); census<CNT>(ct, CR_TRI0 + j);
          v3 n; float t, be, ga;
          if (tri_test(ps.o, ps.d, ps.tmin, tmaxTest, ch.p0[j], ch.e0[j], ch.e1[j], n, t, be, ga)) {   // meshIntersect, Geometry.cu:121-160
            census<CNT>(ct, CR_TRI_HIT);
            if (ps.kind == RK_RADIANCE) {
              if (potential(t, triBase + ch.prim[j], ps.tmin, tv.tbest, tv.bestPrim)) {
                tv.tbest = t; tv.bestPrim = triBase + ch.prim[j]; tv.bestTri = first + base + j; tv.beta = be; tv.gamma = ga; tv.bestCls = ch.shadow[j];
              }
            } else if (sc.shadowNearest) {
              shadow_candidate_tri(sc, ch.shadow[j], ch.mat[j], t, triBase + ch.prim[j], ps.tmin, tv.tbest, tv.bestPrim, tv.att);
            } else if (shadow_any_hit_tri(sc, ch.shadow[j], ch.mat[j], tv.att)) terminated = true;
          }
        }
      }
    }
    if (terminated) tv.node = kTravDone;
    else trav_pop(tv, st);
  }
}
template <bool CNT, class Stack>
PT_HD void trav_leaf_step(const SceneView& sc, const PathState& ps, Trav& tv, Stack& st, Counters& ct) {
  LeafChunk ch;
  leaf_fetch4(sc, tv.node, 0, ch);
  trav_leaf_step_fetched<CNT>(sc, ps, tv, st, ct, ch);
}

// One traversal step for a lane with tv.node != kTravDone (if-if form; the kernels use the
// two halves separately as a while-while loop).
template <bool CNT, bool N64 = false, class Stack>
PT_HD void trav_step(const SceneView& sc, const PathState& ps, Trav& tv, Stack& st, Counters& ct) {
  if (tv.node >= 0) trav_node_step<CNT, N64>(sc, ps, tv, st, ct);
  else trav_leaf_step<CNT>(sc, ps, tv, st, ct);
}

// ---------------------------------------------------------------------------------------
// Path state machine
// ---------------------------------------------------------------------------------------

// Camera.cu:21-38: per-pixel seed, thin-lens offset, jittered pixel, primary ray
template <bool CNT>
PT_HD void begin_sample(const SceneView& sc, PathState& ps, int launchSeed, Counters& ct) {
  const int x = ps.pixel % sc.width, y = ps.pixel / sc.width;
  ps.depth = 1;
  ps.seed = tea16((uint32_t)y * (uint32_t)sc.width + (uint32_t)x, (uint32_t)launchSeed);
  ps.thr = mk3(1.f, 1.f, 1.f); ps.rad = mk3(0.f, 0.f, 0.f);
  const v3 randInLens = rand_in_unit_disk(ps.seed) * sc.cam.lensRadius;
  const v3 offs = sc.cam.u * randInLens.x + sc.cam.v * randInLens.y;
  const float r1 = rnd(ps.seed); const float r2 = rnd(ps.seed);
  const float xyx = ((float)x + r1 - 0.5f) / (float)sc.width;
  const float xyy = ((float)y + r2 - 0.5f) / (float)sc.height;
  ps.o = sc.cam.origin + offs;
  ps.d = normalize((((sc.cam.scrLowerLeftCorner + sc.cam.horizontal * xyx) + sc.cam.vertical * xyy) - sc.cam.origin) - offs);
  ps.tmin = sc.epsT; ps.tmax = kRtDefaultMax; ps.kind = RK_RADIANCE;
  ps.mode = M_TRACE;
  cnt<CNT>(ct.samples); cnt<CNT>(ct.primaryRays); census<CNT>(ct, CR_GEN);
}

// Camera.cu:39: clamp the sample.  The add into accuBuffer (Camera.cu:41) is done by the
// caller, in launch order per pixel (ordered reduction over the per-sample buffer).
PT_HD void end_sample(PathState& ps) {
  ps.accum = mk3(clampf(ps.rad.x, 0.f, 1.f), clampf(ps.rad.y, 0.f, 1.f), clampf(ps.rad.z, 0.f, 1.f));
  ps.mode = M_NEW_SAMPLE;
}

// continuation ray of a closest-hit program: child payload = folkPayload(parent)
PT_HD void bounce(const SceneView& sc, PathState& ps, v3 o, v3 d, uint32_t childSeed) {
  ps.o = o; ps.d = d; ps.tmin = sc.epsT; ps.tmax = kRtDefaultMax; ps.kind = RK_RADIANCE;
  ps.depth++; ps.seed = childSeed;
  ps.mode = M_TRACE;
}

// Material.cu:72-110 glass and :134-168 Disney GLASS branch (fork the seed, THEN draw)
template <bool CNT>
PT_HD void glass_body(const SceneView& sc, PathState& ps, float ior, v3 tint, const HitAttr& h, Counters& ct) {
  v3 normal = h.shadingNormal;
  float cosThetaI = -dot(ps.d, normal);
  float refIdx;
  if (cosThetaI > 0.f) { refIdx = ior; }
  else { refIdx = 1.f / ior; cosThetaI = -cosThetaI; normal = -normal; }
  v3 refracted;
  const bool totalReflection = !refract(refracted, ps.d, normal, refIdx);
  const float cosThetaT = -dot(normal, refracted);
  const float reflectProb = totalReflection ? 1.f : fresnel(cosThetaI, cosThetaT, refIdx);
  const uint32_t childSeed = fork_seed(ps.seed, ps.depth + 1);
  v3 no, nd;
  if (rnd(ps.seed) < reflectProb) { no = h.front; nd = reflect(ps.d, normal); }
  else { no = h.back; nd = refracted; }
  ps.thr = ps.thr * tint;
  cnt<CNT>(ct.bounceRays);
  bounce(sc, ps, no, nd, childSeed);
}

// Hit attributes of the nearest primitive (Geometry.cu:30-37, 81-86, 134-157)
PT_HD void hit_attributes(const SceneView& sc, const PathState& ps, const Trav& tv, HitAttr& h) {
  const float t = tv.tbest;
  if (tv.bestPrim < sc.nSpheres) {
    const DevSphere s = sc.spheres[tv.bestPrim];
    const v3 p = ray_at(ps.o, ps.d, t);
    h.geoNormal = normalize(p - s.center);
    h.shadingNormal = h.geoNormal;
    h.texu = 0.f; h.texv = 0.f;                                 // Geometry.cu:37
    h.front = p; h.back = p;
    h.mat = sc.sphereMat[tv.bestPrim];
  } else if (tv.bestPrim < sc.nSpheres + sc.nQuads) {
    const DevQuad* q = sc.quads + (tv.bestPrim - sc.nSpheres);
    const v3 n = xyz(q->plane);
    h.geoNormal = n; h.shadingNormal = n;
    h.texu = 0.f; h.texv = 0.f;                                 // the quad program writes no texcoord
    h.front = ray_at(ps.o, ps.d, t); h.back = h.front;
    h.mat = q->mat;
  } else {
    const Tri48 tpv = load_const(at32(sc.tris, tv.bestTri));
    const TriShade spv = load_const(at32(sc.triShade, tv.bestTri));
    const Tri48* tp = &tpv;
    const TriShade* sp = &spv;
    // both records are requested before either is used: one memory round trip, not two
    const v3 p0 = tp->p0, e0 = tp->e0, e1 = tp->e1;
    const v3 sn0 = sp->n0, sn1 = sp->n1, sn2 = sp->n2;
    const int hasNormals = sp->hasNormals;
    h.mat = tp->mat;
    h.texu = 0.f; h.texv = 0.f;
    if (sc.triUV != nullptr && at32(sc.mats, h.mat)->albedoTex != 0) {               // Geometry.cu:141-148
      const TriUV* up = sc.triUV + tp->prim;
      if (up->hasUV) {
        const float w0 = 1.f - tv.beta - tv.gamma;
        h.texu = (up->u1 * tv.beta + up->u2 * tv.gamma) + up->u0 * w0;
        h.texv = (up->v1 * tv.beta + up->v2 * tv.gamma) + up->v0 * w0;
      }
    }
    h.geoNormal = normalize(cross(e1, e0));
    if (hasNormals) {
      const v3 n0 = sn0, n1 = sn1, n2 = sn2;
      h.shadingNormal = normalize((n1 * tv.beta + n2 * tv.gamma) + n0 * (1.f - tv.beta - tv.gamma));
    } else {
      h.shadingNormal = h.geoNormal;
    }
    refine_hitpoint(ray_at(ps.o, ps.d, t), ps.d, h.geoNormal, p0, h.back, h.front);
  }
}

// The ray in flight has finished: run miss / closest-hit (radiance) or fold the shadow
// result into the radiance (shadow).
template <bool CNT>
PT_HD void on_result(const SceneView& sc, PathState& ps, const Trav& tv, Counters& ct) {
  if (ps.kind == RK_SHADOW) {                                   // Material.cu:193-201
    if (ps.pendInv != 0.f && length_is_nonzero(tv.att)) {
      const v3 c = (ps.pendW * tv.att) * ps.pendInv;
      ps.rad = ps.rad + ps.thr * c;
    }
    ps.light++;
    ps.mode = M_LIGHTS;
    return;
  }
  if (tv.bestPrim < 0) {                                        // staticMiss, miss.cu:10-12
    census<CNT>(ct, CR_MISS);
    ps.rad = ps.rad + ps.thr * sc.bg;
    end_sample(ps);
    return;
  }
  cnt<CNT>(ct.closestHits); census<CNT>(ct, CR_HIT);
  HitAttr h;
  hit_attributes(sc, ps, tv, h);
  const DevMaterial m = load_const(at32(sc.mats, h.mat));
  if (m.kind == MAT_LIGHT) {                                    // light, Material.cu:238-240
    census<CNT>(ct, CR_LIGHT);
    ps.rad = ps.rad + ps.thr * m.emission;
    end_sample(ps);
    return;
  }
  // Material.cu:29,50,73,119: depth cap -> absorbColor (0,0,0).  The second half of that test,
  // length(payload.color) < rayMinIntensity, is dead in the reference: every payload starts
  // at (1,1,1).
  if (ps.depth > sc.maxDepth) { census<CNT>(ct, CR_DEPTHCAP); end_sample(ps); return; }

  if (m.kind == MAT_LAMBERTIAN) {                               // Material.cu:28-43
    census<CNT>(ct, CR_LAMBERT);
    const v3 no = ray_at(ps.o, ps.d, tv.tbest);
    const v3 nd = normalize(h.geoNormal + rand_in_unit_sphere(ps.seed));
    const uint32_t childSeed = fork_seed(ps.seed, ps.depth + 1);
    ps.thr = ps.thr * m.albedo;
    cnt<CNT>(ct.bounceRays);
    bounce(sc, ps, no, nd, childSeed);
  } else if (m.kind == MAT_METAL) {                             // Material.cu:49-66
    census<CNT>(ct, CR_METAL);
    const v3 no = ray_at(ps.o, ps.d, tv.tbest);
    const v3 nd = normalize(reflect(ps.d, h.geoNormal) + rand_in_unit_sphere(ps.seed) * m.fuzz);
    const uint32_t childSeed = fork_seed(ps.seed, ps.depth + 1);
    ps.thr = ps.thr * m.albedo;
    cnt<CNT>(ct.bounceRays);
    bounce(sc, ps, no, nd, childSeed);
  } else if (m.kind == MAT_GLASS) {                             // Material.cu:72-110
    census<CNT>(ct, CR_GLASS);
    glass_body<CNT>(sc, ps, m.refIdx, m.albedo, h, ct);
  } else {                                                      // disney, Material.cu:118-223
    v3 baseColor = m.color;
    if (m.albedoTex != 0) {                                     // Material.cu:128-132
      baseColor = xyz(tex2d(sc.textures[m.albedoTex - 1], h.texu, h.texv));
      ps.cdlin = srgb2lin(baseColor);
    }
    if (m.brdfType == BRDF_GLASS) { census<CNT>(ct, CR_DISNEY_GLASS); glass_body<CNT>(sc, ps, 1.45f, baseColor, h, ct); return; }
    census<CNT>(ct, CR_DISNEY);
    ps.N = faceforward(h.shadingNormal, -ps.d, h.geoNormal);
    ps.V = -ps.d;
    ps.mat = h.mat;
    ps.o = h.front;                 // every shadow ray and the bounce start at frontHitPoint
    ps.rad = ps.rad + ps.thr * m.emission;
    ps.light = 0;
    ps.mode = M_LIGHTS;
  }
}

// Disney next-event estimation loop + BRDF bounce (Material.cu:170-221).  Runs until the
// lane owns a ray again (shadow ray towards light `ps.light`, or the bounce) or the sample ends.
// The cheap part (light sampling, facing tests, BRDF direction sampling) picks ONE candidate
// direction per lane; disney_pdf/disney_eval -- the expensive part -- are then evaluated once,
// for light candidates and bounce candidates together, so a shading batch runs them on a full
// wave.  Same formulas and the same RNG draw order as the reference's program.
template <bool CNT, bool FAST = false>
PT_HD void on_lights(const SceneView& sc, PathState& ps, Counters& ct) {
  const DevMaterial m = load_const(at32(sc.mats, ps.mat));
  int choice = 0;                       // 0 nothing, 1 shadow ray towards a light, 2 BRDF bounce
  v3 L = mk3(0.f, 0.f, 1.f), H = mk3(0.f, 0.f, 1.f), emission = mk3(0.f, 0.f, 0.f);
  float lightDst = 0.f, lightPdf = 0.f;
  while (ps.light < sc.nLights) {
    const DevLight ltv = load_const(sc.lights + ps.light);      // per-lane light index on this path (variant 3)
    const DevLight* lt = &ltv;
    cnt<CNT>(ct.lightLoads);
    v3 pointOnLight, normalOnLight;
    if (lt->shape == LIGHT_SPHERE) {
      pointOnLight = lt->position + rand_in_unit_sphere(ps.seed) * lt->radius;
      normalOnLight = normalize(pointOnLight - lt->position);
    } else {
      const float r1 = rnd(ps.seed); const float r2 = rnd(ps.seed);
      pointOnLight = (lt->position + lt->u * r1) + lt->v * r2;
      normalOnLight = lt->normal;            // normalize(light.normal) (Material.cu:181), taken once at upload (pt_upload.h)
    }
    L = pointOnLight - ps.o;
    lightDst = length(L);
    L = normalize(L);
    if (dot(L, ps.N) > 0.f && dot(L, normalOnLight) < 0.f) {
      H = normalize(L + ps.V);
      lightPdf = lightDst * lightDst / lt->area / dot(normalOnLight, -L);
      emission = lt->emission;
      choice = 1;
      break;
    }
    ps.light++;
  }
  const Onb onb = make_onb(ps.N);
  if (choice == 0) {
    disney_sample(ps.seed, m, onb, ps.V, L, H);
    if (dot(ps.N, L) > 0.0f && dot(ps.N, ps.V) > 0.0f) choice = 2;
  }
  if (choice == 0) { end_sample(ps); return; }

  v3 Cdlin = m.Cdlin, Cspec0 = m.Cspec0, Csheen = m.Csheen;
  if (m.albedoTex != 0) {
    Cdlin = ps.cdlin;
    disney_color_constants(Cdlin, m.specular, m.specularTint, m.sheenTint, m.metallic, Cspec0, Csheen);
  }
  const float pdf = disney_pdf<FAST>(m, ps.N, L, H);
  const v3 brdf = disney_eval<FAST>(m, Cdlin, Cspec0, Csheen, onb, L, ps.V, H);

  if (choice == 1) {
    if (lightPdf > 0 && pdf > 0) {
      ps.pendW = (brdf * powerHeuristic(lightPdf, pdf)) * emission;
      ps.pendInv = 1.0f / fmaxf_(0.001f, lightPdf);
    } else {
      ps.pendW = mk3(0.f, 0.f, 0.f); ps.pendInv = 0.f;
    }
    ps.d = L; ps.tmin = sc.epsT; ps.tmax = lightDst - sc.epsT; ps.kind = RK_SHADOW;
    ps.mode = M_TRACE;
    cnt<CNT>(ct.shadowRays);
    return;
  }
  const uint32_t childSeed = fork_seed(ps.seed, ps.depth + 1);
  if (pdf > 0) {
    ps.thr = (ps.thr * brdf) * (1.0f / pdf);
    cnt<CNT>(ct.bounceRays);
    bounce(sc, ps, ps.o, L, childSeed);
    return;
  }
  end_sample(ps);
}

}  // namespace pt
