// pt_lbvh.h -- building blocks of the Morton-code LBVH that replaces
// Acceleration("Trbvh") (MinimalOptiX.cpp:378,494,534; closed source in OptiX).
//
//   1. per-triangle bounds + centroid                      (meshBBox, Geometry.cu:162-175)
//   2. Morton code of the centroid in the scene box; key = code << idxBits | face  (unique; the code gets
//      every bit the face index does not need: 15 bits per axis for the coffee scene)
//   3. radix sort of the keys
//   4. Karras 2012 radix tree over the sorted keys (one thread per internal node)
//   5. bottom-up box fit with one atomic arrival counter per node
//   6. collapse: every subtree with <= leafSize triangles becomes one leaf
//      (first,count) in sorted order
//   7. widening: a surviving node absorbs the surviving descendants with the largest surface area until it
//      has four children; the resulting nodes are compacted and emitted as 128-byte four-child nodes
//
// All steps are pure functions of the key order (min/max unions are exact), so the tree is
// bit-reproducible; tests/hostsim runs the same functions sequentially on the host and the
// GPU tests compare the device tree with that mirror word for word.
#pragma once
#include "pt_types.h"

namespace pt {

// Keys are code << idxBits | face: the face index makes them unique, and all bits it does not need go to
// the Morton code (3 x bitsPerAxis): 168 k triangles -> 18 index bits -> 15 bits (32768 cells) per axis.
PT_HD int lbvh_index_bits(int n) { int b = 1; while ((1ll << b) < (long long)n) b++; return b; }
PT_HD int lbvh_bits_per_axis(int n) { const int b = (63 - lbvh_index_bits(n)) / 3; return b > 21 ? 21 : b; }      // bit 63 is kSmallKeyBit
PT_HD uint64_t expand_bits21(uint64_t v) {
  v &= 0x1fffffull;
  v = (v | (v << 32)) & 0x001f00000000ffffull;
  v = (v | (v << 16)) & 0x001f0000ff0000ffull;
  v = (v | (v << 8)) & 0x100f00f00f00f00full;
  v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
}
// Large triangles first (round 6).  A room's walls and floor are a dozen triangles whose boxes span the scene; binned by centroid among a
// million small ones they stay inside subtrees of small geometry for many levels and inflate every box above them, and most rays of an
// interior view end on exactly those triangles.  So a triangle whose box has at least kBigTriShare of the scene box's surface area sorts in
// FRONT of all others (the others carry kSmallKeyBit) and the root's range is split between the two groups (SahTask::force; the Morton radix
// tree of builder 0 splits there by itself: it is the keys' highest bit).  Dining-room stand-in: 4.90 -> 2.64 node steps per ray on the
// per-lane walk, the exact-sweep SAH gives 2.90 (tools/tree_yardstick.py, profiles/r06_tree_yardstick.txt); scenes without such triangles:
// the same tree as before.
constexpr uint64_t kSmallKeyBit = 1ull << 63;
constexpr float kBigTriShare = 1.0f / 64.0f;
PT_HD bool tri_is_big(v3 lo, v3 hi, v3 slo, v3 shi) {
  const float dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z, sx = shi.x - slo.x, sy = shi.y - slo.y, sz = shi.z - slo.z;
  return (dx * dy + dy * dz) + dz * dx >= kBigTriShare * ((sx * sy + sy * sz) + sz * sx);
}
// number of keys in front of the first small one (sorted keys): the forced split of the root's range, 0 = none
PT_HD int big_key_count(const uint64_t* keys, int n) {
  int a = 0, b = n;
  while (a < b) { const int m = (a + b) >> 1; if (keys[m] & kSmallKeyBit) b = m; else a = m + 1; }
  return (a > 0 && a < n) ? a : 0;
}
PT_HD uint64_t morton_key(v3 p, v3 lo, v3 invExt, int bitsPerAxis, int idxBits, int face) {
  const v3 n = (p - lo) * invExt;
  const float scale = (float)(1u << bitsPerAxis), top = scale - 1.0f;
  const uint64_t qx = (uint64_t)fminf_(fmaxf_(n.x * scale, 0.0f), top);
  const uint64_t qy = (uint64_t)fminf_(fmaxf_(n.y * scale, 0.0f), top);
  const uint64_t qz = (uint64_t)fminf_(fmaxf_(n.z * scale, 0.0f), top);
  const uint64_t code = (expand_bits21(qx) << 2) | (expand_bits21(qy) << 1) | expand_bits21(qz);
  return (code << idxBits) | (uint64_t)(uint32_t)face;
}
PT_HD int key_face(uint64_t key, int idxBits) { return (int)(key & ((1ull << idxBits) - 1ull)); }

PT_HD float inv_extent(float lo, float hi) { const float e = hi - lo; return e > 0.0f ? 1.0f / e : 0.0f; }

PT_HD void tri_bounds(v3 p0, v3 p1, v3 p2, v3& lo, v3& hi) {
  lo = mk3(fminf_(fminf_(p0.x, p1.x), p2.x), fminf_(fminf_(p0.y, p1.y), p2.y), fminf_(fminf_(p0.z, p1.z), p2.z));
  hi = mk3(fmaxf_(fmaxf_(p0.x, p1.x), p2.x), fmaxf_(fmaxf_(p0.y, p1.y), p2.y), fmaxf_(fmaxf_(p0.z, p1.z), p2.z));
}
// conservative padding of a leaf box (see pt_path.h slab())
PT_HD float pad_lo(float v, float padAbs) { return v - (padAbs + 1e-6f * __builtin_fabsf(v)); }
PT_HD float pad_hi(float v, float padAbs) { return v + (padAbs + 1e-6f * __builtin_fabsf(v)); }

// Node128 -> Node64.  Every plane moves outwards to the node's 256-step grid, and by a margin of a few ulps more: the
// kernels fold the decode into the slab test (t = fma(q, step/d, (corner - o)/d), pt_path.h), whose rounding differs from
// fma(plane, 1/d, -o/d) by about an ulp of the larger coordinate, so the margin keeps the compressed test at least as
// accepting as the uncompressed one.  Returns false if a step beyond kNode64MaxStep would be needed (|1/d| is capped at
// 1e30 by slab_inv; step/d has to stay finite): scenes wider than 1e10 units.
constexpr float kNode64MaxStep = 6.0e7f;
PT_HD float node64_plane(float corner, float step, int q) { return fma_((float)q, step, corner); }
PT_HD bool compress_node(const Node128& n, Node64& out) {
  const float lo[3][4] = { { n.lox.x, n.lox.y, n.lox.z, n.lox.w }, { n.loy.x, n.loy.y, n.loy.z, n.loy.w }, { n.loz.x, n.loz.y, n.loz.z, n.loz.w } };
  const float hi[3][4] = { { n.hix.x, n.hix.y, n.hix.z, n.hix.w }, { n.hiy.x, n.hiy.y, n.hiy.z, n.hiy.w }, { n.hiz.x, n.hiz.y, n.hiz.z, n.hiz.w } };
  bool ok = true;
  float corner[3], step[3]; uint32_t ql[3] = { 0, 0, 0 }, qh[3] = { 0, 0, 0 };
  for (int a = 0; a < 3; a++) {
    float mn = 3.0e38f, mx = -3.0e38f;
    for (int k = 0; k < 4; k++) if (n.ref[k] != kEmptyRef) { mn = fminf_(mn, lo[a][k]); mx = fmaxf_(mx, hi[a][k]); }
    if (mn > mx) { mn = 0.f; mx = 0.f; }
    const float margin = 1e-6f * fmaxf_(__builtin_fabsf(mn), __builtin_fabsf(mx)) + 1e-30f;
    const float c = mn - margin, top = mx + margin;
    float s = fmaxf_((top - c) * (1.0f / 255.0f), 1e-36f);
    for (int it = 0; it < 64 && !(node64_plane(c, s, 255) >= top); it++) s = s * 1.0000002f + 1e-38f;      // a few ulps at most
    if (!(node64_plane(c, s, 255) >= top) || !(s <= kNode64MaxStep)) ok = false;
    const float inv = 1.0f / s;
    for (int k = 0; k < 4; k++) {
      int a0 = 255, a1 = 0;                 // an unused child is skipped by its ref (kEmptyRef); its bytes only mark it for readers of the array
      if (n.ref[k] != kEmptyRef) {
        const float l = lo[a][k] - margin, h = hi[a][k] + margin;
        a0 = (int)fminf_(fmaxf_(__builtin_floorf((l - c) * inv), 0.f), 255.f);
        while (a0 > 0 && node64_plane(c, s, a0) > l) a0--;
        a1 = (int)fminf_(fmaxf_(__builtin_ceilf((h - c) * inv), 0.f), 255.f);
        while (a1 < 255 && node64_plane(c, s, a1) < h) a1++;
      }
      ql[a] |= (uint32_t)a0 << (8 * k); qh[a] |= (uint32_t)a1 << (8 * k);
    }
    corner[a] = c; step[a] = s;
  }
  out.ox = corner[0]; out.oy = corner[1]; out.oz = corner[2];
  out.sx = step[0]; out.sy = step[1]; out.sz = step[2];
  out.q[0] = ql[0]; out.q[1] = ql[1]; out.q[2] = ql[2]; out.q[3] = qh[0]; out.q[4] = qh[1]; out.q[5] = qh[2];
  for (int k = 0; k < 4; k++) out.ref[k] = n.ref[k];
  return ok;
}

// order-preserving float <-> uint mapping for atomicMin/atomicMax on floats
PT_HD uint32_t float_to_ordered(float f) { uint32_t u = (uint32_t)f2i(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
PT_HD float ordered_to_float(uint32_t u) { return i2f((int32_t)((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u)); }

PT_HD int clz64(uint64_t x) { return x ? __builtin_clzll(x) : 64; }
// common-prefix length of keys i and j (-1 outside the array); keys are unique
PT_HD int lcp(const uint64_t* keys, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  return clz64(keys[i] ^ keys[j]);
}

PT_HD float half_area(const float* lo, const float* hi) {
  const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
  return (dx * dy + dy * dz) + dz * dx;
}
// ---------------------------------------------------------------------------------------------------------------
// Binned-SAH topology ("Trbvh" asks OptiX for a high-quality tree; a plain Morton radix tree needs 9.7 node fetches
// per ray on the coffee scene where a surface-area-heuristic tree needs 7.5).  Same array form as the Karras tree
// (left / right / first / last / parent over the triangles in a final order, leaves = single triangles), so the box
// fit, the collapse, the widening and the emission below are shared.  Built level by level, top down:
//   * the triangles start in Morton order (locality for the passes that follow);
//   * a node with more than leafSize triangles bins the centroids of its range into kSahBins bins per axis inside
//     the range's centroid box, sweeps the 3 x (kSahBins-1) candidate planes and takes the cheapest
//     area(left) * count(left) + area(right) * count(right) (first one in sweep order on a tie); its range is
//     partitioned STABLY by "bin < plane";
//   * a node whose centroids coincide, every node with <= leafSize triangles (they are collapsed into leaves
//     later, the levels below only complete the array form) and every node below level kSahLevels splits its range
//     in the middle;
//   * nodes are numbered breadth first, children in range order.
// min / max / counts are exact and the sweep is one sequential function, so the tree is a pure function of the
// input: tests/hostsim runs the same functions on the host and the GPU tests compare the trees word for word.
#ifndef PT_SAH_BINS
#define PT_SAH_BINS 16
#endif
constexpr int kSahBins = PT_SAH_BINS;
constexpr int kSahLevels = 40;         // deeper levels split in the middle: bounds the depth at 40 + log2(n) <= 64 (wide_level's path)
struct SahBins {                       // per axis and bin: box of the triangles (order-preserving uints) + their number
  uint32_t lo[3][kSahBins][3], hi[3][kSahBins][3];
  int cnt[3][kSahBins];
};
struct SahSplit { int axis, bin, nLeft; };          // axis < 0: split the range in the middle
PT_HD float sah_scale(float lo, float hi) { const float e = hi - lo; return e > 0.0f ? (float)kSahBins / e : 0.0f; }
PT_HD int sah_bin(float c, float lo, float scale) {
  int b = (int)((c - lo) * scale);
  return b < 0 ? 0 : (b >= kSahBins ? kSahBins - 1 : b);
}
PT_HD SahSplit sah_choose(const SahBins& B, v3 cbLo, v3 cbHi, int count) {
  SahSplit best; best.axis = -1; best.bin = 0; best.nLeft = (count + 1) / 2;
  float bestCost = 3.0e38f;
  const float ext[3] = { cbHi.x - cbLo.x, cbHi.y - cbLo.y, cbHi.z - cbLo.z };
  for (int a = 0; a < 3; a++) {
    if (!(ext[a] > 0.0f)) continue;
    float la[kSahBins]; int lc[kSahBins];
    float l0[3] = { 1e37f, 1e37f, 1e37f }, l1[3] = { -1e37f, -1e37f, -1e37f };
    int c = 0;
    for (int i = 0; i < kSahBins; i++) {
      if (B.cnt[a][i] > 0)
        for (int k = 0; k < 3; k++) { l0[k] = fminf_(l0[k], ordered_to_float(B.lo[a][i][k])); l1[k] = fmaxf_(l1[k], ordered_to_float(B.hi[a][i][k])); }
      c += B.cnt[a][i];
      la[i] = c > 0 ? half_area(l0, l1) : 0.0f; lc[i] = c;
    }
    float r0[3] = { 1e37f, 1e37f, 1e37f }, r1[3] = { -1e37f, -1e37f, -1e37f };
    c = 0;
    for (int i = kSahBins - 1; i >= 1; i--) {
      if (B.cnt[a][i] > 0)
        for (int k = 0; k < 3; k++) { r0[k] = fminf_(r0[k], ordered_to_float(B.lo[a][i][k])); r1[k] = fmaxf_(r1[k], ordered_to_float(B.hi[a][i][k])); }
      c += B.cnt[a][i];
      if (c == 0 || lc[i - 1] == 0) continue;
      const float cost = la[i - 1] * (float)lc[i - 1] + half_area(r0, r1) * (float)c;
      if (cost < bestCost) { bestCost = cost; best.axis = a; best.bin = i; best.nLeft = lc[i - 1]; }
    }
  }
  return best;
}

// child reference inside the *uncollapsed* Karras tree: >=0 internal node, <0 -> ~sortedLeaf
struct KarrasNode { int left, right, first, last; };

// Karras, "Maximizing Parallelism in the Construction of BVHs, Octrees, and k-d Trees", 2012, Alg. 1
PT_HD KarrasNode karras_node(const uint64_t* keys, int n, int i) {
  const int d = (lcp(keys, n, i, i + 1) - lcp(keys, n, i, i - 1)) >= 0 ? 1 : -1;
  const int dmin = lcp(keys, n, i, i - d);
  int lmax = 2;
  while (lcp(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
  int l = 0;
  for (int t = lmax >> 1; t >= 1; t >>= 1)
    if (lcp(keys, n, i, i + (l + t) * d) > dmin) l += t;
  const int j = i + l * d;
  const int dnode = lcp(keys, n, i, j);
  int s = 0;
  int t = l;
  do {
    t = (t + 1) >> 1;
    if (lcp(keys, n, i, i + (s + t) * d) > dnode) s += t;
  } while (t > 1);
  const int gamma = i + s * d + (d < 0 ? d : 0);
  KarrasNode kn;
  kn.first = i < j ? i : j;
  kn.last = i < j ? j : i;
  kn.left = (kn.first == gamma) ? ~gamma : gamma;
  kn.right = (kn.last == gamma + 1) ? ~(gamma + 1) : (gamma + 1);
  return kn;
}

// reference of a Karras child in the collapsed tree
//   leaf i                      -> leaf ref (i,1)
//   internal c, count<=leafSize -> leaf ref (first,count)
//   internal c otherwise        -> new (compacted) index
PT_HD int collapsed_ref(int child, const int* first, const int* last, const int* newIndex, int leafSize) {
  if (child < 0) return make_leaf_ref(~child, 1);
  const int count = last[child] - first[child] + 1;
  if (count <= leafSize) return make_leaf_ref(first[child], count);
  return newIndex[child];
}

// ---- widening (binary -> four-wide) ----
// A Karras node survives the collapse when its range holds more than leafSize triangles; every ancestor of a
// surviving node survives too.  A wide node starts from a surviving node's two children and, while it has
// fewer than four, opens the surviving child with the largest surface area (replacing it by its two children,
// order kept).  Opened nodes disappear; the surviving children that are left become wide nodes themselves.
PT_HD bool karras_kept(int c, const int* first, const int* last, int leafSize) {
  return c >= 0 && last[c] - first[c] + 1 > leafSize;
}
// Karras children (>=0 internal, <0 leaf) of wide node r, left to right; opened[] receives the absorbed nodes
PT_HD int wide_children(int r, const int* left, const int* right, const int* first, const int* last, int leafSize,
                        const float* ilo, const float* ihi, int out[4], int opened[2]) {
  int n = 2, nOpen = 0;
  out[0] = left[r]; out[1] = right[r]; opened[0] = -1; opened[1] = -1;
  while (n < 4) {
    int best = -1; float bestA = -1.0f;
    for (int k = 0; k < n; k++) {
      const int c = out[k];
      if (!karras_kept(c, first, last, leafSize)) continue;
      const float A = half_area(ilo + 3 * (size_t)c, ihi + 3 * (size_t)c);
      if (A > bestA) { bestA = A; best = k; }          // ties: the leftmost
    }
    if (best < 0) break;
    const int c = out[best];
    opened[nOpen++] = c;
    for (int k = n; k > best + 1; k--) out[k] = out[k - 1];
    out[best] = left[c]; out[best + 1] = right[c];
    n++;
  }
  return n;
}
// Level (1 = root) of Karras node i in the wide tree, or 0 when i does not survive or is absorbed by a wide
// node above it.  Decided by replaying the choices along the path from the root: O(depth) work per node, no
// communication between nodes, same answer on every run.  opened[2*r], opened[2*r+1] hold the nodes that r would
// absorb as a wide node (wide_children, computed once per surviving node).  Keys are 64 bits, so a path has at
// most 64 nodes.
// path: caller-provided storage for kMaxKarrasPath ints with the given stride (LDS column on the device).
constexpr int kMaxKarrasPath = 66;
PT_HD int wide_level(int i, const int* first, const int* last, const int* parentI, int leafSize, const int* opened,
                     int* path, int stride) {
  if (!karras_kept(i, first, last, leafSize)) return 0;
  int np = 0;
  for (int p = i; p >= 0 && np < kMaxKarrasPath; p = parentI[p]) path[(np++) * stride] = p;
  int k = np - 1, level = 1;
  for (;;) {
    const int cur = path[k * stride];
    if (cur == i) return level;
    const int op0 = opened[2 * (size_t)cur], op1 = opened[2 * (size_t)cur + 1];
    k--;
    while (path[k * stride] == op0 || path[k * stride] == op1) { if (path[k * stride] == i) return 0; k--; }
    level++;
  }
}
// a traversal stack never holds more than 3 entries per level of the wide tree
PT_HD int wide_stack_bound(int wideDepth) { return 3 * wideDepth + 1; }

}  // namespace pt
