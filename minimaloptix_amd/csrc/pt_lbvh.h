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
PT_HD int lbvh_bits_per_axis(int n) { const int b = (64 - lbvh_index_bits(n)) / 3; return b > 21 ? 21 : b; }
PT_HD uint64_t expand_bits21(uint64_t v) {
  v &= 0x1fffffull;
  v = (v | (v << 32)) & 0x001f00000000ffffull;
  v = (v | (v << 16)) & 0x001f0000ff0000ffull;
  v = (v | (v << 8)) & 0x100f00f00f00f00full;
  v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
  v = (v | (v << 2)) & 0x1249249249249249ull;
  return v;
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

// order-preserving float <-> uint mapping for atomicMin/atomicMax on floats
PT_HD uint32_t float_to_ordered(float f) { uint32_t u = (uint32_t)f2i(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
PT_HD float ordered_to_float(uint32_t u) { return i2f((int32_t)((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u)); }

PT_HD int clz64(uint64_t x) { return x ? __builtin_clzll(x) : 64; }
// common-prefix length of keys i and j (-1 outside the array); keys are unique
PT_HD int lcp(const uint64_t* keys, int n, int i, int j) {
  if (j < 0 || j >= n) return -1;
  return clz64(keys[i] ^ keys[j]);
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
PT_HD float half_area(const float* lo, const float* hi) {
  const float dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
  return (dx * dy + dy * dz) + dz * dx;
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
