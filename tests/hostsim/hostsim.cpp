// hostsim.cpp -- TEST INFRASTRUCTURE.  Compiles the megakernel's per-lane code
// (minimaloptix_amd/csrc/pt_*.h) for the host and runs it one pixel at a time, plus a
// sequential mirror of the device LBVH build (same pt_lbvh.h primitives).  Used
//   * here (no GPU) to check the flattened state machine + LBVH traversal against the
//     recursive CPU oracle before any GPU time is spent, and
//   * on the GPU box to compare the device-built BVH with this mirror word for word.
// It is not part of the product: nothing under minimaloptix_amd/ builds or loads it.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <vector>
#include <functional>
#include <omp.h>
#include "../../minimaloptix_amd/csrc/pt_path.h"
#include "../../minimaloptix_amd/csrc/pt_packet.h"
#include "../../minimaloptix_amd/csrc/pt_lbvh.h"
#include "../../minimaloptix_amd/csrc/pt_upload.h"

using namespace pt;

extern "C" {

struct hostsim_scene {
  moptix_params params;
  int32_t nMaterials; const moptix_material* materials;
  int32_t nSpheres; const moptix_sphere_params* spheres; const int32_t* sphereMat;
  int32_t nQuads; const moptix_quad_params* quads; const int32_t* quadMat;
  int32_t nLights; const moptix_light_params* lights;
  int32_t nFaces;
  const float* facePos;      // 9 floats per face: p0 p1 p2
  const float* faceNrm;      // 9 floats per face (ignored where faceHasNrm == 0); may be NULL
  const int32_t* faceHasNrm; // may be NULL
  const int32_t* faceMat;
  const float* faceUV;       // 6 floats per face (u0 v0 u1 v1 u2 v2); may be NULL
  const int32_t* faceHasUV;  // may be NULL
  int32_t nTextures; const int32_t* texSize;   // width,height per texture
  const float* const* texels;                  // nTextures pointers to 4*w*h floats
};

struct hostsim_bvh_out {      // caller-allocated: nodes >= max(1,nFaces-1)*128 B, tris nFaces*48 B
  void* nodes; void* tris; int32_t* triPrim;
  int32_t nNodes, rootRef, depth;
  void* nodes64;                // may be NULL: the same nodes compressed, nNodes*64 B
};

}  // extern "C"

namespace {

struct HostBVH {
  std::vector<Node128> nodes; std::vector<Node64> nodes64; std::vector<Tri48> tris; std::vector<TriShade> shade;
  int rootRef = kEmptyRef; int depth = 0;
};

static int subtree_depth(const std::vector<Node128>& nodes, int ref) {
  if (ref < 0) return 0;
  // iterative to be safe on deep LBVHs
  int best = 0;
  std::vector<std::pair<int, int>> st; st.push_back({ ref, 1 });
  while (!st.empty()) {
    auto [r, d] = st.back(); st.pop_back();
    best = std::max(best, d);
    for (int k = 0; k < nodes[r].count; k++)
      if (nodes[r].ref[k] >= 0) st.push_back({ nodes[r].ref[k], d + 1 });
  }
  return best;
}

// Host mirror of the device's binned-SAH topology (lbvh.hip k_sah_level / k_sah_finalize; shared functions and the
// definition of the algorithm: pt_lbvh.h).  order: Morton order on entry, final order on return; nodes: n-1 entries.
struct SahTask { int node, first, count; v3 cbLo, cbHi; };
static int g_forcedRootSplit = 0;     // device: SahTask::force of the root task
static void build_sah_topology(const std::vector<v3>& lo, const std::vector<v3>& hi, std::vector<int>& order, v3 cbLo, v3 cbHi,
                               int leafSize, std::vector<KarrasNode>& nodes, std::vector<int>& parentI, std::vector<int>& parentL) {
  const int n = (int)order.size();
  nodes.assign(n - 1, KarrasNode{ 0, 0, 0, 0 }); parentI.assign(n - 1, -1); parentL.assign(n, -1);
  std::vector<SahTask> tasks{ SahTask{ 0, 0, n, cbLo, cbHi } }, next;
  int idBase = 0, level = 0;
  std::vector<int> tmp(n);
  while (!tasks.empty()) {
    next.clear();
    if (getenv("HOSTSIM_DEBUG")) fprintf(stderr, "[hostsim] SAH level %d: %zu nodes\n", level, tasks.size());
    const int nextBase = idBase + (int)tasks.size();
    for (const SahTask& t : tasks) {
      SahSplit sp; sp.axis = -1; sp.bin = 0; sp.nLeft = (t.count + 1) / 2;
      // HOSTSIM_SWEEP=1 -- a YARDSTICK, not the device's builder: the exact surface-area heuristic, every one of the
      // 3 x (count - 1) object splits of the range sorted by centroid evaluated, instead of the 3 x (kSahBins - 1) planes of the binned form.
      // HOSTSIM_SWEEP=2 adds the leaf-cost term (a split is taken only if cheaper than the range as one leaf is NOT used here: the array
      // form needs the full topology, and leaves are collapsed by size afterwards as on the device).
      static const int sweepMode = getenv("HOSTSIM_SWEEP") ? atoi(getenv("HOSTSIM_SWEEP")) : 0;
      const bool forced = level == 0 && g_forcedRootSplit > 0;
      if (forced) { sp.axis = -1; sp.nLeft = g_forcedRootSplit; }
      const bool swept = !forced && sweepMode && t.count > leafSize && level < kSahLevels;
      if (swept) {
        float bestCost = 3.0e38f; int bestAxis = -1, bestK = 0;
        std::vector<int> idx(t.count), bestOrder;
        std::vector<float> ra(t.count);
        for (int a = 0; a < 3; a++) {
          for (int i = 0; i < t.count; i++) idx[i] = order[t.first + i];
          auto cenA = [&](int f) { const v3 c = (lo[f] + hi[f]) * 0.5f; return a == 0 ? c.x : a == 1 ? c.y : c.z; };
          std::stable_sort(idx.begin(), idx.end(), [&](int x, int y) { return cenA(x) < cenA(y); });
          v3 bl = mk3(1e37f, 1e37f, 1e37f), bh = mk3(-1e37f, -1e37f, -1e37f);
          auto area = [](v3 l, v3 h) { const float dx = h.x - l.x, dy = h.y - l.y, dz = h.z - l.z; return dx * dy + dy * dz + dz * dx; };
          for (int i = t.count - 1; i >= 1; i--) {
            const int f = idx[i];
            bl = mk3(fminf_(bl.x, lo[f].x), fminf_(bl.y, lo[f].y), fminf_(bl.z, lo[f].z)); bh = mk3(fmaxf_(bh.x, hi[f].x), fmaxf_(bh.y, hi[f].y), fmaxf_(bh.z, hi[f].z));
            ra[i] = area(bl, bh);
          }
          bl = mk3(1e37f, 1e37f, 1e37f); bh = mk3(-1e37f, -1e37f, -1e37f);
          for (int k = 1; k < t.count; k++) {           // left = idx[0 .. k-1]
            const int f = idx[k - 1];
            bl = mk3(fminf_(bl.x, lo[f].x), fminf_(bl.y, lo[f].y), fminf_(bl.z, lo[f].z)); bh = mk3(fmaxf_(bh.x, hi[f].x), fmaxf_(bh.y, hi[f].y), fmaxf_(bh.z, hi[f].z));
            const float cost = area(bl, bh) * (float)k + ra[k] * (float)(t.count - k);
            if (cost < bestCost) { bestCost = cost; bestAxis = a; bestK = k; }
          }
          if (bestAxis == a) bestOrder = idx;
        }
        if (bestAxis >= 0) {
          for (int i = 0; i < t.count; i++) order[t.first + i] = bestOrder[i];
          sp.axis = -1; sp.nLeft = bestK;                // "split the (re-ordered) range at bestK"
        }
      }
      const float scale[3] = { sah_scale(t.cbLo.x, t.cbHi.x), sah_scale(t.cbLo.y, t.cbHi.y), sah_scale(t.cbLo.z, t.cbHi.z) };
      const float base[3] = { t.cbLo.x, t.cbLo.y, t.cbLo.z };
      auto cen = [&](int f, int a) { const v3 c = (lo[f] + hi[f]) * 0.5f; return a == 0 ? c.x : a == 1 ? c.y : c.z; };
      if (!swept && !forced && t.count > leafSize && level < kSahLevels) {
        SahBins B;
        for (int a = 0; a < 3; a++) for (int b = 0; b < kSahBins; b++) { B.cnt[a][b] = 0; for (int k = 0; k < 3; k++) { B.lo[a][b][k] = float_to_ordered(1e37f); B.hi[a][b][k] = float_to_ordered(-1e37f); } }
        for (int i = 0; i < t.count; i++) {
          const int f = order[t.first + i];
          const float l[3] = { lo[f].x, lo[f].y, lo[f].z }, h[3] = { hi[f].x, hi[f].y, hi[f].z };
          for (int a = 0; a < 3; a++) {
            const int b = sah_bin(cen(f, a), base[a], scale[a]);
            B.cnt[a][b]++;
            for (int k = 0; k < 3; k++) { B.lo[a][b][k] = std::min(B.lo[a][b][k], float_to_ordered(l[k])); B.hi[a][b][k] = std::max(B.hi[a][b][k], float_to_ordered(h[k])); }
          }
        }
        sp = sah_choose(B, t.cbLo, t.cbHi, t.count);
      }
      // stable partition of the range + centroid boxes of the two halves
      int nl = 0, nr = 0;
      v3 cl[2] = { mk3(1e37f, 1e37f, 1e37f), mk3(1e37f, 1e37f, 1e37f) }, ch[2] = { mk3(-1e37f, -1e37f, -1e37f), mk3(-1e37f, -1e37f, -1e37f) };
      for (int i = 0; i < t.count; i++) {
        const int f = order[t.first + i];
        const bool left = sp.axis < 0 ? (i < sp.nLeft) : (sah_bin(cen(f, sp.axis), base[sp.axis], scale[sp.axis]) < sp.bin);
        const int side = left ? 0 : 1;
        tmp[t.first + (left ? nl++ : sp.nLeft + nr++)] = f;
        const v3 c = (lo[f] + hi[f]) * 0.5f;
        cl[side] = mk3(fminf_(cl[side].x, c.x), fminf_(cl[side].y, c.y), fminf_(cl[side].z, c.z));
        ch[side] = mk3(fmaxf_(ch[side].x, c.x), fmaxf_(ch[side].y, c.y), fmaxf_(ch[side].z, c.z));
      }
      for (int i = 0; i < t.count; i++) order[t.first + i] = tmp[t.first + i];
      KarrasNode& kn = nodes[t.node];
      kn.first = t.first; kn.last = t.first + t.count - 1;
      const int cf[2] = { t.first, t.first + sp.nLeft }, cc[2] = { sp.nLeft, t.count - sp.nLeft };
      for (int side = 0; side < 2; side++) {
        int ref;
        if (cc[side] == 1) { ref = ~cf[side]; parentL[cf[side]] = t.node; }
        else { ref = nextBase + (int)next.size(); parentI[ref] = t.node; next.push_back(SahTask{ ref, cf[side], cc[side], cl[side], ch[side] }); }
        (side == 0 ? kn.left : kn.right) = ref;
      }
    }
    idBase = nextBase;
    tasks.swap(next);
    level++;
  }
}

static int g_debugPixel = getenv("HOSTSIM_DEBUG_PIXEL") ? atoi(getenv("HOSTSIM_DEBUG_PIXEL")) : -1;   // every ray of this pixel to stderr
static int g_packet = 0;      // 1 = the per-bounce state machine of pt_packet.h (kernel variant 4) instead of the per-ray one
static int g_builder = 1;     // 0 = Morton radix tree (Karras), 1 = binned SAH over the Morton order (device default)

// HOSTSIM_ESC=<k> -- a YARDSTICK (tools/tree_yardstick.py), not the device's builder: early split clipping.  A triangle whose box is large for what
// it holds -- a needle lying diagonally -- is referenced by several boxes, each the bounds of the triangle clipped to one half of the previous box
// (longest axis, midpoint), until a box's surface area is at most k times the triangle's own area x 2 (a flat axis-aligned triangle has ratio ~1)
// or 2^6 pieces; the tree is then built over REFERENCES.  Same triangle records (a triangle met twice ties with itself and is ignored).
static void clip_bounds(const v3 tri[3], v3 blo, v3 bhi, v3& lo, v3& hi) {
  // Sutherland-Hodgman against the six planes of the box, in double
  double poly[16][3], tmp[16][3]; int np = 3;
  for (int i = 0; i < 3; i++) { poly[i][0] = tri[i].x; poly[i][1] = tri[i].y; poly[i][2] = tri[i].z; }
  const double bl[3] = { blo.x, blo.y, blo.z }, bh[3] = { bhi.x, bhi.y, bhi.z };
  for (int a = 0; a < 3 && np > 0; a++)
    for (int side = 0; side < 2 && np > 0; side++) {
      const double plane = side ? bh[a] : bl[a]; const double sgn = side ? -1.0 : 1.0;
      int nt = 0;
      for (int i = 0; i < np; i++) {
        const double* A = poly[i]; const double* B = poly[(i + 1) % np];
        const double da = sgn * (A[a] - plane), db = sgn * (B[a] - plane);
        if (da >= 0) { for (int k = 0; k < 3; k++) tmp[nt][k] = A[k]; nt++; }
        if ((da >= 0) != (db >= 0)) { const double t = da / (da - db); for (int k = 0; k < 3; k++) tmp[nt][k] = A[k] + t * (B[k] - A[k]); tmp[nt][a] = plane; nt++; }
      }
      np = nt; for (int i = 0; i < np; i++) for (int k = 0; k < 3; k++) poly[i][k] = tmp[i][k];
    }
  if (np == 0) { lo = mk3(1e37f, 1e37f, 1e37f); hi = mk3(-1e37f, -1e37f, -1e37f); return; }
  double l[3] = { 1e300, 1e300, 1e300 }, h[3] = { -1e300, -1e300, -1e300 };
  for (int i = 0; i < np; i++) for (int k = 0; k < 3; k++) { l[k] = std::min(l[k], poly[i][k]); h[k] = std::max(h[k], poly[i][k]); }
  // outward in float, and never outside the box it was clipped to
  lo = mk3(std::max(nextafterf((float)l[0], -1e37f), blo.x), std::max(nextafterf((float)l[1], -1e37f), blo.y), std::max(nextafterf((float)l[2], -1e37f), blo.z));
  hi = mk3(std::min(nextafterf((float)h[0], 1e37f), bhi.x), std::min(nextafterf((float)h[1], 1e37f), bhi.y), std::min(nextafterf((float)h[2], 1e37f), bhi.z));
}
static void esc_split(const v3 tri[3], v3 lo, v3 hi, double triArea2, double k, int depth, int face, std::vector<int>& refFace, std::vector<v3>& rlo, std::vector<v3>& rhi) {
  const double dx = hi.x - lo.x, dy = hi.y - lo.y, dz = hi.z - lo.z;
  const double boxHalfArea = dx * dy + dy * dz + dz * dx;
  if (depth >= 6 || boxHalfArea <= k * triArea2 || !(boxHalfArea > 0)) { refFace.push_back(face); rlo.push_back(lo); rhi.push_back(hi); return; }
  const int a = dx >= dy && dx >= dz ? 0 : (dy >= dz ? 1 : 2);
  const float mid = a == 0 ? 0.5f * (lo.x + hi.x) : a == 1 ? 0.5f * (lo.y + hi.y) : 0.5f * (lo.z + hi.z);
  v3 hiL = hi, loR = lo;
  if (a == 0) { hiL.x = mid; loR.x = mid; } else if (a == 1) { hiL.y = mid; loR.y = mid; } else { hiL.z = mid; loR.z = mid; }
  v3 l0, h0, l1, h1;
  clip_bounds(tri, lo, hiL, l0, h0); clip_bounds(tri, loR, hi, l1, h1);
  if (l0.x <= h0.x) esc_split(tri, l0, h0, triArea2, k, depth + 1, face, refFace, rlo, rhi);
  if (l1.x <= h1.x) esc_split(tri, l1, h1, triArea2, k, depth + 1, face, refFace, rlo, rhi);
}

static void build_lbvh(const hostsim_scene& s, int leafSize, HostBVH& out) {
  out.nodes.clear(); out.nodes64.clear(); out.tris.clear(); out.shade.clear(); out.rootRef = kEmptyRef; out.depth = 0;
  if (s.nFaces <= 0) return;
  // references: one per face, or (HOSTSIM_ESC) several clipped boxes of one face
  std::vector<int> refFace; std::vector<v3> lo, hi;
  const double esc = getenv("HOSTSIM_ESC") ? atof(getenv("HOSTSIM_ESC")) : 0.0;
  for (int f = 0; f < s.nFaces; f++) {
    const float* p = s.facePos + 9 * (size_t)f;
    const v3 tri[3] = { mk3(p[0], p[1], p[2]), mk3(p[3], p[4], p[5]), mk3(p[6], p[7], p[8]) };
    v3 l, h; tri_bounds(tri[0], tri[1], tri[2], l, h);
    if (esc > 0.0) {
      const v3 e0 = tri[1] - tri[0], e1 = tri[2] - tri[0]; const v3 c = cross(e0, e1);
      const double area2 = std::sqrt((double)c.x * c.x + (double)c.y * c.y + (double)c.z * c.z);      // 2 x the triangle's area
      esc_split(tri, l, h, area2, esc, 0, f, refFace, lo, hi);
    } else { refFace.push_back(f); lo.push_back(l); hi.push_back(h); }
  }
  const int n = (int)refFace.size();
  if (esc > 0.0 && getenv("HOSTSIM_DEBUG")) fprintf(stderr, "[hostsim] ESC %.2f: %d references for %d faces\n", esc, n, s.nFaces);
  std::vector<v3> cen(n);
  v3 clo = mk3(1e37f, 1e37f, 1e37f), chi = mk3(-1e37f, -1e37f, -1e37f);
  v3 slo = clo, shi = chi;
  for (int f = 0; f < n; f++) {
    cen[f] = (lo[f] + hi[f]) * 0.5f;
    clo = mk3(fminf_(clo.x, cen[f].x), fminf_(clo.y, cen[f].y), fminf_(clo.z, cen[f].z));
    chi = mk3(fmaxf_(chi.x, cen[f].x), fmaxf_(chi.y, cen[f].y), fmaxf_(chi.z, cen[f].z));
    slo = mk3(fminf_(slo.x, lo[f].x), fminf_(slo.y, lo[f].y), fminf_(slo.z, lo[f].z));
    shi = mk3(fmaxf_(shi.x, hi[f].x), fmaxf_(shi.y, hi[f].y), fmaxf_(shi.z, hi[f].z));
  }
  const v3 invExt = mk3(inv_extent(clo.x, chi.x), inv_extent(clo.y, chi.y), inv_extent(clo.z, chi.z));
  const float padAbs = 1e-5f * fmaxf_(fmaxf_(shi.x - slo.x, shi.y - slo.y), shi.z - slo.z) + 1e-30f;
  std::vector<uint64_t> keys(n);
  const int idxBits = lbvh_index_bits(n), bitsPerAxis = getenv("HOSTSIM_MORTON30") ? 10 : lbvh_bits_per_axis(n);
  for (int f = 0; f < n; f++) keys[f] = morton_key(cen[f], clo, invExt, bitsPerAxis, idxBits, f) | (tri_is_big(lo[f], hi[f], slo, shi) ? 0ull : kSmallKeyBit);
  std::sort(keys.begin(), keys.end());
  std::vector<KarrasNode> sahNodes; std::vector<int> sahParentI, sahParentL;
  const bool useSah = g_builder == 1 && n > 1;
  if (useSah) {
    std::vector<int> order(n);
    for (int k = 0; k < n; k++) order[k] = key_face(keys[k], idxBits);
    g_forcedRootSplit = big_key_count(keys.data(), n);      // large triangles first (pt_lbvh.h kSmallKeyBit): the root's range is split between them and the others
    build_sah_topology(lo, hi, order, clo, chi, leafSize, sahNodes, sahParentI, sahParentL);
    for (int k = 0; k < n; k++) keys[k] = (uint64_t)order[k];              // the device keeps the face index only, too
  }

  out.tris.resize(n); out.shade.resize(n);
  std::vector<v3> llo(n), lhi(n);
  for (int k = 0; k < n; k++) {
    const int r = key_face(keys[k], idxBits), f = refFace[r];
    const float* p = s.facePos + 9 * (size_t)f;
    const v3 p0 = mk3(p[0], p[1], p[2]), p1 = mk3(p[3], p[4], p[5]), p2 = mk3(p[6], p[7], p[8]);
    Tri48 t; memset(&t, 0, sizeof(t));
    t.p0 = p0; t.e0 = p1 - p0; t.e1 = p0 - p2; t.mat = s.faceMat[f]; t.prim = f;
    { const DevMaterial dm = make_dev_material(s.materials[s.faceMat[f]]); t.shadow = shadow_class(dm.kind, dm.brdfType); }
    out.tris[k] = t;
    TriShade sh; memset(&sh, 0, sizeof(sh));
    if (s.faceNrm && s.faceHasNrm && s.faceHasNrm[f]) {
      const float* q = s.faceNrm + 9 * (size_t)f;
      sh.n0 = mk3(q[0], q[1], q[2]); sh.n1 = mk3(q[3], q[4], q[5]); sh.n2 = mk3(q[6], q[7], q[8]); sh.hasNormals = 1;
    }
    out.shade[k] = sh;
    llo[k] = mk3(pad_lo(lo[r].x, padAbs), pad_lo(lo[r].y, padAbs), pad_lo(lo[r].z, padAbs));
    lhi[k] = mk3(pad_hi(hi[r].x, padAbs), pad_hi(hi[r].y, padAbs), pad_hi(hi[r].z, padAbs));
  }
  if (n <= leafSize) { out.rootRef = make_leaf_ref(0, n); return; }

  const int ni = n - 1;
  std::vector<KarrasNode> kn(ni);
  std::vector<int> first(ni), last(ni);
  for (int i = 0; i < ni; i++) { kn[i] = useSah ? sahNodes[i] : karras_node(keys.data(), n, i); first[i] = kn[i].first; last[i] = kn[i].last; }
  // boxes of every Karras node = union of the leaf boxes in its range (what the device's
  // bottom-up atomic pass produces; min/max are exact so the order does not matter)
  std::vector<v3> ilo(ni), ihi(ni);
  {
    // process nodes by increasing range size so children are ready before parents
    std::vector<int> order(ni);
    for (int i = 0; i < ni; i++) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return (last[a] - first[a]) < (last[b] - first[b]); });
    auto clo_of = [&](int c) { return c < 0 ? llo[~c] : ilo[c]; };
    auto chi_of = [&](int c) { return c < 0 ? lhi[~c] : ihi[c]; };
    for (int i : order) {
      const v3 a = clo_of(kn[i].left), b = clo_of(kn[i].right), c = chi_of(kn[i].left), d = chi_of(kn[i].right);
      ilo[i] = mk3(fminf_(a.x, b.x), fminf_(a.y, b.y), fminf_(a.z, b.z));
      ihi[i] = mk3(fmaxf_(c.x, d.x), fmaxf_(c.y, d.y), fmaxf_(c.z, d.z));
    }
  }
  // widening: survivors of the collapse at even depth become four-wide nodes (pt_lbvh.h)
  std::vector<int> left(ni), right(ni), parentI(ni, -1);
  for (int i = 0; i < ni; i++) { left[i] = kn[i].left; right[i] = kn[i].right; }
  for (int i = 0; i < ni; i++) { if (left[i] >= 0) parentI[left[i]] = i; if (right[i] >= 0) parentI[right[i]] = i; }
  std::vector<int> newIndex(ni, -1);
  int nKept = 0;
  const float* iloF = reinterpret_cast<const float*>(ilo.data());
  const float* ihiF = reinterpret_cast<const float*>(ihi.data());
  static_assert(sizeof(v3) == 12, "v3 arrays are read as packed floats");
  std::vector<int> openedBy(2 * (size_t)ni, -1);
  for (int i = 0; i < ni; i++) {
    int ch[4], op[2] = { -1, -1 };
    if (karras_kept(i, first.data(), last.data(), leafSize))
      wide_children(i, left.data(), right.data(), first.data(), last.data(), leafSize, iloF, ihiF, ch, op);
    openedBy[2 * (size_t)i] = op[0]; openedBy[2 * (size_t)i + 1] = op[1];
  }
  int pathBuf[kMaxKarrasPath];
  for (int i = 0; i < ni; i++)
    if (wide_level(i, first.data(), last.data(), parentI.data(), leafSize, openedBy.data(), pathBuf, 1) > 0) newIndex[i] = nKept++;
  out.nodes.resize(nKept);
  auto box_of = [&](int child, v3& blo, v3& bhi) {
    if (child < 0) { blo = llo[~child]; bhi = lhi[~child]; } else { blo = ilo[child]; bhi = ihi[child]; }
  };
  for (int i = 0; i < ni; i++) {
    if (newIndex[i] < 0) continue;
    Node128 nd; memset(&nd, 0, sizeof(nd));
    int ch[4], opened[2];
    const int nc = wide_children(i, left.data(), right.data(), first.data(), last.data(), leafSize, iloF, ihiF, ch, opened);
    float lo4[3][4] = { { 0 } }, hi4[3][4] = { { 0 } };
    for (int k = 0; k < 4; k++) {
      if (k < nc) {
        v3 bl, bh; box_of(ch[k], bl, bh);
        lo4[0][k] = bl.x; lo4[1][k] = bl.y; lo4[2][k] = bl.z; hi4[0][k] = bh.x; hi4[1][k] = bh.y; hi4[2][k] = bh.z;
        nd.ref[k] = collapsed_ref(ch[k], first.data(), last.data(), newIndex.data(), leafSize);
      } else nd.ref[k] = kEmptyRef;
    }
    nd.lox = mk4(lo4[0][0], lo4[0][1], lo4[0][2], lo4[0][3]); nd.loy = mk4(lo4[1][0], lo4[1][1], lo4[1][2], lo4[1][3]);
    nd.loz = mk4(lo4[2][0], lo4[2][1], lo4[2][2], lo4[2][3]);
    nd.hix = mk4(hi4[0][0], hi4[0][1], hi4[0][2], hi4[0][3]); nd.hiy = mk4(hi4[1][0], hi4[1][1], hi4[1][2], hi4[1][3]);
    nd.hiz = mk4(hi4[2][0], hi4[2][1], hi4[2][2], hi4[2][3]);
    nd.count = nc;
    out.nodes[newIndex[i]] = nd;
  }
  out.rootRef = 0;
  out.depth = subtree_depth(out.nodes, 0);
  out.nodes64.resize(out.nodes.size());
  for (size_t i = 0; i < out.nodes.size(); i++)
    if (!compress_node(out.nodes[i], out.nodes64[i])) { out.nodes64.clear(); break; }      // too wide for the grid: no 64-byte form (as lbvh.hip)
}

static int g_node64 = 0;        // hostsim_set_node_format: 1 = walk the 64-byte nodes, as the packet kernel does by default

struct LocalStack {
  int data[256];
  inline void store(int sp, int v) { data[sp] = v; }
  inline int load(int sp) const { return data[sp]; }
  inline bool roomy(int) const { return true; }
  inline void store_fast(int sp, int v) { data[sp] = v; }
  static constexpr bool kFlat = false;      // pt_path.h node_step_nearfar: this stack takes the branched tail
  inline bool fits_fast(int, int) const { return false; }
  inline int peek_fast(int) const { return 0; }
};

static inline void host_trav_step(const SceneView& sc, const PathState& ps, Trav& tv, LocalStack& st, Counters& ct) {
  if (g_node64 && sc.nodes64) trav_step<true, true>(sc, ps, tv, st, ct); else trav_step<true, false>(sc, ps, tv, st, ct);
}

struct HostScene {
  std::vector<DevMaterial> mats; std::vector<DevSphere> spheres; std::vector<int> sphereMat;
  std::vector<DevQuad> quads; std::vector<DevLight> lights; HostBVH bvh; SceneView view;
  std::vector<TriUV> faceUV; std::vector<DevTexture> textures;
};

static void make_scene(const hostsim_scene& s, int leafSize, HostScene& hs) {
  for (int i = 0; i < s.nMaterials; i++) hs.mats.push_back(make_dev_material(s.materials[i]));
  for (int i = 0; i < s.nSpheres; i++) { hs.spheres.push_back(make_dev_sphere(s.spheres[i])); hs.sphereMat.push_back(s.sphereMat[i]); }
  for (int i = 0; i < s.nQuads; i++) hs.quads.push_back(make_dev_quad(s.quads[i], s.quadMat[i]));
  for (int i = 0; i < s.nLights; i++) hs.lights.push_back(make_dev_light(s.lights[i]));
  build_lbvh(s, leafSize, hs.bvh);
  SceneView& v = hs.view;
  memset(&v, 0, sizeof(v));
  v.width = (int)s.params.width; v.height = (int)s.params.height;
  v.maxDepth = (int)s.params.rayMaxDepth; v.minIntensity = s.params.rayMinIntensity; v.epsT = s.params.rayEpsilonT;
  v.bg = to_v3(s.params.bgColor); v.cam = make_cam(s.params.cam);
  v.nSpheres = s.nSpheres; v.spheres = hs.spheres.data(); v.sphereMat = hs.sphereMat.data();
  v.nQuads = s.nQuads; v.quads = hs.quads.data();
  v.nLights = s.nLights; v.lights = hs.lights.data();
  v.nMaterials = s.nMaterials; v.mats = hs.mats.data();
  v.shadowNearest = 0;
  for (const DevMaterial& m : hs.mats) if (m.kind == MAT_DISNEY && m.brdfType == BRDF_GLASS) v.shadowNearest = 1;
  v.anyDisneyAnalytic = 0;
  for (int i = 0; i < s.nSpheres; i++) if (hs.mats[s.sphereMat[i]].kind == MAT_DISNEY) v.anyDisneyAnalytic = 1;
  for (int i = 0; i < s.nQuads; i++) if (hs.mats[s.quadMat[i]].kind == MAT_DISNEY) v.anyDisneyAnalytic = 1;
  v.nTris = s.nFaces; v.rootRef = hs.bvh.rootRef;
  v.nodes = hs.bvh.nodes.data(); v.nodes64 = hs.bvh.nodes64.empty() ? nullptr : hs.bvh.nodes64.data(); v.tris = hs.bvh.tris.data(); v.triShade = hs.bvh.shade.data();
  bool anyUV = false;
  for (int f = 0; f < s.nFaces; f++) {
    TriUV uv; memset(&uv, 0, sizeof(uv));
    if (s.faceUV && s.faceHasUV && s.faceHasUV[f]) {
      const float* q = s.faceUV + 6 * (size_t)f;
      uv.u0 = q[0]; uv.v0 = q[1]; uv.u1 = q[2]; uv.v1 = q[3]; uv.u2 = q[4]; uv.v2 = q[5]; uv.hasUV = 1; anyUV = true;
    }
    hs.faceUV.push_back(uv);
  }
  v.triUV = anyUV ? hs.faceUV.data() : nullptr;
  for (int t = 0; t < s.nTextures; t++)
    hs.textures.push_back(DevTexture{ reinterpret_cast<const v4*>(s.texels[t]), s.texSize[2 * t], s.texSize[2 * t + 1] });
  v.nTextures = s.nTextures; v.textures = hs.textures.data();
}

}  // namespace

extern "C" {

void hostsim_set_builder(int builder) { g_builder = builder; }
void hostsim_set_node_format(int bytes) { g_node64 = bytes == 64; }
void hostsim_set_packet(int packet) { g_packet = packet; }

int hostsim_build_bvh(const hostsim_scene* s, int leafSize, hostsim_bvh_out* out) {
  HostBVH b; build_lbvh(*s, leafSize, b);
  out->nNodes = (int)b.nodes.size(); out->rootRef = b.rootRef; out->depth = b.depth;
  if (out->nodes && !b.nodes.empty()) memcpy(out->nodes, b.nodes.data(), b.nodes.size() * sizeof(Node128));
  if (out->nodes64 && !b.nodes64.empty()) memcpy(out->nodes64, b.nodes64.data(), b.nodes64.size() * sizeof(Node64));
  if (out->tris && !b.tris.empty()) memcpy(out->tris, b.tris.data(), b.tris.size() * sizeof(Tri48));
  if (out->triPrim) for (size_t i = 0; i < b.tris.size(); i++) out->triPrim[i] = b.tris[i].prim;
  return 0;
}


// Debugging aid: where does a ray lose a triangle?  Prints, for every node on the way from the root to the leaf that holds
// face `prim`, the slab test of the child that leads there (with tbest = tmax), then the triangle test itself.
extern "C" int hostsim_debug_ray(const hostsim_scene* s, int leafSize, const float o[3], const float d[3], float tmin, int prim) {
  HostScene hs; make_scene(*s, leafSize, hs);
  const HostBVH& b = hs.bvh;
  if (prim < 0) { prim = b.tris[-prim].prim; fprintf(stderr, "[hostsim] triangle record %d is face %d\n", -prim, prim); return prim; }   // record index -> face
  std::vector<std::pair<int, int>> path, cur;      // (node, child)
  bool found = false;
  std::function<void(int)> walk = [&](int ref) {
    if (found || ref == kEmptyRef) return;
    if (ref < 0) {
      for (int k = 0; k < leaf_count(ref); k++) if (b.tris[leaf_first(ref) + k].prim == prim) { found = true; path = cur; }
      return;
    }
    for (int c = 0; c < 4 && !found; c++) { cur.push_back({ ref, c }); walk(b.nodes[ref].ref[c]); cur.pop_back(); }
  };
  walk(b.rootRef);
  if (!found) { fprintf(stderr, "[hostsim] face %d is in no leaf\n", prim); return 1; }
  const v3 ro = mk3(o[0], o[1], o[2]), rd = mk3(d[0], d[1], d[2]);
  const v3 inv = mk3(slab_inv(rd.x), slab_inv(rd.y), slab_inv(rd.z)), noi = neg_o_inv(ro, inv);
  for (auto& pc : path) {
    const Node128& n = b.nodes[pc.first]; const int c = pc.second;
    const float* lo[3] = { &n.lox.x, &n.loy.x, &n.loz.x }; const float* hi[3] = { &n.hix.x, &n.hiy.x, &n.hiz.x };
    const float iv[3] = { inv.x, inv.y, inv.z }, ni[3] = { noi.x, noi.y, noi.z };
    float tn = tmin, tf = 3e38f;
    fprintf(stderr, "[hostsim] node %d child %d ref %d:", pc.first, c, n.ref[c]);
    for (int a = 0; a < 3; a++) {
      const float t0 = fma_(lo[a][c], iv[a], ni[a]), t1 = fma_(hi[a][c], iv[a], ni[a]);
      tn = fmaxf_(tn, fminf_(t0, t1)); tf = fminf_(tf, fmaxf_(t0, t1));
      fprintf(stderr, "  [%.9g %.9g] t %.9g %.9g", lo[a][c], hi[a][c], t0, t1);
    }
    fprintf(stderr, "  -> tn %.9g tf %.9g %s\n", tn, tf, tn <= tf * 1.0000005f ? "entered" : "CULLED");
  }
  for (const Tri48& tr : b.tris) if (tr.prim == prim) {
    v3 n; float t, be, ga;
    const bool hit = tri_test(ro, rd, tmin, 1e27f, tr.p0, tr.e0, tr.e1, n, t, be, ga);
    const v3 p1 = tr.p0 + tr.e0, p2 = tr.p0 - tr.e1;
    fprintf(stderr, "[hostsim] face %d: p0 %.9g %.9g %.9g p1 %.9g %.9g %.9g p2 %.9g %.9g %.9g\n  float: %s t %.9g beta %.9g gamma %.9g n.d %.9g\n", prim, tr.p0.x, tr.p0.y, tr.p0.z,
            p1.x, p1.y, p1.z, p2.x, p2.y, p2.z, hit ? "HIT" : "miss", t, be, ga, dot(n, rd));
    // the same formulas in double
    const double O[3] = { ro.x, ro.y, ro.z }, D[3] = { rd.x, rd.y, rd.z }, P0[3] = { tr.p0.x, tr.p0.y, tr.p0.z }, E0[3] = { tr.e0.x, tr.e0.y, tr.e0.z }, E1[3] = { tr.e1.x, tr.e1.y, tr.e1.z };
    auto crs = [](const double* a, const double* b2, double* c) { c[0] = a[1] * b2[2] - a[2] * b2[1]; c[1] = a[2] * b2[0] - a[0] * b2[2]; c[2] = a[0] * b2[1] - a[1] * b2[0]; };
    auto dt = [](const double* a, const double* b2) { return a[0] * b2[0] + a[1] * b2[1] + a[2] * b2[2]; };
    double N[3]; crs(E1, E0, N);
    const double nd = dt(N, D); double E2[3] = { (P0[0] - O[0]) / nd, (P0[1] - O[1]) / nd, (P0[2] - O[2]) / nd }, I[3]; crs(D, E2, I);
    fprintf(stderr, "  double: t %.12g beta %.12g gamma %.12g n.d %.12g\n", dt(N, E2), dt(I, E1), dt(I, E0), nd);
  }
  return 0;
}

// counters: samples, primary, bounce, shadow, nodeFetches, triTests, closestHits, lightLoads, analyticTests
// timing (may be NULL): [0] seconds of scene set-up + LBVH build (single thread), [1] seconds of rendering (all OpenMP
// threads), [2] the number of threads used -- bench.py's cpu_baseline: "a CPU build of the same megakernel", BVH build
// excluded from the rate and reported separately.
int hostsim_render_timed(const hostsim_scene* s, int leafSize, const int32_t* seeds, int nSeeds, float* accum, uint64_t counters[9], double timing[3]) {
  const double t0 = omp_get_wtime();
  HostScene hs; make_scene(*s, leafSize, hs);
  const double t1 = omp_get_wtime();
  const SceneView& sc = hs.view;
  uint64_t tot[9] = { 0 };
  const int nPix = sc.width * sc.height;
#pragma omp parallel
  {
    uint64_t loc[9] = { 0 };
#pragma omp for schedule(dynamic, 64)
    for (int pix = 0; pix < nPix; pix++) {
      PathState ps; memset(&ps, 0, sizeof(ps));
      Trav tv; memset(&tv, 0, sizeof(tv));
      Counters ct; memset(&ct, 0, sizeof(ct));
      LocalStack st;
      ps.pixel = pix; ps.item = 0;
      v3 acc = mk3(accum[3 * pix], accum[3 * pix + 1], accum[3 * pix + 2]);
      int sIdx = 0;
      Packet pk; packet_primary(pk);
      v3 att[kPacketShadows] = { mk3(1.f, 1.f, 1.f), mk3(1.f, 1.f, 1.f), mk3(1.f, 1.f, 1.f) };
      if (nSeeds > 0) begin_sample<true>(sc, ps, seeds[0], ct); else ps.mode = M_DONE;
      while (ps.mode != M_DONE) {
        if (ps.mode == M_NEW_SAMPLE) {            // Camera.cu:41, one launch after the other
          acc = acc + ps.accum;
          if (++sIdx >= nSeeds) break;
          begin_sample<true>(sc, ps, seeds[sIdx], ct);
          packet_primary(pk);
        } else if (ps.mode == M_TRACE && g_packet) {
          // the packet's rays, each on its own: shadow rays in light order, then the continuation
          for (int i = 0; i < pk.nShadow; i++) {
            PathState r = ps; Trav ts; memset(&ts, 0, sizeof(ts));
            r.d = pk.sd[i]; r.tmin = sc.epsT; r.tmax = pk.stmax[i]; r.kind = RK_SHADOW;
            trav_begin<true>(sc, r, ts, ct);
            while (ts.node != kTravDone) host_trav_step(sc, r, ts, st, ct);
            att[i] = ts.att;
          }
          if (pk.hasBounce) {
            ps.kind = RK_RADIANCE;
            trav_begin<true>(sc, ps, tv, ct);
            while (tv.node != kTravDone) host_trav_step(sc, ps, tv, st, ct);
          }
          ps.mode = M_RESULT;
        } else if (ps.mode == M_TRACE) {
          trav_begin<true>(sc, ps, tv, ct);
          while (tv.node != kTravDone) host_trav_step(sc, ps, tv, st, ct);
          if (pix == g_debugPixel)
            fprintf(stderr, "[hostsim] pixel %d depth %d kind %d o %.9g %.9g %.9g d %.9g %.9g %.9g tmin %.9g tmax %.9g -> t %.9g prim %d tri %d att %.9g %.9g %.9g\n", pix, ps.depth,
                    (int)ps.kind, ps.o.x, ps.o.y, ps.o.z, ps.d.x, ps.d.y, ps.d.z, ps.tmin, ps.tmax, tv.tbest, tv.bestPrim, tv.bestTri, tv.att.x, tv.att.y, tv.att.z);
          ps.mode = M_RESULT;
        } else if (ps.mode == M_RESULT && g_packet) {
          on_result_packet<true>(sc, ps, pk, tv, att, ct, PacketSink{ pk });
        } else if (ps.mode == M_RESULT) {
          on_result<true>(sc, ps, tv, ct);
        } else if (ps.mode == M_LIGHTS) {
          on_lights<true>(sc, ps, ct);
        }
      }
      ps.accum = acc;
      accum[3 * pix] = ps.accum.x; accum[3 * pix + 1] = ps.accum.y; accum[3 * pix + 2] = ps.accum.z;
      loc[0] += ct.samples; loc[1] += ct.primaryRays; loc[2] += ct.bounceRays; loc[3] += ct.shadowRays;
      loc[4] += ct.nodeFetches; loc[5] += ct.triTests; loc[6] += ct.closestHits; loc[7] += ct.lightLoads; loc[8] += ct.analyticTests;
    }
#pragma omp critical
    for (int i = 0; i < 9; i++) tot[i] += loc[i];
  }
  if (counters) for (int i = 0; i < 9; i++) counters[i] = tot[i];
  if (timing) { timing[0] = t1 - t0; timing[1] = omp_get_wtime() - t1; timing[2] = (double)omp_get_max_threads(); }
  return 0;
}

int hostsim_render(const hostsim_scene* s, int leafSize, const int32_t* seeds, int nSeeds, float* accum, uint64_t counters[9]) {
  return hostsim_render_timed(s, leafSize, seeds, nSeeds, accum, counters, nullptr);
}

}  // extern "C"
