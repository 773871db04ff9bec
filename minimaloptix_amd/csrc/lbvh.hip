// lbvh.hip -- Morton-code LBVH build on gfx950 (replaces Acceleration("Trbvh"),
// MinimalOptiX.cpp:378,494,534).  Steps and the shared per-element functions: pt_lbvh.h.
// One thread per element in every kernel; the key sort and the index scan use rocPRIM.
#include <cstring>
#include <utility>
#include <vector>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "lbvh.h"
#include "pt_lbvh.h"

namespace pt {

namespace {

constexpr int kBlock = 256;
inline int grid_for(int n) { return (n + kBlock - 1) / kBlock; }

struct SceneBox { uint32_t cLo[3], cHi[3], sLo[3], sHi[3]; };   // ordered-uint encoded

// Atomics of thousands of waves on ONE address serialise at the memory side (about 50 ns each: k_bounds took 2.3 ms for 1.1 M
// triangles, k_sahw_scatter 0.87 ms per level).  min / max are exact and order independent, so the targets are replicated:
// workgroup b works on replica b % kReplicas and a small kernel folds the replicas into replica 0 afterwards -- same bits.
constexpr int kReplicas = 64;
__global__ void k_init_box(SceneBox* b) {        // <<<kReplicas, 64>>>
  b += blockIdx.x;
  if (threadIdx.x < 3) {
    b->cLo[threadIdx.x] = float_to_ordered(1e37f); b->cHi[threadIdx.x] = float_to_ordered(-1e37f);
    b->sLo[threadIdx.x] = float_to_ordered(1e37f); b->sHi[threadIdx.x] = float_to_ordered(-1e37f);
  }
}
__global__ void k_fold_box(SceneBox* b) {        // <<<1, 64>>>: 12 words, kReplicas each
  const int w = threadIdx.x;
  if (w >= 12) return;
  uint32_t* w0 = reinterpret_cast<uint32_t*>(b);
  const bool isMin = (w / 3) % 2 == 0;             // cLo cHi sLo sHi
  uint32_t v = w0[w];
  for (int r = 1; r < kReplicas; r++) { const uint32_t x = reinterpret_cast<uint32_t*>(b + r)[w]; v = isMin ? min(v, x) : max(v, x); }
  w0[w] = v;
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fminf_(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf_(v, __shfl_xor(v, o));
  return v;
}

// 1. per-triangle bounds (meshBBox) + scene boxes
__global__ void k_bounds(int n, const float* __restrict__ facePos, float* __restrict__ lo, float* __restrict__ hi, SceneBox* box) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  v3 l = mk3(1e37f, 1e37f, 1e37f), h = mk3(-1e37f, -1e37f, -1e37f), c = l, cM = h;
  if (f < n) {
    const float* p = facePos + 9 * (size_t)f;
    tri_bounds(mk3(p[0], p[1], p[2]), mk3(p[3], p[4], p[5]), mk3(p[6], p[7], p[8]), l, h);
    lo[3 * f] = l.x; lo[3 * f + 1] = l.y; lo[3 * f + 2] = l.z;
    hi[3 * f] = h.x; hi[3 * f + 1] = h.y; hi[3 * f + 2] = h.z;
    c = (l + h) * 0.5f; cM = c;
  }
  const float v[12] = { wave_min(c.x), wave_min(c.y), wave_min(c.z), wave_max(cM.x), wave_max(cM.y), wave_max(cM.z),
                        wave_min(l.x), wave_min(l.y), wave_min(l.z), wave_max(h.x), wave_max(h.y), wave_max(h.z) };
  if ((threadIdx.x & 63) == 0) {
    box += blockIdx.x % kReplicas;
    for (int k = 0; k < 3; k++) {
      atomicMin(&box->cLo[k], float_to_ordered(v[k]));     atomicMax(&box->cHi[k], float_to_ordered(v[3 + k]));
      atomicMin(&box->sLo[k], float_to_ordered(v[6 + k])); atomicMax(&box->sHi[k], float_to_ordered(v[9 + k]));
    }
  }
}

// 2. Morton keys
__global__ void k_morton(int n, const float* __restrict__ lo, const float* __restrict__ hi, const SceneBox* box, uint64_t* __restrict__ keys) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n) return;
  const v3 clo = mk3(ordered_to_float(box->cLo[0]), ordered_to_float(box->cLo[1]), ordered_to_float(box->cLo[2]));
  const v3 chi = mk3(ordered_to_float(box->cHi[0]), ordered_to_float(box->cHi[1]), ordered_to_float(box->cHi[2]));
  const v3 invExt = mk3(inv_extent(clo.x, chi.x), inv_extent(clo.y, chi.y), inv_extent(clo.z, chi.z));
  const v3 c = (mk3(lo[3 * f], lo[3 * f + 1], lo[3 * f + 2]) + mk3(hi[3 * f], hi[3 * f + 1], hi[3 * f + 2])) * 0.5f;
  const v3 l = mk3(lo[3 * f], lo[3 * f + 1], lo[3 * f + 2]), h = mk3(hi[3 * f], hi[3 * f + 1], hi[3 * f + 2]);
  const v3 slo = mk3(ordered_to_float(box->sLo[0]), ordered_to_float(box->sLo[1]), ordered_to_float(box->sLo[2]));
  const v3 shi = mk3(ordered_to_float(box->sHi[0]), ordered_to_float(box->sHi[1]), ordered_to_float(box->sHi[2]));
  keys[f] = morton_key(c, clo, invExt, lbvh_bits_per_axis(n), lbvh_index_bits(n), f) | (tri_is_big(l, h, slo, shi) ? 0ull : kSmallKeyBit);      // large triangles first (pt_lbvh.h)
}

// 3. records + padded leaf boxes in sorted order
__global__ void k_leaves(int n, const uint64_t* __restrict__ keys, const float* __restrict__ facePos,
                         const float* __restrict__ faceNrm, const int* __restrict__ faceHasNrm, const int* __restrict__ faceMat,
                         const float* __restrict__ lo, const float* __restrict__ hi, const SceneBox* box,
                         Tri48* __restrict__ tris, TriShade* __restrict__ shade, float* __restrict__ leafLo, float* __restrict__ leafHi) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int f = key_face(keys[k], lbvh_index_bits(n));
  const float* p = facePos + 9 * (size_t)f;
  const v3 p0 = mk3(p[0], p[1], p[2]), p1 = mk3(p[3], p[4], p[5]), p2 = mk3(p[6], p[7], p[8]);
  Tri48 t;
  t.p0 = p0; t.e0 = p1 - p0; t.e1 = p0 - p2; t.mat = faceMat[f] & ((1 << kFaceMatBits) - 1); t.prim = f; t.shadow = faceMat[f] >> kFaceMatBits;
  tris[k] = t;
  TriShade sh;
  sh.n0 = mk3(0, 0, 0); sh.n1 = sh.n0; sh.n2 = sh.n0; sh.hasNormals = 0; sh.pad1 = 0; sh.pad2 = 0;
  if (faceNrm != nullptr && faceHasNrm != nullptr && faceHasNrm[f]) {
    const float* q = faceNrm + 9 * (size_t)f;
    sh.n0 = mk3(q[0], q[1], q[2]); sh.n1 = mk3(q[3], q[4], q[5]); sh.n2 = mk3(q[6], q[7], q[8]); sh.hasNormals = 1;
  }
  shade[k] = sh;
  const float ex = ordered_to_float(box->sHi[0]) - ordered_to_float(box->sLo[0]);
  const float ey = ordered_to_float(box->sHi[1]) - ordered_to_float(box->sLo[1]);
  const float ez = ordered_to_float(box->sHi[2]) - ordered_to_float(box->sLo[2]);
  const float padAbs = 1e-5f * fmaxf_(fmaxf_(ex, ey), ez) + 1e-30f;
  for (int a = 0; a < 3; a++) { leafLo[3 * k + a] = pad_lo(lo[3 * f + a], padAbs); leafHi[3 * k + a] = pad_hi(hi[3 * f + a], padAbs); }
}

// 4. Karras radix tree
__global__ void k_karras(int n, const uint64_t* __restrict__ keys, int* __restrict__ left, int* __restrict__ right,
                         int* __restrict__ first, int* __restrict__ last, int* __restrict__ parentI, int* __restrict__ parentL) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n - 1) return;
  const KarrasNode kn = karras_node(keys, n, i);
  left[i] = kn.left; right[i] = kn.right; first[i] = kn.first; last[i] = kn.last;
  if (kn.left < 0) parentL[~kn.left] = i; else parentI[kn.left] = i;
  if (kn.right < 0) parentL[~kn.right] = i; else parentI[kn.right] = i;
  if (i == 0) parentI[0] = -1;
}

// 4'. binned-SAH topology (pt_lbvh.h), one workgroup per node of the current level
struct SahTask { int node, first, count; float cbLo[3], cbHi[3]; int force; };      // force > 0: split the range at this index (the root: large triangles | the others)
constexpr int kWideTasks = 1024;      // the per-triangle form of a level handles at most this many nodes ...
constexpr int kWideCount = 2048;      // ... while some node still holds more triangles than this
constexpr int kCbReplicas = 32;       // replicas of the children's centroid boxes (see kReplicas)

__global__ void k_sah_init(int n, const uint64_t* __restrict__ keys, int* __restrict__ order, SahTask* task0, const SceneBox* box) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) order[k] = key_face(keys[k], lbvh_index_bits(n));
  if (k == 0) {
    SahTask t; t.node = 0; t.first = 0; t.count = n; t.force = big_key_count(keys, n);
    for (int a = 0; a < 3; a++) { t.cbLo[a] = ordered_to_float(box->cLo[a]); t.cbHi[a] = ordered_to_float(box->cHi[a]); }
    *task0 = t;
  }
}

constexpr int kSahBlock = 256;
__global__ void __launch_bounds__(kSahBlock) k_sah_level(const SahTask* __restrict__ tasks, int leafSize, int useSah,
                                                         const float* __restrict__ lo, const float* __restrict__ hi,
                                                         int* order, int* tmp, int* __restrict__ first, int* __restrict__ last,
                                                         SahTask* __restrict__ children) {
  __shared__ SahBins B;
  __shared__ SahSplit sSplit;
  __shared__ uint32_t sCb[2][2][3];          // [side][lo|hi][axis], order-preserving uints
  __shared__ int sWave[2][kSahBlock / 64];
  __shared__ int sBase[2];
  const SahTask t = tasks[blockIdx.x];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float base[3] = { t.cbLo[0], t.cbLo[1], t.cbLo[2] };
  const float scale[3] = { sah_scale(t.cbLo[0], t.cbHi[0]), sah_scale(t.cbLo[1], t.cbHi[1]), sah_scale(t.cbLo[2], t.cbHi[2]) };
  const bool binned = useSah && t.count > leafSize && t.force == 0;
  if (binned) {
    for (int i = tid; i < 3 * kSahBins; i += kSahBlock) {
      const int a = i / kSahBins, b = i % kSahBins;
      B.cnt[a][b] = 0;
      for (int k = 0; k < 3; k++) { B.lo[a][b][k] = float_to_ordered(1e37f); B.hi[a][b][k] = float_to_ordered(-1e37f); }
    }
  }
  if (tid < 12) sCb[tid / 6][(tid / 3) & 1][tid % 3] = float_to_ordered(((tid / 3) & 1) ? -1e37f : 1e37f);
  if (tid == 0) { sBase[0] = 0; sBase[1] = 0; sSplit.axis = -1; sSplit.bin = 0; sSplit.nLeft = t.force > 0 ? t.force : (t.count + 1) / 2; }
  __syncthreads();
  if (binned) {
    for (int i = tid; i < t.count; i += kSahBlock) {
      const int f = order[t.first + i];
      const float l[3] = { lo[3 * f], lo[3 * f + 1], lo[3 * f + 2] }, h[3] = { hi[3 * f], hi[3 * f + 1], hi[3 * f + 2] };
      for (int a = 0; a < 3; a++) {
        const int b = sah_bin((l[a] + h[a]) * 0.5f, base[a], scale[a]);
        atomicAdd(&B.cnt[a][b], 1);
        for (int k = 0; k < 3; k++) { atomicMin(&B.lo[a][b][k], float_to_ordered(l[k])); atomicMax(&B.hi[a][b][k], float_to_ordered(h[k])); }
      }
    }
    __syncthreads();
    if (tid == 0) sSplit = sah_choose(B, mk3(t.cbLo[0], t.cbLo[1], t.cbLo[2]), mk3(t.cbHi[0], t.cbHi[1], t.cbHi[2]), t.count);
    __syncthreads();
  }
  const SahSplit sp = sSplit;
  // stable partition of the range into tmp, centroid boxes of the halves on the way
  for (int b0 = 0; b0 < t.count; b0 += kSahBlock) {
    const int i = b0 + tid;
    const bool valid = i < t.count;
    int f = 0; bool left = false;
    float c[3] = { 0.f, 0.f, 0.f };
    if (valid) {
      f = order[t.first + i];
      for (int a = 0; a < 3; a++) c[a] = (lo[3 * f + a] + hi[3 * f + a]) * 0.5f;
      left = sp.axis < 0 ? (i < sp.nLeft) : (sah_bin(c[sp.axis], base[sp.axis], scale[sp.axis]) < sp.bin);
    }
    const unsigned long long mL = __ballot(valid && left), mR = __ballot(valid && !left);
    if (lane == 0) { sWave[0][wave] = __popcll(mL); sWave[1][wave] = __popcll(mR); }
    __syncthreads();
    if (valid) {
      const int side = left ? 0 : 1;
      int off = sBase[side];
      for (int w = 0; w < wave; w++) off += sWave[side][w];
      const unsigned long long m = left ? mL : mR;
      off += __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
      tmp[t.first + (left ? 0 : sp.nLeft) + off] = f;
      for (int a = 0; a < 3; a++) { atomicMin(&sCb[side][0][a], float_to_ordered(c[a])); atomicMax(&sCb[side][1][a], float_to_ordered(c[a])); }
    }
    __syncthreads();
    if (tid == 0) for (int s2 = 0; s2 < 2; s2++) { int tot = 0; for (int w = 0; w < kSahBlock / 64; w++) tot += sWave[s2][w]; sBase[s2] += tot; }
    __syncthreads();
  }
  __threadfence_block();
  for (int i = tid; i < t.count; i += kSahBlock) order[t.first + i] = tmp[t.first + i];
  if (tid == 0) {
    first[t.node] = t.first; last[t.node] = t.first + t.count - 1;
    for (int s2 = 0; s2 < 2; s2++) {
      SahTask ch; ch.node = -1; ch.force = 0; ch.first = t.first + (s2 ? sp.nLeft : 0); ch.count = s2 ? t.count - sp.nLeft : sp.nLeft;
      for (int a = 0; a < 3; a++) { ch.cbLo[a] = ordered_to_float(sCb[s2][0][a]); ch.cbHi[a] = ordered_to_float(sCb[s2][1][a]); }
      children[2 * (size_t)blockIdx.x + s2] = ch;
    }
  }
}

// 4''. the same level step for the top of the tree, where a handful of nodes hold most of the triangles: one thread
// per triangle, bins and child centroid boxes by global atomics (exact min / max / counts: same result as the
// workgroup-per-node form), the stable partition from one prefix sum over the "goes left" flags.
__device__ __forceinline__ int sah_task_of(const SahTask* __restrict__ tasks, int nTasks, int pos) {   // tasks are sorted by range
  int lo = 0, hi = nTasks - 1;
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (tasks[mid].first <= pos) lo = mid; else hi = mid - 1; }
  const SahTask& t = tasks[lo];
  return (pos >= t.first && pos < t.first + t.count) ? lo : -1;
}
__global__ void k_sahw_clear(int nTasks, SahBins* bins, uint32_t* childCb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nTasks * 3 * kSahBins) {
    const int t = i / (3 * kSahBins), r = i % (3 * kSahBins), a = r / kSahBins, b = r % kSahBins;
    bins[t].cnt[a][b] = 0;
    for (int k = 0; k < 3; k++) { bins[t].lo[a][b][k] = float_to_ordered(1e37f); bins[t].hi[a][b][k] = float_to_ordered(-1e37f); }
  }
  // [replica][task][side][lo|hi][axis]
  for (int r = 0; r < kCbReplicas; r++) if (i < nTasks * 12) childCb[(size_t)r * 12 * kWideTasks + i] = float_to_ordered(((i / 3) & 1) ? -1e37f : 1e37f);
}
// folds the replicas of the children's centroid boxes into replica 0 (one thread per word)
__global__ void k_sahw_fold(int nTasks, uint32_t* childCb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nTasks * 12) return;
  const bool isMin = ((i / 3) & 1) == 0;
  uint32_t v = childCb[i];
  for (int r = 1; r < kCbReplicas; r++) { const uint32_t x = childCb[(size_t)r * 12 * kWideTasks + i]; v = isMin ? min(v, x) : max(v, x); }
  childCb[i] = v;
}
__global__ void k_sahw_bin(int n, const SahTask* __restrict__ tasks, int nTasks, int leafSize, const float* __restrict__ lo,
                           const float* __restrict__ hi, const int* __restrict__ order, SahBins* bins) {
  // A workgroup whose 256 positions lie in ONE node (almost all of them, at the top of the tree) bins into LDS and
  // flushes 3 x 16 bins once; a workgroup across a range boundary goes to the global bins directly.
  __shared__ SahBins B;
  const int p0 = blockIdx.x * blockDim.x, p = p0 + threadIdx.x;
  const int tFirst = sah_task_of(tasks, nTasks, p0), tLast = sah_task_of(tasks, nTasks, min(n, p0 + (int)blockDim.x) - 1);
  const bool uniform = tFirst >= 0 && tFirst == tLast;
  if (uniform) {
    for (int i = threadIdx.x; i < 3 * kSahBins; i += blockDim.x) {
      const int a = i / kSahBins, b = i % kSahBins;
      B.cnt[a][b] = 0;
      for (int k = 0; k < 3; k++) { B.lo[a][b][k] = float_to_ordered(1e37f); B.hi[a][b][k] = float_to_ordered(-1e37f); }
    }
    __syncthreads();
  }
  const int ti = uniform ? tFirst : (p < n ? sah_task_of(tasks, nTasks, p) : -1);
  if (p < n && ti >= 0 && tasks[ti].count > leafSize) {
    const SahTask t = tasks[ti];
    SahBins* dst = uniform ? &B : &bins[ti];
    const int f = order[p];
    const float l[3] = { lo[3 * f], lo[3 * f + 1], lo[3 * f + 2] }, h[3] = { hi[3 * f], hi[3 * f + 1], hi[3 * f + 2] };
    for (int a = 0; a < 3; a++) {
      const int b = sah_bin((l[a] + h[a]) * 0.5f, t.cbLo[a], sah_scale(t.cbLo[a], t.cbHi[a]));
      atomicAdd(&dst->cnt[a][b], 1);
      for (int k = 0; k < 3; k++) { atomicMin(&dst->lo[a][b][k], float_to_ordered(l[k])); atomicMax(&dst->hi[a][b][k], float_to_ordered(h[k])); }
    }
  }
  if (uniform) {
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * kSahBins; i += blockDim.x) {
      const int a = i / kSahBins, b = i % kSahBins;
      if (B.cnt[a][b] == 0) continue;
      atomicAdd(&bins[tFirst].cnt[a][b], B.cnt[a][b]);
      for (int k = 0; k < 3; k++) { atomicMin(&bins[tFirst].lo[a][b][k], B.lo[a][b][k]); atomicMax(&bins[tFirst].hi[a][b][k], B.hi[a][b][k]); }
    }
  }
}
__global__ void k_sahw_choose(const SahTask* __restrict__ tasks, int nTasks, int leafSize, int useSah, const SahBins* __restrict__ bins,
                              SahSplit* __restrict__ splits) {
  const int ti = blockIdx.x * blockDim.x + threadIdx.x;
  if (ti >= nTasks) return;
  const SahTask t = tasks[ti];
  SahSplit sp; sp.axis = -1; sp.bin = 0; sp.nLeft = t.force > 0 ? t.force : (t.count + 1) / 2;
  if (useSah && t.count > leafSize && t.force == 0) sp = sah_choose(bins[ti], mk3(t.cbLo[0], t.cbLo[1], t.cbLo[2]), mk3(t.cbHi[0], t.cbHi[1], t.cbHi[2]), t.count);
  splits[ti] = sp;
}
__device__ __forceinline__ bool sahw_left(const SahTask& t, const SahSplit& sp, int p, int f, const float* __restrict__ lo, const float* __restrict__ hi) {
  if (sp.axis < 0) return (p - t.first) < sp.nLeft;
  const float c = (lo[3 * f + sp.axis] + hi[3 * f + sp.axis]) * 0.5f;
  return sah_bin(c, t.cbLo[sp.axis], sah_scale(t.cbLo[sp.axis], t.cbHi[sp.axis])) < sp.bin;
}
__global__ void k_sahw_side(int n, const SahTask* __restrict__ tasks, int nTasks, const SahSplit* __restrict__ splits,
                            const float* __restrict__ lo, const float* __restrict__ hi, const int* __restrict__ order, int* __restrict__ leftFlag) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int ti = sah_task_of(tasks, nTasks, p);
  leftFlag[p] = (ti >= 0 && sahw_left(tasks[ti], splits[ti], p, order[p], lo, hi)) ? 1 : 0;
}
__global__ void k_sahw_scatter(int n, const SahTask* __restrict__ tasks, int nTasks, const SahSplit* __restrict__ splits,
                               const float* __restrict__ lo, const float* __restrict__ hi, const int* __restrict__ order,
                               const int* __restrict__ leftFlag, const int* __restrict__ leftScan, int* __restrict__ tmp, uint32_t* childCb) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int ti = p < n ? sah_task_of(tasks, nTasks, p) : -1;
  const bool valid = ti >= 0;
  bool left = false;
  float c[3] = { 0.f, 0.f, 0.f };
  if (valid) {
    const SahTask t = tasks[ti];
    const SahSplit sp = splits[ti];
    const int f = order[p];
    left = leftFlag[p] != 0;
    const int leftRank = leftScan[p] - leftScan[t.first];             // "goes left" among the positions before p in this range
    tmp[left ? t.first + leftRank : t.first + sp.nLeft + (p - t.first - leftRank)] = f;
    for (int a = 0; a < 3; a++) c[a] = (lo[3 * f + a] + hi[3 * f + a]) * 0.5f;
  }
  // centroid boxes of the two halves: one wave = one node almost always, so reduce in the wave and let one lane
  // do the 12 atomics (64 lanes on 12 addresses would serialise in the memory system)
  const int t0 = __builtin_amdgcn_readfirstlane(ti);
  if (__ballot(ti != t0) == 0ull) {
    if (t0 < 0) return;
    float r[12];
    for (int s2 = 0; s2 < 2; s2++)
      for (int a = 0; a < 3; a++) {
        const bool mine = valid && (left == (s2 == 0));
        r[6 * s2 + a] = wave_min(mine ? c[a] : 1e37f); r[6 * s2 + 3 + a] = wave_max(mine ? c[a] : -1e37f);
      }
    if ((threadIdx.x & 63) == 0) {
      uint32_t* cb = childCb + (size_t)(blockIdx.x % kCbReplicas) * 12 * kWideTasks + 12 * (size_t)t0;
      for (int s2 = 0; s2 < 2; s2++) for (int a = 0; a < 3; a++) { atomicMin(&cb[6 * s2 + a], float_to_ordered(r[6 * s2 + a])); atomicMax(&cb[6 * s2 + 3 + a], float_to_ordered(r[6 * s2 + 3 + a])); }
    }
  } else if (valid) {
    uint32_t* cb = childCb + (size_t)(blockIdx.x % kCbReplicas) * 12 * kWideTasks + 12 * (size_t)ti + (left ? 0 : 6);
    for (int a = 0; a < 3; a++) { atomicMin(&cb[a], float_to_ordered(c[a])); atomicMax(&cb[3 + a], float_to_ordered(c[a])); }
  }
}
__global__ void k_sahw_commit(int n, const SahTask* __restrict__ tasks, int nTasks, const SahSplit* __restrict__ splits,
                              const uint32_t* __restrict__ childCb, const int* __restrict__ tmp, int* __restrict__ order,
                              int* __restrict__ first, int* __restrict__ last, SahTask* __restrict__ children) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n && sah_task_of(tasks, nTasks, p) >= 0) order[p] = tmp[p];
  if (p < nTasks) {
    const SahTask t = tasks[p];
    const SahSplit sp = splits[p];
    first[t.node] = t.first; last[t.node] = t.first + t.count - 1;
    for (int s2 = 0; s2 < 2; s2++) {
      SahTask ch; ch.node = -1; ch.force = 0; ch.first = t.first + (s2 ? sp.nLeft : 0); ch.count = s2 ? t.count - sp.nLeft : sp.nLeft;
      for (int a = 0; a < 3; a++) { ch.cbLo[a] = ordered_to_float(childCb[12 * (size_t)p + 6 * s2 + a]); ch.cbHi[a] = ordered_to_float(childCb[12 * (size_t)p + 6 * s2 + 3 + a]); }
      children[2 * (size_t)p + s2] = ch;
    }
  }
}

__global__ void k_sah_flags(int nChildren, const SahTask* __restrict__ children, int* __restrict__ flags) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < nChildren) flags[c] = children[c].count >= 2 ? 1 : 0;
}

// children with two or more triangles become the next level's nodes, numbered in range order after this level's
__global__ void k_sah_finalize(int nChildren, const SahTask* __restrict__ tasks, const SahTask* __restrict__ children,
                               const int* __restrict__ flags, const int* __restrict__ offs, int nextBase,
                               int* __restrict__ left, int* __restrict__ right, int* __restrict__ parentI, int* __restrict__ parentL,
                               SahTask* __restrict__ nextTasks, int* nextCount) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nChildren) return;
  const int node = tasks[c >> 1].node;
  SahTask ch = children[c];
  int ref;
  if (flags[c]) { ref = nextBase + offs[c]; ch.node = ref; parentI[ref] = node; nextTasks[offs[c]] = ch; atomicMax(nextCount + 1, ch.count); }
  else { ref = ~ch.first; parentL[ch.first] = node; }
  if (c & 1) right[node] = ref; else left[node] = ref;
  if (c == nChildren - 1) *nextCount = offs[c] + flags[c];
}

__global__ void k_order_keys(int n, const int* __restrict__ order, uint64_t* __restrict__ keys) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) keys[k] = (uint64_t)(uint32_t)order[k];
}

__device__ __forceinline__ float ld_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// 5. bottom-up fit: the second thread to arrive at a node unions the children and goes on.
// Boxes of other workgroups are read/written with agent-scope accesses (they bypass the
// non-coherent per-CU L1) and ordered by the fences around the arrival counter.
__global__ void k_fit(int n, const int* __restrict__ left, const int* __restrict__ right, const int* __restrict__ parentI,
                      const int* __restrict__ parentL, const float* leafLo, const float* leafHi,
                      float* ilo, float* ihi, unsigned int* arrivals) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  int node = parentL[k];
  while (node >= 0) {
    __threadfence();
    const unsigned int prev = atomicAdd(&arrivals[node], 1u);
    if (prev == 0) return;                      // first arrival: the sibling subtree is not ready yet
    __threadfence();
    const int l = left[node], r = right[node];
    float b[12];
    for (int a = 0; a < 3; a++) {
      const float ll = l < 0 ? ld_agent(&leafLo[3 * (~l) + a]) : ld_agent(&ilo[3 * l + a]);
      const float rl = r < 0 ? ld_agent(&leafLo[3 * (~r) + a]) : ld_agent(&ilo[3 * r + a]);
      const float lh = l < 0 ? ld_agent(&leafHi[3 * (~l) + a]) : ld_agent(&ihi[3 * l + a]);
      const float rh = r < 0 ? ld_agent(&leafHi[3 * (~r) + a]) : ld_agent(&ihi[3 * r + a]);
      b[a] = fminf_(ll, rl); b[3 + a] = fmaxf_(lh, rh);
    }
    for (int a = 0; a < 3; a++) { st_agent(&ilo[3 * node + a], b[a]); st_agent(&ihi[3 * node + a], b[3 + a]); }
    node = parentI[node];
  }
}

// 5'. the same boxes for the binned-SAH topology, whose nodes were numbered level by level: one launch per level from the
// deepest up, every node unions its two children (leaves, or nodes of deeper levels that earlier launches finished).  No
// arrival counters, no fences, no agent-scope accesses: 1.1 M triangles 5.0 ms -> 0.3 ms, the same min / max bits.
__global__ void k_fit_level(int base, int count, const int* __restrict__ left, const int* __restrict__ right,
                            const float* __restrict__ leafLo, const float* __restrict__ leafHi, float* ilo, float* ihi) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const int node = base + i;
  const int l = left[node], r = right[node];
  for (int a = 0; a < 3; a++) {
    const float ll = l < 0 ? leafLo[3 * (size_t)(~l) + a] : ilo[3 * (size_t)l + a], rl = r < 0 ? leafLo[3 * (size_t)(~r) + a] : ilo[3 * (size_t)r + a];
    const float lh = l < 0 ? leafHi[3 * (size_t)(~l) + a] : ihi[3 * (size_t)l + a], rh = r < 0 ? leafHi[3 * (size_t)(~r) + a] : ihi[3 * (size_t)r + a];
    ilo[3 * (size_t)node + a] = fminf_(ll, rl); ihi[3 * (size_t)node + a] = fmaxf_(lh, rh);
  }
}

// 6a. what every surviving node would absorb if it became a four-wide node
__global__ void k_opened(int ni, const int* __restrict__ left, const int* __restrict__ right, const int* __restrict__ first,
                         const int* __restrict__ last, int leafSize, const float* __restrict__ ilo, const float* __restrict__ ihi,
                         int* __restrict__ opened) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ni) return;
  int ch[4], op[2] = { -1, -1 };
  if (karras_kept(i, first, last, leafSize)) wide_children(i, left, right, first, last, leafSize, ilo, ihi, ch, op);
  opened[2 * (size_t)i] = op[0]; opened[2 * (size_t)i + 1] = op[1];
}

// 7a. which Karras nodes become four-wide nodes, and on which level (0 = none)
__global__ void k_kept(int ni, const int* __restrict__ first, const int* __restrict__ last, const int* __restrict__ parentI,
                       int leafSize, const int* __restrict__ opened, int* __restrict__ kept, int* depthOut) {
  __shared__ int sPath[kMaxKarrasPath][64];            // one column per thread: no scratch memory, no bank conflicts
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int d = 0;
  if (i < ni) { d = wide_level(i, first, last, parentI, leafSize, opened, &sPath[0][threadIdx.x], 64); kept[i] = d > 0 ? 1 : 0; }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) d = max(d, __shfl_xor(d, o));
  if ((threadIdx.x & 63) == 0 && d > 0) atomicMax(depthOut, d);
}

// 7b. emit the compacted 128-byte nodes
__global__ void k_emit(int ni, const int* __restrict__ left, const int* __restrict__ right, const int* __restrict__ first,
                       const int* __restrict__ last, const int* __restrict__ kept, const int* __restrict__ newIndex, int leafSize,
                       const float* __restrict__ leafLo, const float* __restrict__ leafHi, const float* __restrict__ ilo,
                       const float* __restrict__ ihi, Node128* __restrict__ nodes) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ni || !kept[i]) return;
  int ch[4], opened[2];
  const int n = wide_children(i, left, right, first, last, leafSize, ilo, ihi, ch, opened);
  float lo[3][4], hi[3][4];
  Node128 nd;
  for (int k = 0; k < 4; k++) {
    if (k < n) {
      const int c = ch[k];
      const float* l = c < 0 ? leafLo + 3 * (size_t)(~c) : ilo + 3 * (size_t)c;
      const float* h = c < 0 ? leafHi + 3 * (size_t)(~c) : ihi + 3 * (size_t)c;
      for (int a = 0; a < 3; a++) { lo[a][k] = l[a]; hi[a][k] = h[a]; }
      nd.ref[k] = collapsed_ref(c, first, last, newIndex, leafSize);
    } else {
      for (int a = 0; a < 3; a++) { lo[a][k] = 0.f; hi[a][k] = 0.f; }
      nd.ref[k] = kEmptyRef;
    }
  }
  nd.lox = mk4(lo[0][0], lo[0][1], lo[0][2], lo[0][3]); nd.loy = mk4(lo[1][0], lo[1][1], lo[1][2], lo[1][3]);
  nd.loz = mk4(lo[2][0], lo[2][1], lo[2][2], lo[2][3]);
  nd.hix = mk4(hi[0][0], hi[0][1], hi[0][2], hi[0][3]); nd.hiy = mk4(hi[1][0], hi[1][1], hi[1][2], hi[1][3]);
  nd.hiz = mk4(hi[2][0], hi[2][1], hi[2][2], hi[2][3]);
  nd.count = n; nd.pad[0] = 0; nd.pad[1] = 0; nd.pad[2] = 0;
  nodes[newIndex[i]] = nd;
}

// 7c. the 64-byte form of the nodes; *bad counts nodes whose box is wider than the grid can span
__global__ void k_compress(int n, const Node128* __restrict__ nodes, Node64* __restrict__ out, int* __restrict__ bad) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Node64 c;
  if (!compress_node(nodes[i], c)) atomicAdd(bad, 1);
  out[i] = c;
}

template <class T> hipError_t dmalloc(T** p, size_t n) { return hipMalloc((void**)p, sizeof(T) * (n ? n : 1)); }

}  // namespace

void lbvh_free(LbvhResult* r) {
  if (!r) return;
  if (r->nodes) (void)hipFree(r->nodes);
  if (r->nodes64) (void)hipFree(r->nodes64);
  if (r->tris) (void)hipFree(r->tris);
  if (r->shade) (void)hipFree(r->shade);
  *r = LbvhResult();
}

#define LB_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { err = e_; goto done; } } while (0)

hipError_t lbvh_build(hipStream_t stream, const float* dFacePos, const float* dFaceNrm, const int* dFaceHasNrm,
                      const int* dFaceMat, int n, int leafSize, int builder, LbvhResult* out) {
  hipError_t err = hipSuccess;
  *out = LbvhResult();
  out->nTris = n; out->leafSize = leafSize;
  if (n <= 0) return hipSuccess;
  if (leafSize < 1) leafSize = 1;
  if (leafSize > kMaxLeaf) leafSize = kMaxLeaf;
  out->leafSize = leafSize;
  const int ni = n - 1;

  float *lo = nullptr, *hi = nullptr, *leafLo = nullptr, *leafHi = nullptr, *ilo = nullptr, *ihi = nullptr;
  uint64_t *keys = nullptr, *keysSorted = nullptr;
  int *left = nullptr, *right = nullptr, *first = nullptr, *last = nullptr, *parentI = nullptr, *parentL = nullptr, *kept = nullptr, *newIndex = nullptr, *opened = nullptr;
  unsigned int* arrivals = nullptr; SceneBox* box = nullptr; int* dDepth = nullptr; void* tmp = nullptr;
  int *order = nullptr, *orderTmp = nullptr, *sahFlags = nullptr, *sahOffs = nullptr, *sahNext = nullptr, *sahLeft = nullptr, *sahLeftScan = nullptr;
  SahBins* sahBins = nullptr; SahSplit* sahSplits = nullptr; uint32_t* sahChildCb = nullptr;
  SahTask *tasksA = nullptr, *tasksB = nullptr, *sahChildren = nullptr;
  size_t tmpSah = 0;
  size_t tmpSort = 0, tmpScan = 0, tmpBytes = 0;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  // Per-level node counts and the final sizes come back through PINNED host memory: a copy into pageable memory is staged by
  // the runtime and cost ~250 us per level of the binned-SAH loop (28 levels on coffee: 7 of the 12.9 ms round 2 reported).
  std::vector<std::pair<int, int> > levelRange;     // binned-SAH builder: (first node id, nodes) of every level
  int* pinned = nullptr;          // [0..1] next level's {active nodes, largest node}, [2..3] node counts, [4] depth

  LB_CHECK(hipEventCreate(&e0)); LB_CHECK(hipEventCreate(&e1));
  LB_CHECK(hipHostMalloc((void**)&pinned, 8 * sizeof(int), hipHostMallocDefault));
  LB_CHECK(dmalloc(&lo, 3 * (size_t)n)); LB_CHECK(dmalloc(&hi, 3 * (size_t)n));
  LB_CHECK(dmalloc(&leafLo, 3 * (size_t)n)); LB_CHECK(dmalloc(&leafHi, 3 * (size_t)n));
  LB_CHECK(dmalloc(&ilo, 3 * (size_t)ni)); LB_CHECK(dmalloc(&ihi, 3 * (size_t)ni));
  LB_CHECK(dmalloc(&keys, (size_t)n)); LB_CHECK(dmalloc(&keysSorted, (size_t)n));
  LB_CHECK(dmalloc(&left, (size_t)ni)); LB_CHECK(dmalloc(&right, (size_t)ni)); LB_CHECK(dmalloc(&first, (size_t)ni)); LB_CHECK(dmalloc(&last, (size_t)ni));
  LB_CHECK(dmalloc(&parentI, (size_t)ni)); LB_CHECK(dmalloc(&parentL, (size_t)n)); LB_CHECK(dmalloc(&kept, (size_t)ni + 1)); LB_CHECK(dmalloc(&newIndex, (size_t)ni + 1)); LB_CHECK(dmalloc(&opened, 2 * (size_t)ni + 2));
  LB_CHECK(dmalloc(&arrivals, (size_t)ni)); LB_CHECK(dmalloc(&box, (size_t)kReplicas)); LB_CHECK(dmalloc(&dDepth, 1));
  LB_CHECK(dmalloc(&out->tris, (size_t)n)); LB_CHECK(dmalloc(&out->shade, (size_t)n));
  LB_CHECK(rocprim::radix_sort_keys(nullptr, tmpSort, keys, keysSorted, (size_t)n, 0, 64, stream));
  if (ni > 0) LB_CHECK(rocprim::exclusive_scan(nullptr, tmpScan, kept, newIndex, 0, (size_t)ni, rocprim::plus<int>(), stream));
  if (builder == 1 && n > leafSize) {
    LB_CHECK(dmalloc(&order, (size_t)n)); LB_CHECK(dmalloc(&orderTmp, (size_t)n));
    LB_CHECK(dmalloc(&sahFlags, (size_t)n + 2)); LB_CHECK(dmalloc(&sahOffs, (size_t)n + 2)); LB_CHECK(dmalloc(&sahNext, 2));
    LB_CHECK(dmalloc(&sahLeft, (size_t)n + 1)); LB_CHECK(dmalloc(&sahLeftScan, (size_t)n + 1));
    LB_CHECK(dmalloc(&sahBins, (size_t)kWideTasks)); LB_CHECK(dmalloc(&sahSplits, (size_t)kWideTasks)); LB_CHECK(dmalloc(&sahChildCb, 12 * (size_t)kWideTasks * kCbReplicas));
    LB_CHECK(dmalloc(&tasksA, (size_t)n / 2 + 2)); LB_CHECK(dmalloc(&tasksB, (size_t)n / 2 + 2)); LB_CHECK(dmalloc(&sahChildren, (size_t)n + 2));
    LB_CHECK(rocprim::exclusive_scan(nullptr, tmpSah, sahFlags, sahOffs, 0, (size_t)n + 2, rocprim::plus<int>(), stream));   // >= any scan below
    if (tmpSah > tmpScan) tmpScan = tmpSah;
  }
  tmpBytes = tmpSort > tmpScan ? tmpSort : tmpScan;
  LB_CHECK(hipMalloc(&tmp, tmpBytes ? tmpBytes : 16));

  LB_CHECK(hipEventRecord(e0, stream));
  k_init_box<<<kReplicas, 64, 0, stream>>>(box);
  k_bounds<<<grid_for(n), kBlock, 0, stream>>>(n, dFacePos, lo, hi, box);
  k_fold_box<<<1, 64, 0, stream>>>(box);
  k_morton<<<grid_for(n), kBlock, 0, stream>>>(n, lo, hi, box, keys);
  LB_CHECK(rocprim::radix_sort_keys(tmp, tmpSort, keys, keysSorted, (size_t)n, 0, 64, stream));
  if (builder == 1 && n > leafSize) {
    // binned-SAH topology over the Morton order, one launch per level (pt_lbvh.h); the final order replaces the keys
    k_sah_init<<<grid_for(n), kBlock, 0, stream>>>(n, keysSorted, order, tasksA, box);
    LB_CHECK(hipMemsetAsync(parentI, 0xff, sizeof(int) * (size_t)ni, stream));          // root: parent -1
    int nActive = 1, idBase = 0, level = 0, maxCount = n;
    SahTask *cur = tasksA, *nxt = tasksB;
    while (nActive > 0) {
      levelRange.push_back(std::make_pair(idBase, nActive));
      const int nChildren = 2 * nActive;
      const int useSah = level < kSahLevels ? 1 : 0;
      LB_CHECK(hipMemsetAsync(sahNext, 0, 2 * sizeof(int), stream));
      if (nActive <= kWideTasks && maxCount > kWideCount) {
        k_sahw_clear<<<grid_for(nActive * 3 * kSahBins), kBlock, 0, stream>>>(nActive, sahBins, sahChildCb);
        if (useSah) k_sahw_bin<<<grid_for(n), kBlock, 0, stream>>>(n, cur, nActive, leafSize, lo, hi, order, sahBins);
        k_sahw_choose<<<grid_for(nActive), kBlock, 0, stream>>>(cur, nActive, leafSize, useSah, sahBins, sahSplits);
        k_sahw_side<<<grid_for(n), kBlock, 0, stream>>>(n, cur, nActive, sahSplits, lo, hi, order, sahLeft);
        size_t tb2 = tmpBytes;
        LB_CHECK(rocprim::exclusive_scan(tmp, tb2, sahLeft, sahLeftScan, 0, (size_t)n, rocprim::plus<int>(), stream));
        k_sahw_scatter<<<grid_for(n), kBlock, 0, stream>>>(n, cur, nActive, sahSplits, lo, hi, order, sahLeft, sahLeftScan, orderTmp, sahChildCb);
        k_sahw_fold<<<grid_for(nActive * 12), kBlock, 0, stream>>>(nActive, sahChildCb);
        k_sahw_commit<<<grid_for(n > nActive ? n : nActive), kBlock, 0, stream>>>(n, cur, nActive, sahSplits, sahChildCb, orderTmp, order, first, last, sahChildren);
      } else
      k_sah_level<<<nActive, kSahBlock, 0, stream>>>(cur, leafSize, useSah, lo, hi, order, orderTmp, first, last, sahChildren);
      k_sah_flags<<<grid_for(nChildren), kBlock, 0, stream>>>(nChildren, sahChildren, sahFlags);
      size_t tb = tmpBytes;
      LB_CHECK(rocprim::exclusive_scan(tmp, tb, sahFlags, sahOffs, 0, (size_t)nChildren, rocprim::plus<int>(), stream));
      k_sah_finalize<<<grid_for(nChildren), kBlock, 0, stream>>>(nChildren, cur, sahChildren, sahFlags, sahOffs, idBase + nActive,
                                                                left, right, parentI, parentL, nxt, sahNext);
      LB_CHECK(hipMemcpyAsync(pinned, sahNext, 2 * sizeof(int), hipMemcpyDeviceToHost, stream));
      LB_CHECK(hipStreamSynchronize(stream));
      idBase += nActive; nActive = pinned[0]; maxCount = pinned[1]; level++;
      SahTask* sw = cur; cur = nxt; nxt = sw;
    }
    k_order_keys<<<grid_for(n), kBlock, 0, stream>>>(n, order, keysSorted);
  }
  k_leaves<<<grid_for(n), kBlock, 0, stream>>>(n, keysSorted, dFacePos, dFaceNrm, dFaceHasNrm, dFaceMat, lo, hi, box,
                                               out->tris, out->shade, leafLo, leafHi);
  if (n <= leafSize) {
    out->rootRef = make_leaf_ref(0, n); out->nNodes = 0; out->depth = 0;
    LB_CHECK(dmalloc(&out->nodes, 1));
    LB_CHECK(dmalloc(&out->nodes64, 1));
  } else {
    LB_CHECK(hipMemsetAsync(arrivals, 0, sizeof(unsigned int) * (size_t)ni, stream));
    LB_CHECK(hipMemsetAsync(dDepth, 0, sizeof(int), stream));
    if (builder != 1) k_karras<<<grid_for(ni), kBlock, 0, stream>>>(n, keysSorted, left, right, first, last, parentI, parentL);
    if (!levelRange.empty()) {
      for (size_t L = levelRange.size(); L-- > 0;)
        k_fit_level<<<grid_for(levelRange[L].second), kBlock, 0, stream>>>(levelRange[L].first, levelRange[L].second, left, right, leafLo, leafHi, ilo, ihi);
    } else
    k_fit<<<grid_for(n), kBlock, 0, stream>>>(n, left, right, parentI, parentL, leafLo, leafHi, ilo, ihi, arrivals);
    k_opened<<<grid_for(ni), kBlock, 0, stream>>>(ni, left, right, first, last, leafSize, ilo, ihi, opened);
    k_kept<<<(ni + 63) / 64, 64, 0, stream>>>(ni, first, last, parentI, leafSize, opened, kept, dDepth);
    LB_CHECK(rocprim::exclusive_scan(tmp, tmpScan, kept, newIndex, 0, (size_t)ni, rocprim::plus<int>(), stream));
    LB_CHECK(hipMemcpyAsync(pinned + 2, newIndex + (ni - 1), sizeof(int), hipMemcpyDeviceToHost, stream));
    LB_CHECK(hipMemcpyAsync(pinned + 3, kept + (ni - 1), sizeof(int), hipMemcpyDeviceToHost, stream));
    LB_CHECK(hipMemcpyAsync(pinned + 4, dDepth, sizeof(int), hipMemcpyDeviceToHost, stream));
    LB_CHECK(hipStreamSynchronize(stream));
    out->nNodes = pinned[2] + pinned[3]; out->depth = pinned[4];
    LB_CHECK(dmalloc(&out->nodes, (size_t)out->nNodes));
    k_emit<<<grid_for(ni), kBlock, 0, stream>>>(ni, left, right, first, last, kept, newIndex, leafSize, leafLo, leafHi, ilo, ihi, out->nodes);
    LB_CHECK(dmalloc(&out->nodes64, (size_t)out->nNodes));
    LB_CHECK(hipMemsetAsync(dDepth, 0, sizeof(int), stream));
    k_compress<<<grid_for(out->nNodes), kBlock, 0, stream>>>(out->nNodes, out->nodes, out->nodes64, dDepth);
    LB_CHECK(hipMemcpyAsync(pinned + 4, dDepth, sizeof(int), hipMemcpyDeviceToHost, stream));
    out->rootRef = 0;
  }
  LB_CHECK(hipEventRecord(e1, stream));
  LB_CHECK(hipStreamSynchronize(stream));
  LB_CHECK(hipGetLastError());
  if (out->nNodes > 0 && pinned[4] != 0) {      // a node wider than 1e10 units (pt_lbvh.h compress_node): this tree has no 64-byte form
    (void)hipFree(out->nodes64); out->nodes64 = nullptr;
  }
  (void)hipEventElapsedTime(&out->buildMs, e0, e1);
  out->stackBound = wide_stack_bound(out->depth);

done:
  for (void* p : { (void*)lo, (void*)hi, (void*)leafLo, (void*)leafHi, (void*)ilo, (void*)ihi, (void*)keys, (void*)keysSorted,
                   (void*)left, (void*)right, (void*)first, (void*)last, (void*)parentI, (void*)parentL, (void*)kept, (void*)newIndex,
                   (void*)arrivals, (void*)box, (void*)dDepth, (void*)opened, tmp, (void*)order, (void*)orderTmp, (void*)sahFlags,
                   (void*)sahOffs, (void*)sahNext, (void*)tasksA, (void*)tasksB, (void*)sahChildren, (void*)sahLeft, (void*)sahLeftScan,
                   (void*)sahBins, (void*)sahSplits, (void*)sahChildCb })
    if (p) (void)hipFree(p);
  if (pinned) (void)hipHostFree(pinned);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (err != hipSuccess) lbvh_free(out);
  return err;
}

}  // namespace pt
