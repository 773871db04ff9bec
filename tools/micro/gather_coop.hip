// Micro-benchmark (experiment): three ways for a wave to fetch 64 random 128-byte BVH nodes, one per lane, in a
// dependent chain (the next node index comes out of the node just read):
//   A  per lane: 7 x global_load_dwordx4 of the lane's own node (what the trace kernel does)
//   B  cooperative, LDS-DMA: 8 lanes fetch the 8 quarters of one node with ONE global_load_lds_dwordx4 (4 adjacent lanes =
//      64 contiguous bytes), 8 instructions cover the wave's 64 nodes; the owner then reads its node from LDS (7 x ds_read_b128)
//   C  as B through registers: global_load_dwordx4 + ds_write_b128
//   D  quad-cooperative in registers: the 4 lanes of a quad fetch the 4 quarters of ONE 64-byte half node per instruction
//      (round r = the quad's lane r owns it; 4 rounds x 2 halves = 8 loads, each quad = one 64-byte access for the L1),
//      then a 4x4 transpose inside the quad (two DPP butterfly stages) hands every lane its own node
//   E  as D without the transpose (the load side alone)
// usage: gather_coop [tableMB] [wavesPerCU] [mode]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
struct alignas(16) f4 { float x, y, z, w; };
__device__ __forceinline__ unsigned lcg(unsigned& s) { s = 1664525u * s + 1013904223u; return s; }

// out = (lane bit `bit` set) ? keep : partner's `send`, partner = lane ^ (1 << bit) inside the quad
template <int BIT> __device__ __forceinline__ float quad_xchg(float send) {
  constexpr int ctrl = BIT == 0 ? 0xB1 : 0x4E;      // quad_perm [1,0,3,2] / [2,3,0,1]
  return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, send), ctrl, 0xf, 0xf, true));
}
// 4x4 transpose of m[r] (one dword per round r) across the 4 lanes of a quad: afterwards m[q] = what lane q held in m[lane & 3]
__device__ __forceinline__ void quad_transpose(float m[4], int lane) {
  const bool b0 = lane & 1, b1 = lane & 2;
  {  // stage 1: swap the off-diagonal elements of the 2x2 blocks (lanes ^1, registers ^1)
    const float s01 = b0 ? m[0] : m[1], s23 = b0 ? m[2] : m[3];
    const float r01 = quad_xchg<0>(s01), r23 = quad_xchg<0>(s23);
    if (b0) { m[0] = r01; m[2] = r23; } else { m[1] = r01; m[3] = r23; }
  }
  {  // stage 2: lanes ^2, registers ^2
    const float s02 = b1 ? m[0] : m[2], s13 = b1 ? m[1] : m[3];
    const float r02 = quad_xchg<1>(s02), r13 = quad_xchg<1>(s13);
    if (b1) { m[0] = r02; m[1] = r13; } else { m[2] = r02; m[3] = r13; }
  }
}

template <int MODE>
__global__ void __launch_bounds__(256) k_gather(const f4* __restrict__ recs, unsigned nRecs, int iters, float* out) {
  __shared__ f4 stage[4][64 * 8];          // per wave: 64 nodes x 8 quarters = 8 KB
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  float acc = 0.f;
  unsigned idx = lcg(s) % nRecs;
  f4* st = stage[wave];
  for (int it = 0; it < iters; it++) {
    f4 v[7];
    if (MODE == 0) {
      const f4* p = recs + (size_t)idx * 8;
#pragma unroll
      for (int q = 0; q < 7; q++) v[q] = p[q];
    } else if (MODE >= 3) {
      f4 a[2][4];                       // [half][round]: quarter (lane & 3) of the half node of the quad's lane `round`
      const int ii = __builtin_bit_cast(int, idx);
      const unsigned o4[4] = { (unsigned)__builtin_amdgcn_mov_dpp(ii, 0x00, 0xf, 0xf, true), (unsigned)__builtin_amdgcn_mov_dpp(ii, 0x55, 0xf, 0xf, true),
                               (unsigned)__builtin_amdgcn_mov_dpp(ii, 0xAA, 0xf, 0xf, true), (unsigned)__builtin_amdgcn_mov_dpp(ii, 0xFF, 0xf, 0xf, true) };   // quad_perm [r,r,r,r]
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const f4* src = recs + (size_t)o4[r] * 8 + (lane & 3);
        a[0][r] = src[0]; a[1][r] = src[4];
      }
      if (MODE == 3) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
          float mx[4] = { a[h][0].x, a[h][1].x, a[h][2].x, a[h][3].x }, my[4] = { a[h][0].y, a[h][1].y, a[h][2].y, a[h][3].y };
          float mz[4] = { a[h][0].z, a[h][1].z, a[h][2].z, a[h][3].z }, mw[4] = { a[h][0].w, a[h][1].w, a[h][2].w, a[h][3].w };
          quad_transpose(mx, lane); quad_transpose(my, lane); quad_transpose(mz, lane); quad_transpose(mw, lane);
#pragma unroll
          for (int q = 0; q < 4; q++) if (4 * h + q < 7) v[4 * h + q] = f4{ mx[q], my[q], mz[q], mw[q] };
        }
      } else {
#pragma unroll
        for (int q = 0; q < 7; q++) v[q] = a[q >> 2][q & 3];
      }
    } else {
#pragma unroll
      for (int i = 0; i < 8; i++) {
        const int owner = 8 * i + (lane >> 3);
        const unsigned oidx = __shfl(idx, owner);
        const f4* src = recs + (size_t)oidx * 8 + (lane & 7);
        if (MODE == 1) {
          __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                           (void __attribute__((address_space(3)))*)(st + i * 64), 16, 0, 0);
        } else {
          st[i * 64 + lane] = *src;
        }
      }
      if (MODE == 1) __builtin_amdgcn_s_waitcnt(0x0f70 & ~0xf);   // vmcnt(0)
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
#pragma unroll
      for (int q = 0; q < 7; q++) v[q] = st[lane * 8 + q];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    }
    float a = 0.f;
#pragma unroll
    for (int q = 0; q < 7; q++) a += (v[q].x + v[q].y) + (v[q].z + v[q].w);
    acc += a;
    idx = (lcg(s) + (unsigned)(__float_as_uint(v[6].w) & 0xff)) % nRecs;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
  const double mb = argc > 1 ? atof(argv[1]) : 3.1;
  const int wavesPerCU = argc > 2 ? atoi(argv[2]) : 12;
  const int only = argc > 3 ? atoi(argv[3]) : -1;
  hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const unsigned nRecs = (unsigned)(mb * 1e6 / 128);
  std::vector<f4> h((size_t)nRecs * 8);
  for (size_t i = 0; i < h.size(); i++) h[i] = { (float)(i & 7), 1.f, 2.f, (float)((i * 7) & 255) * 1e-30f };
  f4* d; float* out;
  const int blocks = cus * wavesPerCU / 4, iters = 2000;
  (void)hipMalloc(&d, h.size() * sizeof(f4)); (void)hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
  (void)hipMemcpy(d, h.data(), h.size() * sizeof(f4), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int mode = 0; mode < 5; mode++) {
    if (only >= 0 && mode != only) continue;
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
      (void)hipEventRecord(e0);
      if (mode == 0) k_gather<0><<<blocks, 256>>>(d, nRecs, iters, out);
      if (mode == 1) k_gather<1><<<blocks, 256>>>(d, nRecs, iters, out);
      if (mode == 2) k_gather<2><<<blocks, 256>>>(d, nRecs, iters, out);
      if (mode == 3) k_gather<3><<<blocks, 256>>>(d, nRecs, iters, out);
      if (mode == 4) k_gather<4><<<blocks, 256>>>(d, nRecs, iters, out);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best) best = ms;
    }
    const double recs = (double)blocks * 256 * iters;
    printf("mode %c table %.1f MB waves/CU %d: %.3f ms  %.1f Gnodes/s  %.2f TB/s (128 B per node)  %.0f ns per dependent step\n", "ABCDE"[mode], mb, wavesPerCU, best,
           recs / best / 1e6, recs * 128 / best / 1e9, best * 1e6 / iters);
  }
  printf("%s\n", hipGetErrorString(hipGetLastError()));
  {  // modes A and D must read the same data: compare the per-thread sums
    std::vector<float> ra((size_t)blocks * 256), rd((size_t)blocks * 256);
    k_gather<0><<<blocks, 256>>>(d, nRecs, 50, out); (void)hipMemcpy(ra.data(), out, ra.size() * 4, hipMemcpyDeviceToHost);
    k_gather<3><<<blocks, 256>>>(d, nRecs, 50, out); (void)hipMemcpy(rd.data(), out, rd.size() * 4, hipMemcpyDeviceToHost);
    size_t bad = 0; for (size_t i = 0; i < ra.size(); i++) bad += ra[i] != rd[i];
    printf("transpose check: %zu of %zu threads differ between modes A and D\n", bad, ra.size());
  }
  return 0;
}
