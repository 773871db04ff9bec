// Microbenchmark: how fast can a wave fetch 64 independent 64-byte records (BVH nodes)?
//  K1: each lane issues 4 x dwordx4 for its own record (what the traversal does today)
//  K2: quad-cooperative: 4 adjacent lanes fetch the 4 quarters of one record (16 records per
//      instruction, 4 instructions), data exchanged through LDS
//  K3: as K2 but with direct-to-LDS loads (global_load_lds_dwordx4)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
struct alignas(16) f4 { float x, y, z, w; };
__device__ __forceinline__ unsigned lcg(unsigned& s) { s = 1664525u * s + 1013904223u; return s; }

template <int MODE>
__global__ void __launch_bounds__(256) k_gather(const f4* __restrict__ nodes, unsigned nNodes, int iters, float* out) {
  __shared__ f4 stage[4][256];   // per wave: 64 records x 4 quarters
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  float acc = 0.f;
  unsigned idx = lcg(s) % nNodes;
  for (int it = 0; it < iters; it++) {
    f4 a, b, c, d;
    if (MODE == 1) {
      const f4* p = nodes + (size_t)idx * 4;
      a = p[0]; b = p[1]; c = p[2]; d = p[3];
    } else {
      f4* st = stage[wave];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int owner = (lane >> 2) + 16 * i;
        const unsigned oidx = __shfl(idx, owner);
        const f4* src = nodes + (size_t)oidx * 4 + (lane & 3);
        if (MODE == 2) {
          st[i * 64 + lane] = *src;                 // record `owner`, quarter lane&3 -> st[owner*4 + q]
        } else {
          __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)src,
                                           (void __attribute__((address_space(3)))*)(st + i * 64), 16, 0, 0);
        }
      }
      if (MODE == 3) __builtin_amdgcn_s_waitcnt(0x0f70 & ~0xf);   // vmcnt(0)
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      a = st[lane * 4 + 0]; b = st[lane * 4 + 1]; c = st[lane * 4 + 2]; d = st[lane * 4 + 3];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    }
    acc += a.x + b.y + c.z + d.w;
    // next index depends on the data (dependent chain like a traversal) + rng
    idx = (lcg(s) + (unsigned)(__float_as_uint(d.w) & 0xff)) % nNodes;
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main(int argc, char** argv) {
  const unsigned nNodes = argc > 1 ? atoi(argv[1]) : 56000;
  const int iters = 2000, blocks = 256 * (argc > 2 ? atoi(argv[2]) : 2);
  std::vector<f4> h(nNodes * 4);
  for (size_t i = 0; i < h.size(); i++) h[i] = { (float)(i & 7), 1.f, 2.f, (float)((i * 7) & 255) * 1e-30f };
  f4* d; float* out;
  hipMalloc(&d, h.size() * sizeof(f4)); hipMalloc(&out, blocks * 256 * sizeof(float));
  hipMemcpy(d, h.data(), h.size() * sizeof(f4), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 1; mode <= 3; mode++) {
    for (int rep = 0; rep < 2; rep++) {
      hipEventRecord(e0);
      if (mode == 1) k_gather<1><<<blocks, 256>>>(d, nNodes, iters, out);
      if (mode == 2) k_gather<2><<<blocks, 256>>>(d, nNodes, iters, out);
      if (mode == 3) k_gather<3><<<blocks, 256>>>(d, nNodes, iters, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double recs = (double)blocks * 256 * iters;
      if (rep == 1) printf("mode %d nodes %u blocks %d: %.3f ms  %.1f Grec/s  %.2f TB/s  (%.0f cycles per wave-step @2.1GHz with %d waves/CU)\n", mode, nNodes, blocks, ms,
             recs / ms / 1e6, recs * 64 / ms / 1e9, ms * 1e-3 * 2.1e9 / iters, blocks * 4 / 256);
    }
  }
  printf("%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
