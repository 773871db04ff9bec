// Micro-benchmark: does the L1 address unit's time for a per-lane gather scale with the ACTIVE lanes of the instruction?
//   gather_mask         one line per (pattern, active lanes): dependent random gathers of 64-byte records from a 3.1 MB table,
//                       12 waves per CU, with only some lanes of every wave taking part (the others sit out behind the exec mask).
// pattern 0: the first K lanes; pattern 1: every (64/K)-th lane.  If a masked lane costs nothing, time ~ K.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct alignas(16) f4 { float x, y, z, w; };
__device__ __forceinline__ unsigned lcg(unsigned& s) { s = 1664525u * s + 1013904223u; return s; }
__global__ void __launch_bounds__(256) k_gather(const f4* __restrict__ recs, unsigned nRecs, int iters, int K, int pattern, float* out) {
  const int lane = threadIdx.x & 63;
  const bool active = pattern == 0 ? lane < K : (lane % (64 / K)) == 0;
  unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  float acc = 0.f;
  unsigned idx = lcg(s) % nRecs;
  if (active) {
    for (int it = 0; it < iters; it++) {
      const f4* p = recs + (size_t)idx * 4;
      const f4 a = p[0], b = p[1], c = p[2], d = p[3];
      acc += a.x + b.x + c.x + d.x;
      idx = (lcg(s) + (unsigned)(__float_as_uint(d.w) & 0xff)) % nRecs;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main() {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const unsigned nRecs = (unsigned)(3.1e6 / 64);
  std::vector<f4> h((size_t)nRecs * 4);
  for (size_t i = 0; i < h.size(); i++) h[i] = { (float)(i & 7), 1.f, 2.f, (float)((i * 7) & 255) * 1e-30f };
  f4* d; float* out;
  hipMalloc(&d, h.size() * sizeof(f4)); hipMalloc(&out, (size_t)cus * 3 * 256 * sizeof(float));
  hipMemcpy(d, h.data(), h.size() * sizeof(f4), hipMemcpyHostToDevice);
  printf("# %s, %d CUs, 12 waves per CU, 64-byte records, 3.1 MB table; lanes sitting out behind the exec mask\n", prop.gcnArchName, cus);
  printf("# pattern active_lanes ms Grec_per_s lookups_per_CU_clock(2.4GHz)\n");
  const int iters = 4000, blocks = cus * 3;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int pattern = 0; pattern < 2; pattern++)
    for (int K : { 64, 48, 32, 16, 8 }) {
      if (pattern == 1 && (K == 48 || K == 64)) continue;
      float best = 1e30f;
      for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        k_gather<<<blocks, 256>>>(d, nRecs, iters, K, pattern, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
      }
      const double recs = (double)blocks * 4 * K * iters;
      printf("%d %d %.3f %.1f %.3f\n", pattern, K, best, recs / best / 1e6, recs * 4 / (best * 1e-3) / (cus * 2.4e9));
    }
  printf("# %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
