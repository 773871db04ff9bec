// Micro-benchmark: the ceiling of the traversal's access pattern on MI355X -- every lane of a wave fetches its OWN
// record (a BVH node: one 128-byte line, or half of one) from a table that lives in the XCD's L2 or in the Infinity
// Cache, and the address of the next record depends on the data of this one (a traversal's dependent chain).
//
//   gather [csv]      prints one line per (record bytes, table MB, waves per CU, chains per lane)
//
// chains = independent dependent-chains a lane keeps in flight (1 = what a ray is; 2 / 4 = what the memory system
// could deliver to the same number of waves if the fetches were independent: the bandwidth side of the ceiling).
// The rate is "useful" bytes: records x record bytes / time, the same currency as bench.py's algorithmic bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
struct alignas(16) f4 { float x, y, z, w; };
__device__ __forceinline__ unsigned lcg(unsigned& s) { s = 1664525u * s + 1013904223u; return s; }

template <int Q, int CH>     // Q = 16-byte quarters per record (4 = 64 B, 8 = 128 B), CH = chains per lane
__global__ void __launch_bounds__(256) k_gather(const f4* __restrict__ recs, unsigned nRecs, int iters, float* out) {
  unsigned s = (blockIdx.x * 256 + threadIdx.x) * 2654435761u + 12345u;
  float acc = 0.f;
  unsigned idx[CH];
#pragma unroll
  for (int c = 0; c < CH; c++) idx[c] = lcg(s) % nRecs;
  for (int it = 0; it < iters; it++) {
    f4 v[CH][Q];
#pragma unroll
    for (int c = 0; c < CH; c++) {
      const f4* p = recs + (size_t)idx[c] * Q;
#pragma unroll
      for (int q = 0; q < Q; q++) v[c][q] = p[q];
    }
#pragma unroll
    for (int c = 0; c < CH; c++) {
      float a = 0.f;
#pragma unroll
      for (int q = 0; q < Q; q++) a += v[c][q].x + v[c][q].y;
      acc += a;
      // the next index depends on the record just read (low bits of its last word) + a per-lane random walk
      idx[c] = (lcg(s) + (unsigned)(__float_as_uint(v[c][Q - 1].w) & 0xff)) % nRecs;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int Q, int CH>
static double run(const f4* d, unsigned nRecs, int blocks, int iters, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    hipEventRecord(e0);
    k_gather<Q, CH><<<blocks, 256>>>(d, nRecs, iters, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  hipEventDestroy(e0); hipEventDestroy(e1);
  return best;
}

int main(int argc, char** argv) {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const double tableMB[] = { 3.1, 8.1, 38.0, 151.0 };     // coffee's node array / its triangle records / Infinity Cache resident / larger
  const int wavesPerCU[] = { 12, 16, 32 };                // the trace kernel holds 12; 32 is the hardware maximum
  const size_t maxBytes = (size_t)(151.0 * 1e6) + 4096;
  std::vector<f4> h(maxBytes / 16);
  for (size_t i = 0; i < h.size(); i++) h[i] = { (float)(i & 7), 1.f, 2.f, (float)((i * 7) & 255) * 1e-30f };
  f4* d; float* out;
  hipMalloc(&d, h.size() * sizeof(f4)); hipMalloc(&out, (size_t)cus * 8 * 256 * sizeof(float));
  hipMemcpy(d, h.data(), h.size() * sizeof(f4), hipMemcpyHostToDevice);
  printf("# %s, %d CUs; dependent random gather, one record per lane per step; rate = records x record bytes / time\n", prop.gcnArchName, cus);
  printf("# rec_B table_MB waves_per_CU chains ms Grec_per_s TB_per_s ns_per_dependent_step\n");
  for (int recB : { 64, 128 }) {
    for (double mb : tableMB) {
      const unsigned nRecs = (unsigned)(mb * 1e6 / recB);
      for (int w : wavesPerCU) {
        const int blocks = cus * w / 4;
        for (int ch : { 1, 2, 4 }) {
          if (recB == 128 && ch == 4) continue;           // 128 registers of payload: not a shape any kernel would run
          const int iters = 4000 / ch;
          double ms = 0;
          if (recB == 64) ms = ch == 1 ? run<4, 1>(d, nRecs, blocks, iters, out) : ch == 2 ? run<4, 2>(d, nRecs, blocks, iters, out) : run<4, 4>(d, nRecs, blocks, iters, out);
          else ms = ch == 1 ? run<8, 1>(d, nRecs, blocks, iters, out) : run<8, 2>(d, nRecs, blocks, iters, out);
          const double recs = (double)blocks * 256 * iters * ch;
          printf("%d %.1f %d %d %.3f %.1f %.2f %.0f\n", recB, mb, w, ch, ms, recs / ms / 1e6, recs * recB / ms / 1e9, ms * 1e6 / iters);
          fflush(stdout);
        }
      }
    }
  }
  printf("# %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
