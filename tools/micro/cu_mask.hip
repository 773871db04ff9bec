// cu_mask.hip -- what a HIP stream's CU mask means on this GPU (hipExtStreamCreateWithCUMask), and whether two kernels on streams with
// complementary masks run side by side without touching each other's compute units.
//   1. which (XCC, SE, CU) the workgroups of a kernel land on: no mask, a mask without its first R bits, the complement (only those R bits)
//   2. a long "hog" kernel on the big mask and a short latency kernel on the small one: each alone, then together
// build: hipcc --offload-arch=gfx950 -O2 -o tools/micro/cu_mask tools/micro/cu_mask.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <set>
#include <chrono>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void where(unsigned int* out, int spinTicks) {
  unsigned int hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spinTicks) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
// a dependent chain of FMAs: the time of one wave's serial work, n iterations
__global__ void chain(float* out, int n) {
  float x = threadIdx.x * 1e-3f;
  for (int i = 0; i < n; i++) x = __builtin_fmaf(x, 1.0000001f, 1e-7f);
  out[blockIdx.x * blockDim.x + threadIdx.x] = x;
}
// 3 workgroups of 256 threads with 53 KB of LDS per CU, busy for `ticks` of the 100 MHz clock
__global__ void __launch_bounds__(256) hog(float* out, int ticks) {
  __shared__ float lds[13000];
  float x = threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  int it = 0;
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) {
    for (int i = 0; i < 256; i++) x = __builtin_fmaf(x, 1.0000001f, 1e-7f);
    lds[(threadIdx.x + it) % 13000] = x; it++;
  }
  out[blockIdx.x * 256 + threadIdx.x] = x + lds[threadIdx.x];
}

static void census(const char* name, hipStream_t s, unsigned int* d, int nWG) {
  std::vector<unsigned int> h(2 * nWG);
  hipLaunchKernelGGL(where, dim3(nWG), dim3(64), 0, s, d, 20000);      // 200 us each: every CU the stream may use gets workgroups
  CK(hipStreamSynchronize(s));
  CK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
  std::set<unsigned int> cus; int perXcc[16] = {0};
  for (int i = 0; i < nWG; i++) {
    const unsigned int hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
    const unsigned int cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    const unsigned int key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
    if (cus.insert(key).second) perXcc[xcc]++;
  }
  printf("%-28s distinct CUs %3zu  per XCC:", name, cus.size());
  for (int x = 0; x < 8; x++) printf(" %d", perXcc[x]);
  printf("\n");
}

int main(int argc, char** argv) {
  const int R = argc > 1 ? atoi(argv[1]) : 8;
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int nCU = p.multiProcessorCount;
  printf("%s: %d CUs; R = %d\n", p.name, nCU, R);
  const int words = (nCU + 31) / 32;
  std::vector<uint32_t> big(words, 0xffffffffu), small(words, 0u);
  for (int i = 0; i < R; i++) { big[i / 32] &= ~(1u << (i % 32)); small[i / 32] |= 1u << (i % 32); }
  hipStream_t s0, sBig, sSmall;
  CK(hipStreamCreate(&s0));
  CK(hipExtStreamCreateWithCUMask(&sBig, words, big.data()));
  CK(hipExtStreamCreateWithCUMask(&sSmall, words, small.data()));
  unsigned int* d; CK(hipMalloc(&d, 2 * 8192 * 4));
  census("no mask", s0, d, 8192);
  census("mask without bits 0..R-1", sBig, d, 8192);
  census("mask of bits 0..R-1 only", sSmall, d, 8192);
  // a strided choice: bits 0, 32, 64 ... (one per mask word)
  std::vector<uint32_t> strided(words, 0u); for (int w = 0; w < words; w++) strided[w] = 1u;
  hipStream_t sStr; CK(hipExtStreamCreateWithCUMask(&sStr, words, strided.data()));
  census("bits 0, 32, 64, ...", sStr, d, 8192);

  float* f; CK(hipMalloc(&f, 4096 * 256 * 4));
  auto ms = [&](auto fn) { CK(hipDeviceSynchronize()); const auto t0 = std::chrono::steady_clock::now(); fn(); CK(hipDeviceSynchronize());
                           return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
  const int nChain = 2000000;      // ~ 8 ms of one wave's dependent FMAs
  for (int rep = 0; rep < 2; rep++) {
    const double tc = ms([&] { hipLaunchKernelGGL(chain, dim3(R * 8), dim3(64), 0, sSmall, f, nChain); });
    const double th = ms([&] { hipLaunchKernelGGL(hog, dim3((nCU - R) * 3), dim3(256), 0, sBig, f, 2000000); });      // 20 ms
    const double tb = ms([&] { hipLaunchKernelGGL(hog, dim3((nCU - R) * 3), dim3(256), 0, sBig, f, 2000000);
                               hipLaunchKernelGGL(chain, dim3(R * 8), dim3(64), 0, sSmall, f, nChain); });
    const double tb2 = ms([&] { hipLaunchKernelGGL(chain, dim3(R * 8), dim3(64), 0, sSmall, f, nChain);
                                hipLaunchKernelGGL(hog, dim3((nCU - R) * 3), dim3(256), 0, sBig, f, 2000000); });
    // the same pair without masks: the chain's waves compete with the hog's for the same SIMDs
    const double tn = ms([&] { hipLaunchKernelGGL(hog, dim3((nCU - R) * 3), dim3(256), 0, s0, f, 2000000);
                               hipLaunchKernelGGL(chain, dim3(R * 8), dim3(64), 0, sStr, f, nChain); });
    printf("chain alone %.2f ms | hog alone %.2f ms | hog then chain, masked %.2f ms | chain then hog, masked %.2f ms | hog (no mask) + chain on strided mask %.2f ms\n", tc, th, tb, tb2, tn);
  }
  return 0;
}
