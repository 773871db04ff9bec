// How much LDS may a 256-thread workgroup use and still run three (four) to a CU?  Prints the occupancy the runtime
// reports for a range of dynamic LDS sizes.  Build: hipcc --offload-arch=gfx950 -O2 lds_limit.hip -o lds_limit
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ int dyn[];
__global__ void __launch_bounds__(256) k(int* out) { dyn[threadIdx.x] = threadIdx.x; __syncthreads(); out[threadIdx.x] = dyn[255 - threadIdx.x]; }
int main() {
  int last = -1;
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  for (int b = 32768; b <= 163840; b += 64) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k, 256, b) != hipSuccess) { printf("query failed at %d\n", b); break; }
    if (n != last) { printf("LDS %d B -> %d workgroups per CU\n", b, n); last = n; }
  }
  return 0;
}
