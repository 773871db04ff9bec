// Micro-benchmark: the issue ceiling of the vector ALUs on MI355X, by instruction and by waves per SIMD.
//
//   valu_issue [iters]        prints one line per (instruction, waves per SIMD)
//
// The trace kernel (csrc/packetkernel.hip) is priced in wave-level vector instructions per second; this program measures
// what the chip delivers for streams of INDEPENDENT vector instructions (16 accumulators per lane, so no stream waits
// for a result) at 1, 2, 3, 4 and 8 waves per SIMD, 256-thread workgroups (one wave per SIMD each), every CU busy.
// Waves per SIMD are fixed by the dynamic LDS a workgroup asks for (160 KB / n), so n workgroups sit on every CU.
//
// Per line: wall ms (best of 3, HIP events), chip-wide G wave-instructions/s, cycles per wave-instruction per SIMD =
// SIMDs x shader clock / that rate, and the shader clock the chip held in the loop (delta s_memtime / delta s_memrealtime x
// 100 MHz, mean over waves: the chip lowers its clock under dense vector work).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

// id, name, asm text.  Operands: %0 accumulator (read + written), %1 %2 two more vector registers, %3 an SGPR pair.
#define KINDS32(X) \
  X(0, "v_fma_f32", "v_fma_f32 %0, %0, %1, %2") \
  X(1, "v_add_f32", "v_add_f32 %0, %0, %1") \
  X(2, "v_mul_f32", "v_mul_f32 %0, %0, %1") \
  X(3, "v_fmac_f32", "v_fmac_f32 %0, %1, %2") \
  X(4, "v_min_f32", "v_min_f32 %0, %0, %1") \
  X(5, "v_max_f32", "v_max_f32 %0, %0, %1") \
  X(6, "v_max3_f32", "v_max3_f32 %0, %0, %1, %2") \
  X(7, "v_min3_f32", "v_min3_f32 %0, %0, %1, %2") \
  X(8, "v_med3_f32", "v_med3_f32 %0, %0, %1, %2") \
  X(9, "v_cvt_f32_ubyte0", "v_cvt_f32_ubyte0 %0, %0") \
  X(10, "v_cvt_f32_ubyte3", "v_cvt_f32_ubyte3 %0, %0") \
  X(11, "v_cvt_f32_u32", "v_cvt_f32_u32 %0, %0") \
  X(12, "v_cvt_u32_f32", "v_cvt_u32_f32 %0, %0") \
  X(13, "v_mov_b32", "v_mov_b32 %0, %1") \
  X(14, "v_add_u32", "v_add_u32 %0, %0, %1") \
  X(15, "v_and_b32", "v_and_b32 %0, %0, %1") \
  X(16, "v_or_b32", "v_or_b32 %0, %0, %1") \
  X(17, "v_xor_b32", "v_xor_b32 %0, %0, %1") \
  X(18, "v_lshrrev_b32", "v_lshrrev_b32 %0, 3, %0") \
  X(19, "v_lshlrev_b32", "v_lshlrev_b32 %0, 3, %0") \
  X(20, "v_bfe_u32", "v_bfe_u32 %0, %0, 8, 8") \
  X(21, "v_and_or_b32", "v_and_or_b32 %0, %0, %1, %2") \
  X(22, "v_lshl_or_b32", "v_lshl_or_b32 %0, %0, 3, %2") \
  X(23, "v_lshl_add_u32", "v_lshl_add_u32 %0, %0, 3, %2") \
  X(24, "v_add3_u32", "v_add3_u32 %0, %0, %1, %2") \
  X(25, "v_perm_b32", "v_perm_b32 %0, %0, %1, %2") \
  X(26, "v_bfi_b32", "v_bfi_b32 %0, %0, %1, %2") \
  X(27, "v_min_u32", "v_min_u32 %0, %0, %1") \
  X(28, "v_max_u32", "v_max_u32 %0, %0, %1") \
  X(29, "v_min3_u32", "v_min3_u32 %0, %0, %1, %2") \
  X(30, "v_mad_u32_u24", "v_mad_u32_u24 %0, %0, %1, %2") \
  X(31, "v_mul_lo_u32", "v_mul_lo_u32 %0, %0, %1") \
  X(32, "v_cndmask_b32(vcc)", "v_cndmask_b32 %0, %0, %1, vcc") \
  X(33, "v_cndmask_b32(other_dst)", "v_cndmask_b32 %0, %1, %2, vcc") \
  X(34, "v_cndmask_b32_e64(sgpr)", "v_cndmask_b32_e64 %0, %0, %1, %3") \
  X(35, "v_cmp_lt_f32(vcc)", "v_cmp_lt_f32 vcc, %0, %1") \
  X(36, "v_cmp_lt_f32_e64(sgpr)", "v_cmp_lt_f32_e64 s[20:21], %0, %1") \
  X(37, "v_cmp+v_cndmask(vcc)", "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc") \
  X(38, "v_cmp+2cndmask(vcc)", "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc\n v_cndmask_b32 %0, %0, %1, vcc") \
  X(39, "v_cmp_class_f32(vcc)", "v_cmp_class_f32 vcc, %0, %1") \
  X(40, "v_rcp_f32", "v_rcp_f32 %0, %0") \
  X(41, "v_sqrt_f32", "v_sqrt_f32 %0, %0") \
  X(42, "v_rsq_f32", "v_rsq_f32 %0, %0") \
  X(43, "v_mov_b32_dpp(quad_perm)", "v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") \
  X(44, "v_readlane_b32", "v_readlane_b32 s22, %0, 5") \
  X(45, "v_writelane_b32", "v_writelane_b32 %0, s22, 5") \
  X(46, "v_readfirstlane_b32", "v_readfirstlane_b32 s22, %0") \
  X(47, "v_ldexp_f32", "v_ldexp_f32 %0, %0, %1") \
  X(48, "v_frexp_mant_f32", "v_frexp_mant_f32 %0, %0") \
  X(49, "v_div_scale+fixup-ish:v_div_fmas_f32", "v_div_fmas_f32 %0, %0, %1, %2") \
  X(50, "v_div_fixup_f32", "v_div_fixup_f32 %0, %0, %1, %2") \
  X(51, "v_floor_f32", "v_floor_f32 %0, %0") \
  X(52, "v_sub_f32", "v_sub_f32 %0, %0, %1") \
  X(53, "v_fma_f32(2sgpr-free,neg)", "v_fma_f32 %0, -%0, %1, %2") \
  X(54, "v_mbcnt_lo_u32_b32", "v_mbcnt_lo_u32_b32 %0, %1, %0") \
  X(55, "v_bcnt_u32_b32", "v_bcnt_u32_b32 %0, %0, %1") \
  X(60, "cmp(vcc)+4cndmask(vcc)[5]", "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 v40, %0, %2, vcc\n v_cndmask_b32 v41, %2, %0, vcc\n v_cndmask_b32 v42, %1, %2, vcc\n v_cndmask_b32 v43, %2, %1, vcc") \
  X(61, "cmp_e64(s)+4cndmask_e64(s)[5]", "v_cmp_lt_f32_e64 s[20:21], %0, %1\n v_cndmask_b32_e64 v40, %0, %2, s[20:21]\n v_cndmask_b32_e64 v41, %2, %0, s[20:21]\n v_cndmask_b32_e64 v42, %1, %2, s[20:21]\n v_cndmask_b32_e64 v43, %2, %1, s[20:21]") \
  X(62, "v_cndmask_b32_e64(vcc)", "v_cndmask_b32_e64 %0, %0, %1, vcc") \
  X(63, "cmp(vcc)+fma+cndmask(vcc)[3]", "v_cmp_lt_f32 vcc, %0, %1\n v_fma_f32 v40, %1, %2, %2\n v_cndmask_b32 %0, %0, %2, vcc") \
  X(64, "cmp(vcc)+cndmask+fma+cndmask[4]", "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 v40, %0, %2, vcc\n v_fma_f32 v41, %1, %2, %2\n v_cndmask_b32 v42, %2, %0, vcc") \
  X(65, "fma+cndmask(vcc,stale)[2]", "v_fma_f32 %0, %0, %1, %2\n v_cndmask_b32 v40, %0, %2, vcc") \
  X(66, "cmp(vcc)+2cndmask_indep[3]", "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 v40, %0, %2, vcc\n v_cndmask_b32 v41, %2, %0, vcc") \
  X(67, "cmp_e64(s)+2cndmask_e64_indep[3]", "v_cmp_lt_f32_e64 s[20:21], %0, %1\n v_cndmask_b32_e64 v40, %0, %2, s[20:21]\n v_cndmask_b32_e64 v41, %2, %0, s[20:21]") \
  X(68, "v_cndmask_b32(vcc,literal0)", "v_cndmask_b32 %0, 0, %0, vcc") \
  X(69, "v_addc_co_u32(vcc)", "v_addc_co_u32 %0, vcc, %0, %1, vcc") \
  X(70, "v_fma_mix_f32(f16lo,f32,f32)", "v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]") \
  X(71, "v_fma_mix_f32(f16hi,f32,f32)", "v_fma_mix_f32 %0, %0, %1, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]") \
  X(72, "v_cvt_f32_f16", "v_cvt_f32_f16 %0, %0") \
  X(73, "v_ashrrev_i32", "v_ashrrev_i32 %0, 3, %0") \
  X(74, "v_sub_u32", "v_sub_u32 %0, %0, %1") \
  X(75, "v_or3_b32", "v_or3_b32 %0, %0, %1, %2") \
  X(76, "v_xad_u32", "v_xad_u32 %0, %0, %1, %2") \
  X(77, "v_mul_u32_u24", "v_mul_u32_u24 %0, %0, %1") \
  X(78, "v_sad_u32", "v_sad_u32 %0, %0, %1, %2") \
  X(79, "v_cvt_f32_i32", "v_cvt_f32_i32 %0, %0") \
  X(80, "v_pk_max_f16", "v_pk_max_f16 %0, %0, %1") \
  X(81, "v_pk_fma_f16", "v_pk_fma_f16 %0, %0, %1, %2") \
  X(82, "v_mul_legacy_f32", "v_mul_legacy_f32 %0, %0, %1") \
  X(83, "v_cmp_le_f32_e64(sgpr)+s_and", "v_cmp_le_f32_e64 s[20:21], %0, %1\n s_and_b64 s[22:23], s[20:21], s[20:21]")
// 64-bit accumulators (register pairs)
#define KINDS64(X) \
  X(100, "v_pk_fma_f32", "v_pk_fma_f32 %0, %0, %1, %2") \
  X(101, "v_pk_mul_f32", "v_pk_mul_f32 %0, %0, %1") \
  X(102, "v_pk_add_f32", "v_pk_add_f32 %0, %0, %1") \
  X(103, "v_pk_mov_b32", "v_pk_mov_b32 %0, %1, %2") \
  X(104, "v_min_f64", "v_min_f64 %0, %0, %1") \
  X(105, "v_max_f64", "v_max_f64 %0, %0, %1") \
  X(106, "v_fma_f64", "v_fma_f64 %0, %0, %1, %2") \
  X(107, "v_add_f64", "v_add_f64 %0, %0, %1") \
  X(108, "v_lshrrev_b64", "v_lshrrev_b64 %0, 3, %0") \
  X(109, "v_cmp_lt_f64(vcc)", "v_cmp_lt_f64 vcc, %0, %1")

template <int KIND> struct Op;
// No clobber lists: with "vcc" or SGPRs named as clobbered the compiler's hazard recogniser puts an s_nop between any two
// statements (it has to assume a VALU write of an SGPR followed by a VALU read).  The statements that do write vcc / s[20:22]
// are the only users of those registers in the kernel (the loop counter compares through scc), so nothing is lost; the build
// is checked for s_nop in the loops (tools/micro/check_valu_issue.sh).
#define DEF32(id, name, text) \
  template <> struct Op<id> { static constexpr bool wide = false; \
    static __device__ __forceinline__ void run(float& a, float b, float c, unsigned long long m) { \
      if constexpr (id == 34) asm volatile(text : "+v"(a) : "v"(b), "v"(c), "s"(m)); else asm volatile(text : "+v"(a) : "v"(b), "v"(c)); } };
#define DEF64(id, name, text) \
  template <> struct Op<id> { static constexpr bool wide = true; \
    static __device__ __forceinline__ void run(double& a, double b, double c, unsigned long long m) { asm volatile(text : "+v"(a) : "v"(b), "v"(c)); } };
KINDS32(DEF32)
KINDS64(DEF64)

template <int KIND>
__global__ void __launch_bounds__(256) k_valu(int iters, float seedf, unsigned long long mask, unsigned long long* stamps, float* out) {
  extern __shared__ char lds[];
  float a[16]; double p[8];
#pragma unroll
  for (int i = 0; i < 16; i++) a[i] = seedf + (float)(threadIdx.x + i) * 1e-3f;
#pragma unroll
  for (int i = 0; i < 8; i++) p[i] = (double)a[i] * 1.0000001;
  const float b = 0.99999f, c = 1e-7f;
  const double pb = 0.99999, pc = 1e-7;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int rep = 0; rep < 4; rep++) {
#pragma unroll
      for (int i = 0; i < 16; i++) {
        if constexpr (Op<KIND>::wide) Op<KIND>::run(p[i & 7], pb, pc, mask);
        else Op<KIND>::run(a[i], b, c, mask);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; i++) s += a[i];
#pragma unroll
  for (int i = 0; i < 8; i++) s += (float)p[i];
  if (s == 12345.678f) out[0] = s + (float)lds[threadIdx.x];      // keeps the accumulators (and the LDS request) alive
  if ((threadIdx.x & 63) == 0) {
    const size_t w = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    stamps[2 * w] = t1 - t0; stamps[2 * w + 1] = r1 - r0;
  }
}

typedef void (*kern_t)(int, float, unsigned long long, unsigned long long*, float*);
struct Entry { const char* name; kern_t fn; int perOp; };
#define ENT(id, name, text) { name, k_valu<id>, 1 },
static Entry entries[] = { KINDS32(ENT) KINDS64(ENT) };

int main(int argc, char** argv) {
  hipDeviceProp_t prop; (void)hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const int iters = argc > 1 ? atoi(argv[1]) : 10000;       // x 64 asm statements per iteration
  const char* only = argc > 2 ? argv[2] : nullptr;          // run only the rows whose name contains this
  const int maxWaves = 8;
  unsigned long long* dStamps; float* dOut;
  (void)hipMalloc(&dStamps, sizeof(unsigned long long) * 2 * (size_t)cus * maxWaves * 4); (void)hipMalloc(&dOut, 64);
  std::vector<unsigned long long> h(2 * (size_t)cus * maxWaves * 4);
  printf("# %s, %d CUs, %d SIMDs; independent vector instructions (16 accumulators per lane), %d x 64 statements per wave; clock rate reported %d MHz\n",
         prop.gcnArchName, cus, cus * 4, iters, prop.clockRate / 1000);
  printf("# a statement is ONE instruction except the v_cmp+... rows (2 and 3): their rate is statements/s\n");
  printf("# instruction waves_per_SIMD ms G_statements_per_s cycles_per_statement_per_SIMD shader_clock_GHz\n");
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (const Entry& en : entries) {
    if (only && !strstr(en.name, only)) continue;
    for (int w : { 1, 2, 3, 4, 8 }) {
      const int blocks = cus * w;
      const size_t ldsBytes = (size_t)(160 * 1024 / w) - 1024;      // n workgroups per CU: each asks for 1/n of the CU's LDS
      (void)hipFuncSetAttribute((const void*)en.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
      float best = 1e30f;
      for (int rep = 0; rep < 3; rep++) {
        (void)hipEventRecord(e0);
        en.fn<<<blocks, 256, ldsBytes>>>(iters, 1.0f + rep, 0x5555aaaa3333ccccull, dStamps, dOut);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      (void)hipMemcpy(h.data(), dStamps, sizeof(unsigned long long) * 2 * (size_t)blocks * 4, hipMemcpyDeviceToHost);
      double cyc = 0, real = 0;
      for (size_t i = 0; i < (size_t)blocks * 4; i++) { cyc += (double)h[2 * i]; real += (double)h[2 * i + 1]; }
      const double ghz = cyc / real * 0.1;
      const double gps = (double)blocks * 4 * (double)iters * 64 / (best * 1e-3) / 1e9;
      printf("%-36s %d %.3f %.1f %.3f %.3f\n", en.name, w, best, gps, cus * 4 * ghz / gps, ghz);
      fflush(stdout);
    }
  }
  printf("# %s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
