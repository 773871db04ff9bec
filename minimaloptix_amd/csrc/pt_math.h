// pt_math.h -- float3 arithmetic for the path-tracing megakernel (gfx950).
//
// Implements the arithmetic contract of DESIGN.md (AC1..AC7) so that the HIP
// kernels take exactly the same ray/hit/sampling decisions as the CPU oracle:
//   AC1 dot  = fma(a.z,b.z, fma(a.y,b.y, a.x*b.x))
//   AC2 cross= ( fma(a.y,b.z,-(a.z*b.y)), ... )
//   AC3 length = sqrtf(dot), normalize = v*(1/sqrtf(dot)), v/s = v*(1/s)
//   AC4 every other operator is a single IEEE binary32 op (build with -ffp-contract=off)
//   AC5 sin/cos are one specified binary32 algorithm (sincos_ac below: Cody-Waite reduction + fixed polynomials)
//   AC7 point on ray = fma(t, d, o)
// These are the optixu_math_namespace.h semantics the reference's programs rely on
// (SURVEY.md Appendix A1), with the fused forms nvcc's default -fmad=true produces.
//
// The header is plain C++ so that tests/hostsim can compile the per-lane code for the
// host and compare it with the oracle without a GPU; the product only ever runs it on
// the device.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define PT_HD __host__ __device__ __forceinline__
#define PT_D __device__ __forceinline__
#if defined(PT_OUTLINE_BRDF)
#define PT_HD_BRDF static __host__ __device__ __attribute__((noinline))
#else
#define PT_HD_BRDF PT_HD
#endif
#else
#define PT_HD inline
#define PT_D inline
#define PT_HD_BRDF inline
#endif

namespace pt {

struct v3 { float x, y, z; };
struct alignas(16) v4 { float x, y, z, w; };

PT_HD v3 mk3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_HD v3 splat3(float s) { return mk3(s, s, s); }
PT_HD v4 mk4(float x, float y, float z, float w) { v4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
PT_HD v3 xyz(const v4& a) { return mk3(a.x, a.y, a.z); }
PT_HD v3 operator+(v3 a, v3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_HD v3 operator-(v3 a, v3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_HD v3 operator*(v3 a, v3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_HD v3 operator*(v3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
PT_HD v3 operator*(float s, v3 a) { return mk3(a.x * s, a.y * s, a.z * s); }
PT_HD v3 operator+(v3 a, float s) { return mk3(a.x + s, a.y + s, a.z + s); }
PT_HD v3 operator-(v3 a) { return mk3(-a.x, -a.y, -a.z); }
PT_HD v3 operator/(v3 a, float s) { float inv = 1.0f / s; return a * inv; }   // AC3

PT_HD float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
PT_HD float dot(v3 a, v3 b) { return fma_(a.z, b.z, fma_(a.y, b.y, a.x * b.x)); }                 // AC1
PT_HD v3 cross(v3 a, v3 b) {                                                                        // AC2
  return mk3(fma_(a.y, b.z, -(a.z * b.y)), fma_(a.z, b.x, -(a.x * b.z)), fma_(a.x, b.y, -(a.y * b.x)));
}
PT_HD float length(v3 a) { return __builtin_sqrtf(dot(a, a)); }
#if defined(PT_EXPERIMENT_RSQ_ERR)
// Experiment (tests/hostsim only, never in the product build): the reference is compiled with -use_fast_math, where
// 1/sqrtf is rsqrt.approx (max relative error 2^-22.4).  This models that error as a deterministic pseudo-random
// relative perturbation of the reciprocal length, to measure how far the image moves ("coffee.png pin").
PT_HD v3 normalize(v3 a) {
  const float d = dot(a, a);
  float inv = 1.0f / __builtin_sqrtf(d);
  uint32_t h = (uint32_t)__builtin_bit_cast(int32_t, d) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  const float u = (float)(h >> 8) * (1.0f / 16777216.0f) * 2.0f - 1.0f;           // [-1, 1)
  inv = inv * (1.0f + u * (float)(PT_EXPERIMENT_RSQ_ERR));
  return a * inv;
}
#else
PT_HD v3 normalize(v3 a) { float inv = 1.0f / __builtin_sqrtf(dot(a, a)); return a * inv; }
#endif
PT_HD v3 ray_at(v3 o, v3 d, float t) { return mk3(fma_(t, d.x, o.x), fma_(t, d.y, o.y), fma_(t, d.z, o.z)); }  // AC7
PT_HD float lerp(float a, float b, float t) { return a + t * (b - a); }
PT_HD v3 lerp(v3 a, v3 b, float t) { return a + (b - a) * t; }
PT_HD float fminf_(float a, float b) { return __builtin_fminf(a, b); }
PT_HD float fmaxf_(float a, float b) { return __builtin_fmaxf(a, b); }
PT_HD float clampf(float x, float lo, float hi) { return fmaxf_(lo, fminf_(x, hi)); }
PT_HD float sqr(float x) { return x * x; }
// AC5: sin and cos as ONE specified binary32 algorithm (both sides of the parity contract evaluate exactly these
// operations): quadrant q = floor(x*2/pi + 1/2), three-step Cody-Waite reduction r = x - q*pi/2 with fma, degree-7 /
// degree-8 polynomials on |r| <= pi/4 (coefficients of the classic single-precision kernels), quadrant fix-up.
// Absolute error < 2e-7 for |x| < 100; the path tracer only calls it with x in [0, 2 pi].
PT_HD void sincos_ac(float x, float& s, float& c) {
  const float qf = __builtin_floorf(fma_(x, 0.636619772f, 0.5f));
  float r = fma_(qf, -1.5703125f, x);
  r = fma_(qf, -4.837512969970703125e-4f, r);
  r = fma_(qf, -7.54978995489188216e-8f, r);
  const float z = r * r;
  const float sp = fma_(fma_(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f);
  const float cp = fma_(fma_(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f);
  const float sr = fma_(sp * z, r, r);
  const float cr = fma_(cp * z, z, fma_(-0.5f, z, 1.0f));
  const int q = (int)qf & 3;
  s = (q == 0) ? sr : (q == 1) ? cr : (q == 2) ? -sr : -cr;
  c = (q == 0) ? cr : (q == 1) ? -sr : (q == 2) ? -cr : sr;
}

PT_HD int32_t f2i(float f) { return __builtin_bit_cast(int32_t, f); }
PT_HD float i2f(int32_t i) { return __builtin_bit_cast(float, i); }

constexpr float kPi = 3.14159265358979323846f;   // M_PIf
constexpr float kRtDefaultMax = 1e27f;           // RT_DEFAULT_MAX

// length(a) != 0 without the square root: sqrtf(x) is zero exactly when x is (and NaN for NaN, which compares unequal to zero
// either way), so the decision is the one `length(a) != 0.0f` makes (Material.cu:197) -- a correctly rounded square root is
// ~19 vector instructions here.
PT_HD bool length_is_nonzero(v3 a) { return dot(a, a) != 0.0f; }

// optixu reflect / faceforward / refract
PT_HD v3 reflect(v3 i, v3 n) { return i - (n * 2.0f) * dot(n, i); }
PT_HD v3 faceforward(v3 n, v3 i, v3 nref) { return n * __builtin_copysignf(1.0f, dot(i, nref)); }
PT_HD bool refract(v3& r, v3 i, v3 n, float ior) {
  v3 nn = n;
  float negNdotV = dot(i, nn);
  float eta;
  if (negNdotV > 0.0f) { eta = ior; nn = -n; negNdotV = -negNdotV; }
  else { eta = 1.0f / ior; }
  const float k = 1.0f - eta * eta * (1.0f - negNdotV * negNdotV);
  if (k < 0.0f) { r = mk3(0.f, 0.f, 0.f); return false; }
  r = normalize(i * eta - nn * (eta * negNdotV + __builtin_sqrtf(k)));
  return true;
}

// optix::Onb
struct Onb { v3 tangent, binormal, normal; };
PT_HD Onb make_onb(v3 n) {
  Onb o; o.normal = n;
  if (__builtin_fabsf(n.x) > __builtin_fabsf(n.z)) o.binormal = mk3(-n.y, n.x, 0.f);
  else                                             o.binormal = mk3(0.f, -n.z, n.y);
  o.binormal = normalize(o.binormal);
  o.tangent = cross(o.binormal, o.normal);
  return o;
}
PT_HD v3 onb_inverse(const Onb& o, v3 p) { return (o.tangent * p.x + o.binormal * p.y) + o.normal * p.z; }

}  // namespace pt
