// pt_disney.h -- Disney BRDF sampling / pdf / evaluation (disney.h:9-91) and the
// microfacet helpers (utils_device.h:130-185), on the pre-derived material record.
#pragma once
#include "pt_types.h"
#include "pt_rng.h"

namespace pt {

// Arithmetic of the three Disney functions.  FAST = false (default, and the only mode the parity contract knows):
// correctly rounded binary32 division and square root, as the oracle evaluates them.  FAST = true (opt-in,
// moptix_set_option("fast_shading", 1); device only): the hardware approximations v_rcp_f32 / v_sqrt_f32 / v_rsq_f32
// (1 ulp), which is what the reference's own -use_fast_math build does (utils_host.cpp:30-32).  Only disney_pdf and
// disney_eval use it -- pure weights.  Intersection, hit-point refinement, RNG, light sampling and the sampled bounce
// direction (disney_sample) stay exact, so the same rays are traced and only BRDF / pdf values move (by ~1e-6 relative).
template <bool FAST> struct ShadeMath {
#if defined(__HIP_DEVICE_COMPILE__)
  static PT_HD float rcp(float x) { if constexpr (FAST) return __builtin_amdgcn_rcpf(x); else return 1.0f / x; }
  static PT_HD float div(float a, float b) { if constexpr (FAST) return a * __builtin_amdgcn_rcpf(b); else return a / b; }
  static PT_HD float sqrt(float x) { if constexpr (FAST) return __builtin_amdgcn_sqrtf(x); else return __builtin_sqrtf(x); }
  static PT_HD v3 normalize(v3 a) { if constexpr (FAST) return a * __builtin_amdgcn_rsqf(dot(a, a)); else return pt::normalize(a); }
#else                                   // host pass of hipcc / tests/hostsim: exact arithmetic whatever FAST says
  static PT_HD float rcp(float x) { return 1.0f / x; }
  static PT_HD float div(float a, float b) { return a / b; }
  static PT_HD float sqrt(float x) { return __builtin_sqrtf(x); }
  static PT_HD v3 normalize(v3 a) { return pt::normalize(a); }
#endif
};

// utils_device.h:63-67
PT_HD float fresnel(float cosThetaI, float cosThetaT, float refIdx) {
  float rs = (cosThetaI - cosThetaT * refIdx) / (cosThetaI + refIdx * cosThetaT);
  float rp = (cosThetaI * refIdx - cosThetaT) / (cosThetaI * refIdx + cosThetaT);
  return 0.5f * (rs * rs + rp * rp);
}
// utils_device.h:130-137 with a = ccAlpha of the material (the only `a` it is called with)
template <bool FAST = false>
PT_HD float GTR1_cc(float NDotH, const DevMaterial& m) {
  if (m.ccAlpha >= 1.f) return 1.f / kPi;
  float t = 1.f + m.ccA2m1 * NDotH * NDotH;
  return ShadeMath<FAST>::div(m.ccA2m1, m.ccPiLogA2 * t);
}
// utils_device.h:139-143
template <bool FAST = false>
PT_HD float GTR2(float NDotH, float a) {
  float a2 = a * a;
  float t = 1.f + (a2 - 1.f) * NDotH * NDotH;
  return ShadeMath<FAST>::div(a2, kPi * t * t);
}
// utils_device.h:149-151
template <bool FAST = false>
PT_HD float GTR2Aniso(float NdotH, float HdotX, float HdotY, float ax, float ay) {
  typedef ShadeMath<FAST> SM;
  return SM::rcp(kPi * ax * ay * sqr(sqr(SM::div(HdotX, ax)) + sqr(SM::div(HdotY, ay)) + NdotH * NdotH));
}
// utils_device.h:153-157
PT_HD float schlickFresnel(float u) {
  float m = clampf(1.f - u, 0.f, 1.f);
  float m2 = m * m;
  return m2 * m2 * m;
}
// utils_device.h:159-163
template <bool FAST = false>
PT_HD float smithGGgx(float NdotV, float alphaG) {
  float a = alphaG * alphaG;
  float b = NdotV * NdotV;
  return ShadeMath<FAST>::rcp(NdotV + ShadeMath<FAST>::sqrt(a + b - a * b));
}
// utils_device.h:165-167
template <bool FAST = false>
PT_HD float smithGGgxAniso(float NdotV, float VdotX, float VdotY, float ax, float ay) {
  return ShadeMath<FAST>::rcp(NdotV + ShadeMath<FAST>::sqrt(sqr(VdotX * ax) + sqr(VdotY * ay) + sqr(NdotV)));
}
// utils_device.h:182-185
PT_HD float powerHeuristic(float a, float b) { float t = a * a; return t / (b * b + t); }

// optix cosine_sample_hemisphere
PT_HD v3 cosine_sample_hemisphere(float u1, float u2) {
  const float r = __builtin_sqrtf(u1);
  const float phi = (2.0f * kPi) * u2;
  v3 p;
  float sinPhi, cosPhi;
  sincos_ac(phi, sinPhi, cosPhi);
  p.x = r * cosPhi;
  p.y = r * sinPhi;
  p.z = __builtin_sqrtf(fmaxf_(0.0f, 1.0f - p.x * p.x - p.y * p.y));
  return p;
}

// disney.h:9-30
// onb = make_onb(N): disney.h builds it in disneySample and again in disneyEval; the caller builds it once.
//
// The two lobes are written as ONE instruction stream with selects: a wave whose lanes chose different lobes would otherwise run
// both branches one after the other (two sincos, four square roots, four normalisations: ~350 vector instructions against ~250
// here).  Every lane performs exactly the operations of its own branch of disney.h on exactly its operands, in the same order
// -- the other lobe's values are computed beside them and dropped --, so L and H are the same bits as before:
//   diffuse (disney.h:12-18)   u1 = ra, u2 = rb: cosine_sample_hemisphere -> r = sqrt(u1), phi = (2 pi) u2,
//                              p = (r cos phi, r sin phi, sqrt(max(0, 1 - px px - py py))); L = normalize(onb p); H = normalize(L + V)
//   specular (disney.h:20-29)  phi = ra * 2 * pi, random = rb: cosTheta = sqrt((1 - random) / (1 + (a a - 1) random)),
//                              sinTheta = sqrt(1 - cosTheta cosTheta), h = onb (sinTheta cos phi, sinTheta sin phi, cosTheta);
//                              L = normalize(h * (2 dot(V, h)) - V); H = normalize(h)
PT_HD_BRDF void disney_sample(uint32_t& seed, const DevMaterial& m, const Onb& onb, v3 V, v3& L, v3& H) {
  typedef ShadeMath<false> SM;         // directions are part of the path: always exact
  const bool diffuse = rnd(seed) < m.diffuseRatio;
  const float ra = rnd(seed), rb = rnd(seed);
  const float phi = diffuse ? (2.0f * kPi) * rb : ra * 2.0f * kPi;
  float sinPhi, cosPhi;
  sincos_ac(phi, sinPhi, cosPhi);
  const float a = m.specAlpha;
  const float s1 = SM::sqrt(diffuse ? ra : SM::div(1.f - rb, 1.0f + (a * a - 1.f) * rb));      // r | cosTheta
  const float px = s1 * cosPhi, py = s1 * sinPhi;                                               // the diffuse lobe's p.x, p.y
  const float s2 = SM::sqrt(diffuse ? fmaxf_(0.0f, 1.0f - px * px - py * py) : 1.0f - (s1 * s1));   // p.z | sinTheta
  const v3 local = diffuse ? mk3(px, py, s2) : mk3(s2 * cosPhi, s2 * sinPhi, s1);
  const v3 w = onb_inverse(onb, local);                                                          // l | h
  L = SM::normalize(diffuse ? w : w * (2.0f * dot(V, w)) - V);
  H = SM::normalize(diffuse ? L + V : w);
}

// disney.h:32-46
template <bool FAST = false>
PT_HD_BRDF float disney_pdf(const DevMaterial& m, v3 N, v3 L, v3 H) {
  typedef ShadeMath<FAST> SM;
  float specularRatio = 1.f - m.diffuseRatio;
  float cosTheta = __builtin_fabsf(dot(N, H));
  float pdfGTR1 = GTR1_cc<FAST>(cosTheta, m) * cosTheta;
  float pdfGTR2 = GTR2<FAST>(cosTheta, m.specAlpha) * cosTheta;
  float pdfH = lerp(pdfGTR1, pdfGTR2, m.ccRatio);
  float pdfL = SM::div(pdfH, 4.0f * __builtin_fabsf(dot(L, H)));
  float pdfDiff = SM::div(__builtin_fabsf(dot(N, L)), kPi);
  return m.diffuseRatio * pdfDiff + specularRatio * pdfL;
}

// disney.h:48-91
// What disneyEval derives from the hit alone (N, V and the material), the same for every direction L it is evaluated for at
// that hit: the tangent frame X, Y (disney.h:76-77), schlickFresnel(NdotV) (:65) and the V factors of the two Smith terms
// (:81-82, :86).  The reference recomputes them in every call; a Disney hit with three facing lights calls disneyEval four
// times, so the packet visit (pt_packet.h) takes them once -- the same operations on the same inputs, the same bits.
struct DisneyView { v3 X, Y; float NdotV, FV, GsV, GrV; };
template <bool FAST = false>
PT_HD_BRDF DisneyView disney_view(const DevMaterial& m, const Onb& onb, v3 V) {
  typedef ShadeMath<FAST> SM;
  const v3 N = onb.normal;
  DisneyView dv;
  dv.NdotV = dot(N, V);
  dv.FV = schlickFresnel(dv.NdotV);
  dv.X = SM::normalize(onb.tangent);
  dv.Y = SM::normalize(cross(N, dv.X));
  dv.GsV = smithGGgxAniso<FAST>(dv.NdotV, dot(V, dv.X), dot(V, dv.Y), m.ax, m.ay);
  dv.GrV = smithGGgx<FAST>(dv.NdotV, 0.25f);
  return dv;
}
// Cdlin/Cspec0/Csheen: the material's constants, or the per-hit ones of a textured material
template <bool FAST = false>
PT_HD_BRDF v3 disney_eval(const DevMaterial& m, v3 Cdlin, v3 Cspec0, v3 Csheen, v3 N, const DisneyView& dv, v3 L, v3 H) {
  typedef ShadeMath<FAST> SM;
  float NdotL = dot(N, L), NdotV = dv.NdotV, NdotH = dot(N, H), LdotH = dot(L, H);
  const v3 one = mk3(1.f, 1.f, 1.f);

  float FL = schlickFresnel(NdotL);
  float FV = dv.FV;
  float Fd90 = 0.5f + 2.f * LdotH * LdotH * m.roughness;
  float Fd = lerp(1.f, Fd90, FL) * lerp(1.f, Fd90, FV);

  float Fss90 = LdotH * LdotH * m.roughness;
  float Fss = lerp(1.0f, Fss90, FL) * lerp(1.0f, Fss90, FV);
  float ss = 1.25f * (Fss * (SM::rcp(NdotL + NdotV) - 0.5f) + 0.5f);

  const v3 X = dv.X, Y = dv.Y;
  float Ds = GTR2Aniso<FAST>(NdotH, dot(H, X), dot(H, Y), m.ax, m.ay);
  float FH = schlickFresnel(LdotH);
  v3 Fs = lerp(Cspec0, one, FH);
  float Gs = smithGGgxAniso<FAST>(NdotL, dot(L, X), dot(L, Y), m.ax, m.ay) * dv.GsV;
  v3 Fsheen = Csheen * (FH * m.sheen);
  float Dr = GTR1_cc<FAST>(NdotH, m);
  float Fr = lerp(0.04f, 1.f, FH);
  float Gr = smithGGgx<FAST>(NdotL, 0.25f) * dv.GrV;
  v3 diffuse = (Cdlin * ((1.0f / kPi) * lerp(Fd, ss, m.subsurface)) + Fsheen) * m.oneMinusMetallic;
  v3 spec = (Fs * Gs) * Ds;
  float cc = 0.25f * m.clearcoat * Gr * Fr * Dr;
  return (diffuse + spec) + cc;
}
// one evaluation, as disney.h writes it
template <bool FAST = false>
PT_HD_BRDF v3 disney_eval(const DevMaterial& m, v3 Cdlin, v3 Cspec0, v3 Csheen, const Onb& onb, v3 L, v3 V, v3 H) {
  const DisneyView dv = disney_view<FAST>(m, onb, V);
  return disney_eval<FAST>(m, Cdlin, Cspec0, Csheen, onb.normal, dv, L, H);
}

}  // namespace pt
