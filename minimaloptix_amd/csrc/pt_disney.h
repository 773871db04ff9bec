// pt_disney.h -- Disney BRDF sampling / pdf / evaluation (disney.h:9-91) and the
// microfacet helpers (utils_device.h:130-185), on the pre-derived material record.
#pragma once
#include "pt_types.h"
#include "pt_rng.h"

namespace pt {

// utils_device.h:63-67
PT_HD float fresnel(float cosThetaI, float cosThetaT, float refIdx) {
  float rs = (cosThetaI - cosThetaT * refIdx) / (cosThetaI + refIdx * cosThetaT);
  float rp = (cosThetaI * refIdx - cosThetaT) / (cosThetaI * refIdx + cosThetaT);
  return 0.5f * (rs * rs + rp * rp);
}
// utils_device.h:130-137 with a = ccAlpha of the material (the only `a` it is called with)
PT_HD float GTR1_cc(float NDotH, const DevMaterial& m) {
  if (m.ccAlpha >= 1.f) return 1.f / kPi;
  float t = 1.f + m.ccA2m1 * NDotH * NDotH;
  return m.ccA2m1 / (m.ccPiLogA2 * t);
}
// utils_device.h:139-143
PT_HD float GTR2(float NDotH, float a) {
  float a2 = a * a;
  float t = 1.f + (a2 - 1.f) * NDotH * NDotH;
  return a2 / (kPi * t * t);
}
// utils_device.h:149-151
PT_HD float GTR2Aniso(float NdotH, float HdotX, float HdotY, float ax, float ay) {
  return 1 / (kPi * ax * ay * sqr(sqr(HdotX / ax) + sqr(HdotY / ay) + NdotH * NdotH));
}
// utils_device.h:153-157
PT_HD float schlickFresnel(float u) {
  float m = clampf(1.f - u, 0.f, 1.f);
  float m2 = m * m;
  return m2 * m2 * m;
}
// utils_device.h:159-163
PT_HD float smithGGgx(float NdotV, float alphaG) {
  float a = alphaG * alphaG;
  float b = NdotV * NdotV;
  return 1.f / (NdotV + __builtin_sqrtf(a + b - a * b));
}
// utils_device.h:165-167
PT_HD float smithGGgxAniso(float NdotV, float VdotX, float VdotY, float ax, float ay) {
  return 1.0f / (NdotV + __builtin_sqrtf(sqr(VdotX * ax) + sqr(VdotY * ay) + sqr(NdotV)));
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
PT_HD_BRDF void disney_sample(uint32_t& seed, const DevMaterial& m, v3 N, v3 V, v3& L, v3& H) {
  Onb onb = make_onb(N);
  if (rnd(seed) < m.diffuseRatio) {
    float u1 = rnd(seed); float u2 = rnd(seed);
    v3 l = cosine_sample_hemisphere(u1, u2);
    l = onb_inverse(onb, l);
    L = normalize(l);
    H = normalize(L + V);
  } else {
    float a = m.specAlpha;
    float phi = rnd(seed) * 2.0f * kPi;
    float random = rnd(seed);
    float cosTheta = __builtin_sqrtf((1.f - random) / (1.0f + (a * a - 1.f) * random));
    float sinTheta = __builtin_sqrtf(1.0f - (cosTheta * cosTheta));
    float sinPhi, cosPhi;
    sincos_ac(phi, sinPhi, cosPhi);
    v3 h = mk3(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta);
    h = onb_inverse(onb, h);
    L = normalize(h * (2.0f * dot(V, h)) - V);
    H = normalize(h);
  }
}

// disney.h:32-46
PT_HD_BRDF float disney_pdf(const DevMaterial& m, v3 N, v3 L, v3 H) {
  float specularRatio = 1.f - m.diffuseRatio;
  float cosTheta = __builtin_fabsf(dot(N, H));
  float pdfGTR1 = GTR1_cc(cosTheta, m) * cosTheta;
  float pdfGTR2 = GTR2(cosTheta, m.specAlpha) * cosTheta;
  float pdfH = lerp(pdfGTR1, pdfGTR2, m.ccRatio);
  float pdfL = pdfH / (4.0f * __builtin_fabsf(dot(L, H)));
  float pdfDiff = __builtin_fabsf(dot(N, L)) / kPi;
  return m.diffuseRatio * pdfDiff + specularRatio * pdfL;
}

// disney.h:48-91
// Cdlin/Cspec0/Csheen: the material's constants, or the per-hit ones of a textured material
PT_HD_BRDF v3 disney_eval(const DevMaterial& m, v3 Cdlin, v3 Cspec0, v3 Csheen, v3 N, v3 L, v3 V, v3 H) {
  Onb onb = make_onb(N);
  float NdotL = dot(N, L), NdotV = dot(N, V), NdotH = dot(N, H), LdotH = dot(L, H);
  const v3 one = mk3(1.f, 1.f, 1.f);

  float FL = schlickFresnel(NdotL);
  float FV = schlickFresnel(NdotV);
  float Fd90 = 0.5f + 2.f * LdotH * LdotH * m.roughness;
  float Fd = lerp(1.f, Fd90, FL) * lerp(1.f, Fd90, FV);

  float Fss90 = LdotH * LdotH * m.roughness;
  float Fss = lerp(1.0f, Fss90, FL) * lerp(1.0f, Fss90, FV);
  float ss = 1.25f * (Fss * (1.f / (NdotL + NdotV) - 0.5f) + 0.5f);

  v3 X = normalize(onb.tangent);
  v3 Y = normalize(cross(N, X));
  float Ds = GTR2Aniso(NdotH, dot(H, X), dot(H, Y), m.ax, m.ay);
  float FH = schlickFresnel(LdotH);
  v3 Fs = lerp(Cspec0, one, FH);
  float Gs = smithGGgxAniso(NdotL, dot(L, X), dot(L, Y), m.ax, m.ay) *
             smithGGgxAniso(NdotV, dot(V, X), dot(V, Y), m.ax, m.ay);
  v3 Fsheen = Csheen * (FH * m.sheen);
  float Dr = GTR1_cc(NdotH, m);
  float Fr = lerp(0.04f, 1.f, FH);
  float Gr = smithGGgx(NdotL, 0.25f) * smithGGgx(NdotV, 0.25f);
  v3 diffuse = (Cdlin * ((1.0f / kPi) * lerp(Fd, ss, m.subsurface)) + Fsheen) * m.oneMinusMetallic;
  v3 spec = (Fs * Gs) * Ds;
  float cc = 0.25f * m.clearcoat * Gr * Fr * Dr;
  return (diffuse + spec) + cc;
}

}  // namespace pt
