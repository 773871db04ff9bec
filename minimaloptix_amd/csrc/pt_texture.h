// pt_texture.h -- albedo textures: rtTex2D<float4> with the sampler the reference creates
// (MinimalOptiX.cpp:449-474: RT_WRAP_REPEAT, normalized coordinates, RT_FILTER_LINEAR) and the
// colour-dependent Disney constants of disney.h:49-60, which become per-hit values when the base
// colour comes from a texture (Material.cu:128-132).
#pragma once
#include "pt_types.h"

namespace pt {

// AC6: x^2.2 through double precision, so that host upload and device agree
PT_HD float pow22(float x) { return (float)pow((double)x, (double)2.2f); }
PT_HD v3 srgb2lin(v3 c) { return mk3(pow22(c.x), pow22(c.y), pow22(c.z)); }      // utils_device.h:173-175

// Cspec0 / Csheen from the linear base colour (disney.h:52-57), shared by upload and the textured path
PT_HD void disney_color_constants(v3 Cdlin, float specular, float specularTint, float sheenTint, float metallic,
                                  v3& Cspec0, v3& Csheen) {
  const v3 one = mk3(1.f, 1.f, 1.f);
  const float Cdlum = dot(Cdlin, mk3(0.3f, 0.6f, 0.1f));
  const v3 Ctint = Cdlum > 0.f ? Cdlin / Cdlum : one;
  Cspec0 = lerp(lerp(one, Ctint, specularTint) * (specular * 0.08f), Cdlin, metallic);
  Csheen = lerp(one, Ctint, sheenTint);
}

PT_HD int tex_wrap(int i, int n) { i %= n; return i < 0 ? i + n : i; }

// Bilinear fetch as CUDA texture units do it (OptiX 5 samplers are CUDA textures): texel centres at
// +0.5, interpolation weights quantised to 8 fractional bits, repeat addressing of both taps.
PT_HD v4 tex2d(const DevTexture& t, float u, float v) {
  const float x = (u - floorf(u)) * (float)t.width - 0.5f;
  const float y = (v - floorf(v)) * (float)t.height - 0.5f;
  const float fx = floorf(x), fy = floorf(y);
  const float ax = (float)(int)((x - fx) * 256.0f + 0.5f) * (1.0f / 256.0f);
  const float ay = (float)(int)((y - fy) * 256.0f + 0.5f) * (1.0f / 256.0f);
  const int i0 = tex_wrap((int)fx, t.width), i1 = tex_wrap((int)fx + 1, t.width);
  const int j0 = tex_wrap((int)fy, t.height), j1 = tex_wrap((int)fy + 1, t.height);
  const v4 t00 = t.texels[(size_t)j0 * t.width + i0], t10 = t.texels[(size_t)j0 * t.width + i1];
  const v4 t01 = t.texels[(size_t)j1 * t.width + i0], t11 = t.texels[(size_t)j1 * t.width + i1];
  v4 r;
  { const float lo = t00.x + ax * (t10.x - t00.x), hi = t01.x + ax * (t11.x - t01.x); r.x = lo + ay * (hi - lo); }
  { const float lo = t00.y + ax * (t10.y - t00.y), hi = t01.y + ax * (t11.y - t01.y); r.y = lo + ay * (hi - lo); }
  { const float lo = t00.z + ax * (t10.z - t00.z), hi = t01.z + ax * (t11.z - t01.z); r.z = lo + ay * (hi - lo); }
  { const float lo = t00.w + ax * (t10.w - t00.w), hi = t01.w + ax * (t11.w - t01.w); r.w = lo + ay * (hi - lo); }
  return r;
}

}  // namespace pt
