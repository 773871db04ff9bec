// pt_upload.h -- host-side conversion of the C-ABI records (include/moptix.h, i.e. the
// reference's Structures.h PODs) into the device layout of pt_types.h.  Header-only so the
// device layer and tests/hostsim convert identically.
#pragma once
#include <math.h>
#include <string.h>
#include "../../include/moptix.h"
#include "pt_texture.h"

namespace pt {

inline v3 to_v3(const moptix_float3& f) { return mk3(f.x, f.y, f.z); }

// DisneyParams -> DevMaterial incl. the per-material constants of disney.h:49-77 / :32-38.
// Same formulas, same order as the reference evaluates per call (AC6: logf = host libm, x^2.2 through double).
inline DevMaterial make_dev_material(const moptix_material& m) {
  DevMaterial d;
  memset(&d, 0, sizeof(d));
  d.kind = m.kind;
  d.albedo = to_v3(m.albedo); d.fuzz = m.fuzz; d.refIdx = m.refIdx;
  d.emission = to_v3(m.emission);
  d.color = mk3(1.f, 1.f, 1.f);
  if (m.kind == MAT_DISNEY) {
    const moptix_disney_params& p = m.disney;
    d.brdfType = p.brdfType;
    d.color = to_v3(p.color);
    d.emission = to_v3(p.emission);
    d.metallic = p.metallic; d.roughness = p.roughness; d.subsurface = p.subsurface;
    d.sheen = p.sheen; d.clearcoat = p.clearcoat;
    d.albedoTex = p.albedoID; d.specular = p.specular; d.specularTint = p.specularTint; d.sheenTint = p.sheenTint;
    d.Cdlin = srgb2lin(d.color);                                           // AC6
    disney_color_constants(d.Cdlin, p.specular, p.specularTint, p.sheenTint, p.metallic, d.Cspec0, d.Csheen);
    d.oneMinusMetallic = 1.0f - p.metallic;
    d.diffuseRatio = 0.5f * (1.0f - p.metallic);
    d.specAlpha = fmaxf(0.001f, p.roughness);
    d.ccAlpha = lerp(0.1f, 0.001f, p.clearcoatGloss);
    d.ccRatio = 1.0f / (1.0f + p.clearcoat);
    const float aspect = sqrtf(1 - p.anisotropic * 0.9f);
    d.ax = fmaxf(.001f, sqr(p.roughness) / aspect);
    d.ay = fmaxf(.001f, sqr(p.roughness) * aspect);
    const float a2 = d.ccAlpha * d.ccAlpha;
    d.ccA2m1 = a2 - 1.f;
    d.ccPiLogA2 = kPi * logf(a2);
  }
  return d;
}

inline DevQuad make_dev_quad(const moptix_quad_params& q, int mat) {
  DevQuad d;
  memset(&d, 0, sizeof(d));
  d.plane.x = q.plane.x; d.plane.y = q.plane.y; d.plane.z = q.plane.z; d.plane.w = q.plane.w;
  d.v1 = to_v3(q.v1); d.v2 = to_v3(q.v2); d.anchor = to_v3(q.anchor); d.mat = mat;
  return d;
}
inline DevSphere make_dev_sphere(const moptix_sphere_params& s) {
  DevSphere d; d.center = to_v3(s.center); d.radius = s.radius; return d;
}
inline DevLight make_dev_light(const moptix_light_params& l) {
  DevLight d;
  memset(&d, 0, sizeof(d));
  d.position = to_v3(l.position); d.emission = to_v3(l.emission);
  // Material.cu:181 normalizes the quad light's normal at every use; same correctly rounded operations, once (a sphere
  // light's normal is recomputed per sample from the sampled point and never read from here)
  d.normal = l.shape == MOPTIX_LIGHT_QUAD ? normalize(to_v3(l.normal)) : to_v3(l.normal);
  d.u = to_v3(l.u); d.v = to_v3(l.v); d.area = l.area; d.radius = l.radius; d.shape = l.shape;
  return d;
}
inline Cam make_cam(const moptix_cam_params& c) {
  Cam d;
  d.origin = to_v3(c.origin); d.horizontal = to_v3(c.horizontal); d.vertical = to_v3(c.vertical);
  d.scrLowerLeftCorner = to_v3(c.scrLowerLeftCorner); d.u = to_v3(c.u); d.v = to_v3(c.v);
  d.lensRadius = c.lensRadius;
  return d;
}

}  // namespace pt
