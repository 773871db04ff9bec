// pt_geom.h -- primitive intersectors and hit attributes (Geometry.cu) + hit-point
// refinement (utils_device.h:72-128).  One function per reference program; the
// traversal code decides *which* primitives are tested, these decide hit/no-hit.
#pragma once
#include "pt_types.h"

namespace pt {

// rtPotentialIntersection with the deterministic equal-t rule:
// t in (tmin, tbest) wins; at exactly equal t the lower primitive id wins.
PT_HD bool potential(float t, int prim, float tmin, float tbest, int bestPrim) {
  return ((t > tmin) & (t < tbest)) | ((t == tbest) & (bestPrim >= 0) & (prim < bestPrim));
}

// Geometry.cu:18-28: roots of the unit-direction quadratic
PT_HD bool sphere_roots(v3 center, float radius, v3 o, v3 d, float& t1, float& t2) {
  v3 oc = o - center;
  float b = dot(d, oc);
  float c = dot(oc, oc) - radius * radius;
  float disc = b * b - c;
  if (disc < 0) return false;
  float sq = __builtin_sqrtf(disc);
  t1 = -b - sq; t2 = -b + sq;
  return true;
}

// Geometry.cu:70-91 (without the rtPotentialIntersection step)
PT_HD bool quad_test(const v4& plane, v3 v1, v3 v2, v3 anchor, v3 o, v3 d, float tmin, float tmax, float& tOut) {
  v3 n = xyz(plane);
  float dt = dot(d, n);
  float t = (plane.w - dot(n, o)) / dt;
  if (t > tmin && t < tmax) {
    v3 p = ray_at(o, d, t);
    v3 vi = p - anchor;
    float a1 = dot(v1, vi);
    if (a1 >= 0 && a1 <= 1) {
      float a2 = dot(v2, vi);
      if (a2 >= 0 && a2 <= 1) { tOut = t; return true; }
    }
  }
  return false;
}

// optix intersect_triangle_branchless on the precomputed edges
// e0 = p1-p0, e1 = p0-p2 (same subtractions the reference performs per call).
PT_HD bool tri_test(v3 o, v3 d, float tmin, float tmax, v3 p0, v3 e0, v3 e1,
                    v3& n, float& t, float& beta, float& gamma) {
  n = cross(e1, e0);
  const v3 e2 = (p0 - o) * (1.0f / dot(n, d));
  const v3 i = cross(d, e2);
  beta = dot(i, e1);
  gamma = dot(i, e0);
  t = dot(n, e2);
  return (t < tmax) & (t > tmin) & (beta >= 0.0f) & (gamma >= 0.0f) & (beta + gamma <= 1.0f);
}

// utils_device.h:82-104 offset(): nudge along the normal in ULPs (or absolutely near 0)
PT_HD float offset1(float h, float n) {
  const float epsilon = 1.0e-4f;
  const float offs = 4096.0f * 2.0f;
  if ((f2i(h) & 0x7fffffff) < f2i(epsilon)) return h + epsilon * n;
  return i2f(f2i(h) + (int32_t)(__builtin_copysignf(offs, h) * n));
}
PT_HD v3 offset_pt(v3 p, v3 n) { return mk3(offset1(p.x, n.x), offset1(p.y, n.y), offset1(p.z, n.z)); }
// utils_device.h:72-79
PT_HD float intersect_plane(v3 origin, v3 direction, v3 normal, v3 point) {
  return -(dot(normal, origin - point)) / dot(normal, direction);
}
// utils_device.h:108-128
PT_HD void refine_hitpoint(v3 original, v3 direction, v3 normal, v3 p, v3& back, v3& front) {
  float refined_t = intersect_plane(original, direction, normal, p);
  v3 refined = ray_at(original, direction, refined_t);
  if (dot(direction, normal) > 0.0f) { back = offset_pt(refined, normal); front = offset_pt(refined, -normal); }
  else                               { back = offset_pt(refined, -normal); front = offset_pt(refined, normal); }
}

struct HitAttr { v3 geoNormal, shadingNormal, front, back; int mat; float texu, texv; };

}  // namespace pt
