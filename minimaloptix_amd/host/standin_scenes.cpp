// standin_scenes.cpp -- deterministic stand-ins for the BASELINE.json configurations whose
// assets are absent from the reference checkout.  They follow the
// structure of the reference's file scenes (meshes with Disney materials + a light list,
// Trbvh) and use the reference's camera formulas for the scene they stand in for.
#include <cmath>
#include <cstring>
#include <stdexcept>

#include "scene_desc.h"

using pt::v3;
using pt::mk3;

namespace moptix {

namespace {

moptix_disney_params disneyDefaults() {
  moptix_disney_params d; memset(&d, 0, sizeof(d));
  d.color = { 1.f, 1.f, 1.f }; d.specular = 0.5f; d.roughness = 0.5f; d.sheenTint = 0.5f; d.clearcoatGloss = 1.0f;
  return d;
}

moptix_light_params quadLight(v3 pos, v3 v1, v3 v2, float e) {   // same derivation as scene.cpp:77-83
  moptix_light_params l; memset(&l, 0, sizeof(l));
  l.shape = MOPTIX_LIGHT_QUAD;
  l.position = { pos.x, pos.y, pos.z };
  const v3 u = v1 - pos, w = v2 - pos;
  l.u = { u.x, u.y, u.z }; l.v = { w.x, w.y, w.z };
  l.area = pt::length(pt::cross(u, w));
  const v3 n = pt::normalize(pt::cross(u, w));
  l.normal = { n.x, n.y, n.z };
  l.emission = { e, e, e };
  return l;
}

void addLightGeometry(SceneDesc& s, const moptix_light_params& l) {   // MinimalOptiX.cpp:495-521
  moptix_material m; memset(&m, 0, sizeof(m)); m.kind = MOPTIX_MAT_LIGHT; m.emission = l.emission;
  const int mat = s.addMaterial(m);
  if (l.shape == MOPTIX_LIGHT_QUAD) {
    moptix_quad_params q;
    setQuadParams(mk3(l.position.x, l.position.y, l.position.z), mk3(l.u.x, l.u.y, l.u.z), mk3(l.v.x, l.v.y, l.v.z), q);
    s.quads.push_back(q); s.quadMat.push_back(mat);
  } else {
    moptix_sphere_params sp; memset(&sp, 0, sizeof(sp)); sp.radius = l.radius; sp.center = l.position;
    s.spheres.push_back(sp); s.sphereMat.push_back(mat);
  }
  s.lights.push_back(l);
}

// axis-aligned quad as two triangles with a constant normal
void addQuadMesh(MeshDesc& m, v3 a, v3 b, v3 c, v3 d, v3 n) {
  const int base = (int)(m.positions.size() / 3), nb = (int)(m.normals.size() / 3);
  for (const v3& p : { a, b, c, d }) { m.positions.push_back(p.x); m.positions.push_back(p.y); m.positions.push_back(p.z); }
  m.normals.push_back(n.x); m.normals.push_back(n.y); m.normals.push_back(n.z);
  const int tri[6] = { 0, 1, 2, 0, 2, 3 };
  for (int k = 0; k < 6; k++) { m.vIdx.push_back(base + tri[k]); m.nIdx.push_back(nb); m.tIdx.push_back(-1); }
}

void includeMesh(SceneDesc& s, const MeshDesc& m) {
  for (size_t k = 0; k < m.vIdx.size(); k++) {
    const int i = m.vIdx[k];
    s.aabb.include(mk3(m.positions[3 * i], m.positions[3 * i + 1], m.positions[3 * i + 2]));
  }
  s.nVertices += m.positions.size() / 3; s.nFaces += m.vIdx.size() / 3;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// Config 4: K coffee makers (rotated about y, laid out on a grid) inside a three-walled room.
void buildDiningStandInScene(SceneDesc& s, const std::string& baseSceneFolder, int copies, uint32_t width, uint32_t height) {
  SceneDesc coffee;
  buildFileScene(coffee, baseSceneFolder, "coffee", width, height, true);
  s = SceneDesc();
  s.name = "dining_standin";
  defaultParams(s.params, width, height);
  s.params.bgColor = { 0.f, 0.f, 0.f };                                    // MinimalOptiX.cpp:287
  s.accel = "Trbvh";
  s.warnings = coffee.warnings;
  if (copies < 1) copies = 1;

  // material variants per copy: rough / metallic / clearcoat / glass mix
  const int cols = (int)ceilf(sqrtf((float)copies));
  const float pitch = 2.4f;
  for (int c = 0; c < copies; c++) {
    const float ang = 0.7f * (float)c;
    const float ca = cosf(ang), sa = sinf(ang);
    const v3 shift = mk3(pitch * (float)(c % cols), 0.f, pitch * (float)(c / cols));
    std::vector<int> matMap(coffee.materials.size(), -1);
    for (const MeshDesc& src : coffee.meshes) {
      if (src.source == "Mesh004.obj") continue;   // the coffee scene's own floor plate
      MeshDesc m = src;
      if (matMap[src.matId] < 0) {
        moptix_material mat = coffee.materials[src.matId];
        moptix_disney_params& d = mat.disney;
        switch (c % 4) {
          case 0: break;
          case 1: d.roughness = fminf(1.f, d.roughness + 0.35f); d.sheen = 0.5f; break;
          case 2: d.metallic = 1.f; d.roughness = fmaxf(0.05f, d.roughness); d.color = { 0.9f, 0.75f, 0.4f }; break;
          case 3: d.clearcoat = 1.f; d.clearcoatGloss = 0.9f; d.color = { 0.15f + 0.1f * (float)(c % 7), 0.3f, 0.6f }; break;
        }
        if (c % 5 == 4 && d.metallic == 0.f && d.color.x > 0.9f) d.brdfType = MOPTIX_BRDF_GLASS;
        matMap[src.matId] = s.addMaterial(mat);
      }
      m.matId = matMap[src.matId];
      for (size_t i = 0; i + 2 < m.positions.size(); i += 3) {
        const float x = m.positions[i], z = m.positions[i + 2];
        m.positions[i] = ca * x + sa * z + shift.x; m.positions[i + 2] = -sa * x + ca * z + shift.z;
      }
      for (size_t i = 0; i + 2 < m.normals.size(); i += 3) {
        const float x = m.normals[i], z = m.normals[i + 2];
        m.normals[i] = ca * x + sa * z; m.normals[i + 2] = -sa * x + ca * z;
      }
      includeMesh(s, m);
      s.meshes.push_back(std::move(m));
    }
  }
  // room: floor, ceiling, back/left/right walls (open towards -x where the camera sits)
  const v3 lo = s.aabb.m_min - mk3(1.5f, 0.f, 1.5f), hi = s.aabb.m_max + mk3(1.5f, 1.2f, 1.5f);
  moptix_disney_params wall = disneyDefaults(); wall.color = { 0.7f, 0.68f, 0.64f }; wall.roughness = 0.8f;
  moptix_disney_params floorP = disneyDefaults(); floorP.color = { 0.578f, 0.578f, 0.578f }; floorP.roughness = 0.05f;
  MeshDesc room; room.source = "room"; room.matId = s.addMaterial(disneyMaterial(wall));
  MeshDesc floorM; floorM.source = "floor"; floorM.matId = s.addMaterial(disneyMaterial(floorP));
  addQuadMesh(floorM, mk3(lo.x, lo.y, lo.z), mk3(lo.x, lo.y, hi.z), mk3(hi.x, lo.y, hi.z), mk3(hi.x, lo.y, lo.z), mk3(0, 1, 0));
  addQuadMesh(room, mk3(lo.x, hi.y, lo.z), mk3(hi.x, hi.y, lo.z), mk3(hi.x, hi.y, hi.z), mk3(lo.x, hi.y, hi.z), mk3(0, -1, 0));
  addQuadMesh(room, mk3(hi.x, lo.y, lo.z), mk3(hi.x, lo.y, hi.z), mk3(hi.x, hi.y, hi.z), mk3(hi.x, hi.y, lo.z), mk3(-1, 0, 0));
  addQuadMesh(room, mk3(lo.x, lo.y, lo.z), mk3(hi.x, lo.y, lo.z), mk3(hi.x, hi.y, lo.z), mk3(lo.x, hi.y, lo.z), mk3(0, 0, 1));
  addQuadMesh(room, mk3(lo.x, lo.y, hi.z), mk3(lo.x, hi.y, hi.z), mk3(hi.x, hi.y, hi.z), mk3(hi.x, lo.y, hi.z), mk3(0, 0, -1));
  includeMesh(s, floorM); includeMesh(s, room);
  s.meshes.push_back(std::move(floorM)); s.meshes.push_back(std::move(room));
  // three quad lights below the ceiling, facing down
  const v3 c = s.aabb.center(); const v3 e = s.aabb.extent();
  const float y = hi.y - 0.05f;
  for (int k = 0; k < 3; k++) {
    const float cx = c.x + (float)(k - 1) * 0.3f * e.x, hw = 0.08f * e.x, hd = 0.15f * e.z;
    addLightGeometry(s, quadLight(mk3(cx - hw, y, c.z - hd), mk3(cx + hw, y, c.z - hd), mk3(cx - hw, y, c.z + hd), 12.f));
  }
  const v3 lookFrom = s.aabb.center() + mk3(-0.7f, 0.f, 0.f) * s.aabb.extent();   // MinimalOptiX.cpp:289-293
  const v3 lookAt = s.aabb.center();
  setCamParams(lookFrom, lookAt, mk3(0.f, 1.f, 0.f), 45, (float)width / (float)height, 0.f, 1.f, s.params.cam);
}

// ------------------------------------------------------------------------------------------
// coffee + a stand-in for its glass pot.  coffee.scene's mesh entry "Mesh010.obj / material Glass" names a file the
// reference checkout does not ship (.MISSING_LARGE_BLOBS:1), so the shipped scene never runs the Disney GLASS branch
// (Material.cu:134-168) or a shadow ray through glass (Material.cu:226-227).  The stand-in is a thin-walled lathe body
// (outer wall, rim, inner wall, double bottom; 96 segments, 8,256 triangles) in the pot's place between the base plate
// (Mesh012, top at y = 0.025) and the lid (Mesh008, y = 0.114), radius 0.072 like the metal band Mesh015, with smooth
// vertex normals and the scene file's Glass material (color 1 1 1, brdf 1).  It lies inside the room, so the scene
// box and with it the camera (MinimalOptiX.cpp:263-267) are those of the shipped scene.
void buildCoffeePotStandInScene(SceneDesc& s, const std::string& baseSceneFolder, uint32_t width, uint32_t height) {
  buildFileScene(s, baseSceneFolder, "coffee", width, height, true);
  s.name = "coffee_pot_standin";
  static const double prof[12][2] = { { 0.0, 0.028 }, { 0.066, 0.028 }, { 0.071, 0.036 }, { 0.072, 0.075 }, { 0.068, 0.105 }, { 0.064, 0.114 },
                                      { 0.061, 0.114 }, { 0.065, 0.105 }, { 0.069, 0.075 }, { 0.068, 0.038 }, { 0.063, 0.031 }, { 0.0, 0.031 } };
  std::vector<double> pr, py;
  for (int k = 0; k + 1 < 12; k++)
    for (int j = 0; j < 4; j++) { const double t = j / 4.0; pr.push_back(prof[k][0] + (prof[k + 1][0] - prof[k][0]) * t); py.push_back(prof[k][1] + (prof[k + 1][1] - prof[k][1]) * t); }
  pr.push_back(prof[11][0]); py.push_back(prof[11][1]);
  const int n = (int)pr.size(), seg = 96;
  std::vector<double> nr(n), ny(n);                       // profile normal = tangent rotated by -90 degrees (outwards along the list)
  for (int i = 0; i < n; i++) {
    const int a = i > 0 ? i - 1 : i, b = i + 1 < n ? i + 1 : i;
    const double tr = (pr[b] - pr[a]) / (b - a), ty = (py[b] - py[a]) / (b - a), l = sqrt(tr * tr + ty * ty);
    nr[i] = ty / l; ny[i] = -tr / l;
  }
  moptix_disney_params glass = disneyDefaults(); glass.brdfType = MOPTIX_BRDF_GLASS; glass.color = { 1.f, 1.f, 1.f };   // coffee.scene:19-23
  MeshDesc pot; pot.source = "pot_standin(Mesh010.obj)"; pot.matId = s.addMaterial(disneyMaterial(glass));
  const double twoPi = 6.283185307179586;
  for (int j = 0; j < seg; j++) {
    const double a = twoPi * j / seg, ca = cos(a), sa = sin(a);
    for (int i = 0; i < n; i++) {
      pot.positions.push_back((float)(pr[i] * ca)); pot.positions.push_back((float)py[i]); pot.positions.push_back((float)(pr[i] * sa));
      pot.normals.push_back((float)(nr[i] * ca)); pot.normals.push_back((float)ny[i]); pot.normals.push_back((float)(nr[i] * sa));
    }
  }
  auto tri = [&](int a, int b, int c) { for (int v : { a, b, c }) { pot.vIdx.push_back(v); pot.nIdx.push_back(v); pot.tIdx.push_back(-1); } };
  for (int j = 0; j < seg; j++) {
    const int j1 = (j + 1) % seg;
    for (int i = 0; i + 1 < n; i++) {
      const int a = j * n + i, b = j * n + i + 1, c = j1 * n + i + 1, d = j1 * n + i;
      if (pr[i] == 0.0) tri(a, c, b);
      else if (pr[i + 1] == 0.0) tri(a, d, b);
      else { tri(a, d, c); tri(a, c, b); }
    }
  }
  includeMesh(s, pot);
  s.meshes.push_back(std::move(pot));
}

// ------------------------------------------------------------------------------------------
// Config 5: displaced (3,2) torus-knot tube of ~nTrisTarget triangles (glass), three Disney
// tessellated spheres, a floor, one sphere light.  bg 0.5 and the dragon camera
// (MinimalOptiX.cpp:336-353).
void buildProceduralMillionScene(SceneDesc& s, int nTrisTarget, uint32_t width, uint32_t height) {
  s = SceneDesc();
  s.name = "million_standin";
  defaultParams(s.params, width, height);
  s.params.bgColor = { 0.5f, 0.5f, 0.5f };
  s.accel = "Trbvh";
  if (nTrisTarget < 1000) nTrisTarget = 1000;
  const int nv = 250;                                         // around the tube
  const int nu = std::max(8, (int)((long long)nTrisTarget * 9 / 10 / (2 * nv)));   // along the knot
  const double twoPi = 6.283185307179586;

  moptix_disney_params glass = disneyDefaults(); glass.brdfType = MOPTIX_BRDF_GLASS; glass.color = { 0.95f, 0.98f, 1.0f };
  MeshDesc knot; knot.source = "torus_knot"; knot.matId = s.addMaterial(disneyMaterial(glass));
  auto centre = [&](double t, double out[3]) {
    const double r = 2.0 + cos(3.0 * t);
    out[0] = r * cos(2.0 * t); out[1] = r * sin(2.0 * t) * 0.0 + sin(3.0 * t) + 1.8; out[2] = r * sin(2.0 * t);
  };
  for (int i = 0; i < nu; i++) {
    const double t = twoPi * i / nu, dt = 1e-4;
    double c0[3], c1[3]; centre(t, c0); centre(t + dt, c1);
    double T[3] = { c1[0] - c0[0], c1[1] - c0[1], c1[2] - c0[2] };
    double tl = sqrt(T[0] * T[0] + T[1] * T[1] + T[2] * T[2]); for (double& x : T) x /= tl;
    double up[3] = { 0, 1, 0 };
    double B[3] = { T[1] * up[2] - T[2] * up[1], T[2] * up[0] - T[0] * up[2], T[0] * up[1] - T[1] * up[0] };
    double bl = sqrt(B[0] * B[0] + B[1] * B[1] + B[2] * B[2]); for (double& x : B) x /= bl;
    double N[3] = { B[1] * T[2] - B[2] * T[1], B[2] * T[0] - B[0] * T[2], B[0] * T[1] - B[1] * T[0] };
    for (int j = 0; j < nv; j++) {
      const double a = twoPi * j / nv;
      const double rad = 0.38 + 0.05 * sin(9.0 * a + 40.0 * t) * cos(7.0 * t);   // displacement
      double n[3], p[3];
      for (int k = 0; k < 3; k++) { n[k] = cos(a) * N[k] + sin(a) * B[k]; p[k] = c0[k] + rad * n[k]; }
      knot.positions.push_back((float)p[0]); knot.positions.push_back((float)p[1]); knot.positions.push_back((float)p[2]);
      knot.normals.push_back((float)n[0]); knot.normals.push_back((float)n[1]); knot.normals.push_back((float)n[2]);
    }
  }
  for (int i = 0; i < nu; i++)
    for (int j = 0; j < nv; j++) {
      const int i1 = (i + 1) % nu, j1 = (j + 1) % nv;
      const int a = i * nv + j, b = i1 * nv + j, c = i1 * nv + j1, d = i * nv + j1;
      const int tri[6] = { a, b, c, a, c, d };
      for (int k = 0; k < 6; k++) { knot.vIdx.push_back(tri[k]); knot.nIdx.push_back(tri[k]); knot.tIdx.push_back(-1); }
    }
  includeMesh(s, knot);
  s.meshes.push_back(std::move(knot));

  // three tessellated Disney spheres ("spheres-as-meshes")
  const int remaining = std::max(3000, nTrisTarget - 2 * nu * nv);
  const int seg = std::max(16, (int)sqrtf((float)remaining / 3.f / 2.f));
  for (int k = 0; k < 3; k++) {
    moptix_disney_params p = disneyDefaults();
    if (k == 0) { p.color = { 0.8f, 0.2f, 0.15f }; p.roughness = 0.3f; p.clearcoat = 1.f; }
    if (k == 1) { p.color = { 0.95f, 0.8f, 0.3f }; p.metallic = 1.f; p.roughness = 0.15f; }
    if (k == 2) { p.color = { 0.2f, 0.5f, 0.8f }; p.roughness = 0.6f; p.sheen = 1.f; }
    MeshDesc m; m.source = "sphere_mesh"; m.matId = s.addMaterial(disneyMaterial(p));
    const v3 ctr = mk3(-4.5f + 4.5f * (float)k, 0.7f, 4.6f); const float r = 0.7f;
    for (int i = 0; i <= seg; i++)
      for (int j = 0; j < 2 * seg; j++) {
        const double th = 3.141592653589793 * i / seg, ph = twoPi * j / (2 * seg);
        const v3 n = mk3((float)(sin(th) * cos(ph)), (float)cos(th), (float)(sin(th) * sin(ph)));
        const v3 p3 = ctr + n * r;
        m.positions.push_back(p3.x); m.positions.push_back(p3.y); m.positions.push_back(p3.z);
        m.normals.push_back(n.x); m.normals.push_back(n.y); m.normals.push_back(n.z);
      }
    for (int i = 0; i < seg; i++)
      for (int j = 0; j < 2 * seg; j++) {
        const int j1 = (j + 1) % (2 * seg);
        const int a = i * 2 * seg + j, b = (i + 1) * 2 * seg + j, c = (i + 1) * 2 * seg + j1, d = i * 2 * seg + j1;
        const int tri[6] = { a, c, b, a, d, c };
        for (int t = 0; t < 6; t++) { m.vIdx.push_back(tri[t]); m.nIdx.push_back(tri[t]); m.tIdx.push_back(-1); }
      }
    includeMesh(s, m);
    s.meshes.push_back(std::move(m));
  }
  moptix_disney_params floorP = disneyDefaults(); floorP.color = { 0.6f, 0.6f, 0.6f }; floorP.roughness = 0.4f;
  MeshDesc floorM; floorM.source = "floor"; floorM.matId = s.addMaterial(disneyMaterial(floorP));
  addQuadMesh(floorM, mk3(-9, 0, -9), mk3(-9, 0, 9), mk3(9, 0, 9), mk3(9, 0, -9), mk3(0, 1, 0));
  includeMesh(s, floorM);
  s.meshes.push_back(std::move(floorM));

  moptix_light_params sl; memset(&sl, 0, sizeof(sl));          // scene.cpp:84-87
  sl.shape = MOPTIX_LIGHT_SPHERE; sl.position = { 0.f, 9.f, 0.f }; sl.radius = 1.5f; sl.normal = { 0.f, -1.f, 0.f };
  sl.emission = { 20.f, 20.f, 20.f }; sl.area = 4.0f * pt::kPi * sl.radius * sl.radius;
  addLightGeometry(s, sl);

  const v3 lookFrom = s.aabb.center() + mk3(0.05f, 0.3f, -0.005f) * s.aabb.extent() * 3.0f;   // dragon camera direction, pulled back
  const v3 lookAt = s.aabb.center();
  setCamParams(lookFrom, lookAt, mk3(0.f, 1.f, 0.f), 30, (float)width / (float)height, 0.f, 1.f, s.params.cam);
}

}  // namespace moptix
