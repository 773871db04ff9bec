// pt_types.h -- device-resident scene layout for the megakernel (HBM layout; DESIGN.md "Data layout").
//
// Everything the traversal loop touches is 16-byte aligned and sized so that one
// record = a whole number of dwordx4 loads:
//   Node128 : 128 B four-child BVH node (one L2 cache line: 4 child boxes SoA + 4 child refs) -> 7 x dwordx4
//   Tri48   : 48 B  triangle record p0,e0,e1 (+ material / primitive id in .w) -> 3 x dwordx4
//   TriShade: 48 B  the three vertex normals, fetched once per closest hit
// Analytic primitives (spheres/quads, incl. light geometry) live in short brute-force
// lists that are read with wave-uniform (scalar) loads.
#pragma once
#include "pt_math.h"

namespace pt {

// child reference encoding: ref >= 0 -> internal node index; ref < 0 -> leaf, ~ref = (firstTri << 3) | (count-1)
constexpr int kMaxLeaf = 8;
// Scene tables (everything SceneView points at) are IMMUTABLE while a render kernel runs: they are written by uploads and by the
// acceleration build, never by a trace kernel.  load_const reads them through the constant address space, which is how that
// contract reaches the compiler.  A persistent kernel stores to global memory all the time (slot records, samples), so without
// it every scene load counts as clobbered: `table[i]` with a wave-uniform i becomes a VECTOR load per lane (64 lanes fetching
// the same bytes into 64 x n registers) and per-lane gathers are re-issued after every store.  With it
//   * a wave-uniform index (light and quad lists, the sphere chunks) becomes one s_load into scalar registers -- load_uniform,
//   * a per-lane index (nodes, triangle and shading records, the hit's material) stays a vector gather that the compiler may
//     merge, hoist and keep across the kernel's own stores -- load_const.
// Both are the same function; the two names say which of the two the call site relies on.  CONSEQUENCE: a kernel that ever
// updates one of these tables in place (a refit, a writable material) must not read that table through here in the same
// launch -- it would see hoisted / stale values without any diagnostic.
template <class T>
PT_HD T load_const(const T* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  static_assert(sizeof(T) % 4 == 0, "whole dwords");
  typedef __attribute__((address_space(4))) const unsigned int cu32;
  cu32* w = (cu32*)reinterpret_cast<const unsigned int*>(p);
  T out;
  unsigned int* o = reinterpret_cast<unsigned int*>(&out);
#pragma unroll
  for (unsigned i = 0; i < sizeof(T) / 4; i++) o[i] = w[i];
  return out;
#else
  return *p;
#endif
}
template <class T> PT_HD T load_uniform(const T* p) { return load_const(p); }      // the index is the same in every lane of the wave
// &table[index] with a 32-bit byte offset (every scene table is far below 4 GB): on the device the address is then the table's
// base in scalar registers + one 32-bit vector offset -- one shift (or multiply) per gather instead of 64-bit address arithmetic.
template <class T> PT_HD T* at32(T* table, int index) {
#if defined(__HIP_DEVICE_COMPILE__)
  return reinterpret_cast<T*>(reinterpret_cast<char*>(table) + (uint32_t)((uint32_t)index * (uint32_t)sizeof(T)));
#else
  return table + index;
#endif
}
template <class T> PT_HD const T* at32(const T* table, int index) {
#if defined(__HIP_DEVICE_COMPILE__)
  return reinterpret_cast<const T*>(reinterpret_cast<const char*>(table) + (uint32_t)((uint32_t)index * (uint32_t)sizeof(T)));
#else
  return table + index;
#endif
}

struct alignas(16) i4r { int x, y, z, w; };      // four child references as one 16-byte load

PT_HD int make_leaf_ref(int first, int count) { return ~((first << 3) | (count - 1)); }
PT_HD int leaf_first(int ref) { return (~ref) >> 3; }
PT_HD int leaf_count(int ref) { return ((~ref) & 7) + 1; }
constexpr int kCensusRegions = 24;     // counting build: divergent regions of the passes whose lanes are counted (pt_path.h census<>)
constexpr int kTravDone = 0x7fffffff;   // traversal finished sentinel in Trav::node
constexpr int kEmptyRef = 0x7ffffffe;   // "no triangles" root

// Four-wide node: every second level of the binary radix tree is folded into its parent, so a ray makes half
// as many dependent fetches, each of exactly one 128-byte L2 line.  Unused child slots hold kEmptyRef.
struct alignas(128) Node128 {
  v4 lox, loy, loz;     // lower corners of children 0..3 (component k of each = child k)
  v4 hix, hiy, hiz;     // upper corners
  int ref[4];           // child references (node index, leaf ref or kEmptyRef)
  int count;            // children in use (2..4)
  int pad[3];
};
static_assert(sizeof(Node128) == 128, "Node128 must be 128 bytes");

// The same node in 64 bytes, the form the trace kernels fetch (PT_NODE64, pt_path.h): a lane's own node costs the L1 four
// look-ups instead of seven, and the node array of a scene takes half the L2.  The children's boxes sit on a 256-step grid
// laid over the node's own box: plane = corner + q * step per axis, rounded OUTWARDS when the node is written
// (pt_lbvh.h compress_node), so a quantised box contains the box it stands for and the traversal can only enter more
// boxes, never fewer: which boxes are entered changes the amount of work, not a result (rule D5).
struct alignas(64) Node64 {
  float ox, oy, oz;     // lower corner of the node's box (minus the margin compress_node adds)
  float sx, sy, sz;     // grid step per axis: the smallest float with corner + 255 * step >= the upper corner
  uint32_t q[6];        // lox loy loz hix hiy hiz: byte k = child k, in grid steps from the corner
  int ref[4];           // as Node128::ref (unused children: kEmptyRef; their plane bytes are 255 / 0)
};
static_assert(sizeof(Node64) == 64, "Node64 must be 64 bytes");

struct alignas(16) Tri48 {
  v3 p0; int mat;        // material id of the face
  v3 e0; int prim;       // e0 = p1-p0 ; prim = original face index (upload order)
  v3 e1; int shadow;     // e1 = p0-p2 ; what the face is to a shadow ray (SHADOW_*, from its material at build time)
};
static_assert(sizeof(Tri48) == 48, "Tri48 must be 48 bytes");

struct alignas(16) TriShade {
  v3 n0; int hasNormals;
  v3 n1; int pad1;
  v3 n2; int pad2;
};
static_assert(sizeof(TriShade) == 48, "TriShade must be 48 bytes");

// texcoords of one face (upload order, indexed by Tri48::prim); fetched only when the hit material is textured
struct alignas(16) TriUV { float u0, v0, u1, v1, u2, v2; int hasUV; int pad; };
static_assert(sizeof(TriUV) == 32, "TriUV must be 32 bytes");

// RT_FORMAT_FLOAT4 texture buffer behind a sampler (MinimalOptiX.cpp:449-474); row 0 = v 0
struct DevTexture { const v4* texels; int width, height; };

struct alignas(16) DevQuad {   // QuadParams (Structures.h:28) + material
  v4 plane;
  v3 v1; int mat;
  v3 v2; int pad0;
  v3 anchor; int pad1;
};
static_assert(sizeof(DevQuad) == 64, "DevQuad must be 64 bytes");

struct alignas(16) DevSphere { v3 center; float radius; };   // + sphereMat[] side array
static_assert(sizeof(DevSphere) == 16, "DevSphere must be 16 bytes");

struct alignas(16) DevLight {  // LightParams (Structures.h:70)
  v3 position; float area;
  v3 normal;   float radius;   // quad lights: normalize(LightParams.normal), taken at upload (pt_upload.h)
  v3 emission; int shape;
  v3 u; int pad0;
  v3 v; int pad1;
};
static_assert(sizeof(DevLight) == 80, "DevLight must be 80 bytes");

enum { MAT_LAMBERTIAN = 0, MAT_METAL = 1, MAT_GLASS = 2, MAT_DISNEY = 3, MAT_LIGHT = 4 };
enum { BRDF_NORMAL = 0, BRDF_GLASS = 1 };
enum { LIGHT_SPHERE = 0, LIGHT_QUAD = 1 };
// What a primitive is to a shadow ray (disneyAnyHit, Material.cu:225-232): no any-hit program at all (lights, non-Disney
// materials), an opaque Disney surface (attenuation 0, rtTerminateRay) or a Disney GLASS surface (attenuation *= colour).  The
// builder writes it into every triangle record, so that a shadow ray's leaf visit needs the material table only for the
// colour of a glass surface: the fetch of the material behind a hit triangle was a second dependent round trip inside the pass.
// Materials are fixed once the acceleration structure is built (moptix_add_material invalidates it).
enum { SHADOW_NONE = 0, SHADOW_OPAQUE = 1, SHADOW_GLASS = 2 };
PT_HD int shadow_class(int kind, int brdfType) { return kind != MAT_DISNEY ? SHADOW_NONE : (brdfType == BRDF_GLASS ? SHADOW_GLASS : SHADOW_OPAQUE); }
// the builder's per-face material input: material id in the low bits, SHADOW_* above them
constexpr int kFaceMatBits = 28;
PT_HD int face_mat_word(int mat, int shadow) { return mat | (shadow << kFaceMatBits); }

// Material record: program parameters + the Disney constants that depend on the
// material only (disney.h:49-77 evaluates them per call; they are pure functions of
// DisneyParams, so they are evaluated once at upload with the same formulas -- AC6).
struct alignas(16) DevMaterial {
  int kind; int brdfType; float fuzz; float refIdx;
  v3 albedo;   float metallic;
  v3 emission; float roughness;
  v3 color;    float subsurface;
  // derived (valid for kind == MAT_DISNEY, untextured)
  v3 Cdlin;    float sheen;          // srgb2lin(color)
  v3 Cspec0;   float clearcoat;      // lerp(specular*.08*lerp(1,Ctint,specularTint), Cdlin, metallic)
  v3 Csheen;   float oneMinusMetallic;
  float diffuseRatio;                // .5*(1-metallic)
  float specAlpha;                   // max(.001, roughness)
  float ccAlpha;                     // lerp(.1,.001,clearcoatGloss)
  float ccRatio;                     // 1/(1+clearcoat)
  float ax, ay;                      // max(.001, roughness^2/aspect), max(.001, roughness^2*aspect)
  float ccA2m1;                      // ccAlpha^2 - 1
  float ccPiLogA2;                   // M_PIf * logf(ccAlpha^2)
  // textured materials (albedoTex != 0) derive Cdlin/Cspec0/Csheen per hit from the sampled base colour
  int albedoTex;                     // 0 == RT_TEXTURE_ID_NULL, else textures[albedoTex-1]
  float specular, specularTint, sheenTint;
};
static_assert(sizeof(DevMaterial) == 160, "DevMaterial layout");

struct Cam { v3 origin, horizontal, vertical, scrLowerLeftCorner, u, v; float lensRadius; };

// Everything a launch needs, passed by value as the kernel argument (scalar registers).  Every array behind these pointers is
// read-only for the duration of a launch (see load_const above).
struct SceneView {
  int width, height;
  int maxDepth; float minIntensity; float epsT;
  v3 bg;
  Cam cam;
  int nSpheres; const DevSphere* spheres; const int* sphereMat;
  int nQuads;   const DevQuad* quads;
  int nLights;  const DevLight* lights;
  int nMaterials; const DevMaterial* mats;
  int anyDisneyAnalytic;          // any sphere/quad carries a Disney material (shadow any-hit applies)
  int shadowNearest;              // the scene has a Disney GLASS material: shadow rays are decided by their NEAREST any-hit surface (pt_path.h)
  int nTris; int rootRef;         // rootRef: node index, leaf ref or kEmptyRef
  const Node128* nodes; const Tri48* tris; const TriShade* triShade;
  const Node64* nodes64;          // the nodes again, compressed (same indices): what the kernels fetch when PT_NODE64 is on
  const TriUV* triUV;             // per face in upload order, or nullptr (no mesh has texcoords)
  int nTextures; const DevTexture* textures;
};

}  // namespace pt
