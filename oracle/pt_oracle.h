/*
 * pt_oracle.h -- CPU ORACLE for the MinimalOptiX path-tracing hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, bench.py's
 * `cpu_baseline` leg and __graft_entry__.smoke() may load it, and only as the
 * checker.  Nothing under minimaloptix_amd/ links, imports or calls it.
 *
 * It is a plain-C restatement of the reference's device programs
 * (/root/reference/MinimalOptiX/{Camera,Geometry,Material,miss}.cu, disney.h,
 * utils_device.h, Structures.h) plus the closed-source OptiX 5.1.1 pieces they
 * depend on (rtTrace nearest-hit search, any-hit shadow semantics, the
 * optixu_math helpers), restated from SURVEY.md Appendix A1/A2.
 *
 * PARITY STATUS: the reference cannot be built or run here (OptiX 5.1.1 + CUDA
 * 9.1 NVRTC + NVIDIA GPU), has no tests and no golden vectors.  The integer
 * RNG and the camera maths are pinned by the known-answer vectors of SURVEY.md
 * Appendix A3 (tests/test_oracle_kat.py); the full chain is pinned statistically
 * against fixtures derived from the reference's demo/spheres_{lens,pinhole}.png
 * (tests/golden/).  The OptiX-SDK helper semantics are "parity unpinned".
 *
 * Arithmetic contract (shared, by specification, with the HIP kernels -- each
 * side implements it independently; see DESIGN.md "Arithmetic contract"):
 *   AC1 dot(a,b)   = fmaf(a.z,b.z, fmaf(a.y,b.y, a.x*b.x))
 *   AC2 cross(a,b) = ( fmaf(a.y,b.z,-(a.z*b.y)), fmaf(a.z,b.x,-(a.x*b.z)), fmaf(a.x,b.y,-(a.y*b.x)) )
 *   AC3 length(v)=sqrtf(dot(v,v)); normalize(v)=v*(1.0f/sqrtf(dot(v,v))); v/s = v*(1.0f/s)
 *   AC4 every other C operator is one IEEE binary32 operation, in source order
 *       (compile with -ffp-contract=off)
 *   AC5 sinf/cosf(x) := one specified binary32 algorithm (quadrant + 3-step Cody-Waite reduction with fma +
 *       degree-7/8 kernels; see sincos_ac), absolute error < 2e-7 on [0, 2 pi]
 *   AC6 logf of per-material constants is evaluated with the host libm; x^2.2 (srgb2lin) is
 *       (float)pow((double)x, (double)2.2f) -- it is per-hit for textured materials
 *   AC7 point on ray  p = fmaf(t, d, o) per component
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* material kinds == which closest-hit program the instance carries */
enum { ORC_LAMBERTIAN = 0, ORC_METAL = 1, ORC_GLASS = 2, ORC_DISNEY = 3, ORC_LIGHT = 4 };
enum { ORC_BRDF_NORMAL = 0, ORC_BRDF_GLASS = 1 };      /* Structures.h:49 */
enum { ORC_LIGHT_SPHERE = 0, ORC_LIGHT_QUAD = 1 };     /* Structures.h:68 */

/* One record for every material program's parameters (Structures.h:35-66,70-80). */
typedef struct OrcMaterial {
  int32_t kind;            /* ORC_* */
  float   albedo[3];       /* lambertian / metal / glass albedo */
  float   fuzz;            /* metal */
  float   refIdx;          /* glass */
  float   emission[3];     /* light: LightParams.emission ; disney: DisneyParams.emission */
  /* DisneyParams */
  float   color[3];
  float   metallic, subsurface, specular, roughness, specularTint, anisotropic;
  float   sheen, sheenTint, clearcoat, clearcoatGloss;
  int32_t brdfType;
  int32_t albedoID;        /* 0 == RT_TEXTURE_ID_NULL, else textures[albedoID-1] */
} OrcMaterial;

typedef struct OrcSphere { float center[3]; float radius; int32_t mat; } OrcSphere;          /* Structures.h:22 */
typedef struct OrcQuad   { float plane[4]; float v1[3]; float v2[3]; float anchor[3]; int32_t mat; } OrcQuad; /* :28 */

typedef struct OrcLight {                                                                 /* Structures.h:70 */
  float position[3], normal[3], emission[3], u[3], v[3];
  float area, radius;
  int32_t shape;
} OrcLight;

typedef struct OrcCam {                                                                   /* Structures.h:12 */
  float origin[3], horizontal[3], vertical[3], scrLowerLeftCorner[3], u[3], v[3];
  float lensRadius;
} OrcCam;

/* RT_FORMAT_FLOAT4 texture buffer with the sampler of MinimalOptiX.cpp:449-474:
 * RT_WRAP_REPEAT, normalized coordinates, RT_FILTER_LINEAR.  Row 0 = v 0. */
typedef struct OrcTexture { int32_t width, height; const float* rgba; } OrcTexture;

typedef struct OrcScene {
  int32_t width, height;
  OrcCam  cam;
  float   bgColor[3];
  int32_t rayMaxDepth;       /* 256   MinimalOptiX.h:85 */
  float   rayMinIntensity;   /* 1e-3  MinimalOptiX.h:88 */
  float   rayEpsilonT;       /* 1e-3  MinimalOptiX.h:89 */

  int32_t nMaterials; const OrcMaterial* materials;
  int32_t nSpheres;   const OrcSphere*   spheres;
  int32_t nQuads;     const OrcQuad*     quads;
  int32_t nLights;    const OrcLight*    lights;    /* the `lights` buffer used for NEE */

  /* flattened triangle meshes (Geometry.cu:114-119 buffers, concatenated) */
  int32_t nVerts;  const float* positions;   /* 3*nVerts  */
  int32_t nNorms;  const float* normals;     /* 3*nNorms  */
  int32_t nUVs;    const float* texcoords;   /* 2*nUVs    */
  int32_t nFaces;
  const int32_t* vIdx;   /* 3*nFaces */
  const int32_t* nIdx;   /* 3*nFaces, <0 => face has no shading normals */
  const int32_t* tIdx;   /* 3*nFaces, <0 => no texcoords */
  const int32_t* faceMat;/* nFaces */

  int32_t bruteForceTris;    /* !=0: skip the oracle's BVH (validation of the BVH itself) */
  int32_t nTextures; const OrcTexture* textures;
} OrcScene;

typedef struct OrcStats {
  uint64_t primaryRays, bounceRays, shadowRays;
  uint64_t samples, closestHits, misses, depthCapped;
} OrcStats;

/* Render samples for pixels x in [x0,x1), y in [y0,y1).  For each seed s in
 * seeds[0..nSeeds) -- in that order -- one sample per pixel is traced with the
 * launch seed s and `accum[(y*W+x)*3+c] += clamp(color,0,1)` exactly as
 * Camera.cu:21-42 does per launch.  accum is the full W*H*3 buffer, row 0 =
 * bottom row.  Returns 0, or <0 on bad arguments.  nThreads<=0 => all cores. */
int orc_render(const OrcScene* sc, const int32_t* seeds, int nSeeds,
               float* accum, int x0, int y0, int x1, int y1,
               int nThreads, OrcStats* stats);

/* trace one radiance path for a given ray/seed (unit tests / debugging) */
void orc_trace_one(const OrcScene* sc, const float org[3], const float dir[3],
                   int32_t seed, float outColor[3]);

/* nearest-hit query used by BVH-vs-brute-force tests. returns prim id or -1 */
int orc_closest_hit(const OrcScene* sc, const float org[3], const float dir[3],
                    float tmin, float tmax, float* tHit);
/* "disney_binary64" = 1: evaluate disneySample/Pdf/Eval in binary64 (analysis only; see pt_oracle.c) */
int orc_set_option(const char* name, int value);
/* analysis: effect of the per-sample clamp (raw sums capped at `cap`, samples above 1, clamped sums; [H][W][3] each) */
int orc_render_clamp_stats(const OrcScene* sc, const int32_t* seeds, int nSeeds, int x0, int y0, int x1, int y1, float cap,
                           float* rawSum, float* nClamped, float* clampedSum);
/* analysis: per-pixel sums of the clamped samples and sample counts by the deepest radiance rtTrace (see pt_oracle.c) */
int orc_render_by_depth(const OrcScene* sc, const int32_t* seeds, int nSeeds, int x0, int y0, int x1, int y1,
                        int nBuckets, float* colourSum, float* count);
int orc_closest_hit_batch(const OrcScene* sc, const float* rays, int n, int32_t* outPrim, float* outT);

/* ---- small pure functions exported for known-answer tests ---------------- */
uint32_t orc_tea16(uint32_t v0, uint32_t v1);                 /* utils_device.h:8  */
uint32_t orc_lcg(int32_t* seed);                              /* utils_device.h:24 */
float    orc_rand(int32_t* seed);                             /* utils_device.h:32 */
int32_t  orc_launch_seed(uint32_t i, uint32_t baseSeed);      /* SURVEY 8d seed schedule */
void orc_set_cam_params(const float from[3], const float at[3], const float up[3],
                        float vFoV, float aspect, float aperture, float focus, OrcCam* out); /* utils_host.cpp:77 */
void orc_set_quad_params(const float anchor[3], const float v1[3], const float v2[3], OrcQuad* out); /* utils_host.cpp:67 */
void orc_init_disney(OrcMaterial* m);                         /* utils_host.cpp:101 */
int  orc_refract(float r[3], const float i[3], const float n[3], float ior);
void orc_offset(const float p[3], const float n[3], float out[3]);               /* utils_device.h:82 */
float orc_disney_pdf(const OrcMaterial* m, const float N[3], const float L[3], const float V[3], const float H[3]);
void  orc_disney_eval(const OrcMaterial* m, const float base[3], const float N[3], const float L[3],
                      const float V[3], const float H[3], float out[3]);
void  orc_disney_sample(int32_t* seed, const OrcMaterial* m, const float N[3], const float V[3],
                        float L[3], float H[3]);
int   orc_intersect_triangle(const float o[3], const float d[3], float tmin, float tmax,
                             const float p0[3], const float p1[3], const float p2[3],
                             float n[3], float* t, float* beta, float* gamma);
/* MinimalOptiX::move (MinimalOptiX.cpp:562-585): one sphere {center[3], radius, velocity[3]} for `time` seconds */
void  orc_move_sphere(float center[3], float radius, float velocity[3], float time);
/* rtTex2D<float4> restated (SURVEY A1 / CUDA linear filtering with 8-bit interpolation weights) */
void  orc_tex2d(const OrcTexture* t, float u, float v, float out[4]);
void  orc_sincos(float x, float* s, float* c);                /* AC5 */
int   orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
