/*
 * moptix.h -- C ABI of the MI355X-native device layer that replaces the OptiX 5
 * host API + rtTrace/acceleration + .cu programs on the MinimalOptiX render path.
 *
 * The reference has no FFI of its own: the path sits behind the OptiX C++ host
 * API (optixpp) as called by class MinimalOptiX.  Each entry point below names
 * the reference call site(s) it replaces (paths relative to
 * /root/reference/MinimalOptiX/).  Plain pointers and sizes only; host memory
 * passed in is copied (as OptiX does on setUserData/map+memcpy); no exceptions
 * cross this boundary -- every function returns MOPTIX_OK (0) or a negative
 * error code and moptix_last_error() gives the text.
 *
 * Built as minimaloptix_amd/lib/libmoptix.so (hipcc --offload-arch=gfx950).
 * There is NO CPU fallback behind this ABI: without a gfx950 device every
 * compute entry point fails with MOPTIX_ERR_NO_DEVICE.
 */
#ifndef MOPTIX_H
#define MOPTIX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOPTIX_OK                 0
#define MOPTIX_ERR_INVALID       -1   /* bad argument / bad handle                 */
#define MOPTIX_ERR_NO_DEVICE     -2   /* no HIP device / not gfx950                 */
#define MOPTIX_ERR_HIP           -3   /* HIP runtime error (text in last_error)     */
#define MOPTIX_ERR_STATE         -4   /* call order (e.g. launch before build_accel)*/
#define MOPTIX_ERR_LIMIT         -5   /* scene exceeds a compiled limit             */
#define MOPTIX_ERR_COMM          -6   /* a collective did not complete (peer missing / communicator error); the communicator was aborted */

typedef struct moptix_context_t* moptix_context;

/* ---- records crossing host<->device (Structures.h) ----------------------- */
typedef struct moptix_float3 { float x, y, z; } moptix_float3;
typedef struct moptix_float4 { float x, y, z, w; } moptix_float4;

/* Structures.h:12-20 CamParams */
typedef struct moptix_cam_params {
  moptix_float3 origin, horizontal, vertical, scrLowerLeftCorner, u, v;
  float lensRadius;
} moptix_cam_params;

/* Structures.h:22-26 SphereParams (velocity is host-side animation state, unused on device) */
typedef struct moptix_sphere_params { float radius; moptix_float3 center; moptix_float3 velocity; } moptix_sphere_params;

/* Structures.h:28-33 QuadParams, as produced by setQuadParams (utils_host.cpp:67-75).
 * Unpadded here; the reference's float4 alignment padding is not part of the ABI. */
typedef struct moptix_quad_params { moptix_float4 plane; moptix_float3 v1, v2, anchor; } moptix_quad_params;

/* which closest-hit program the material carries (Material.cu:28,49,72,118,238) */
enum { MOPTIX_MAT_LAMBERTIAN = 0, MOPTIX_MAT_METAL = 1, MOPTIX_MAT_GLASS = 2, MOPTIX_MAT_DISNEY = 3, MOPTIX_MAT_LIGHT = 4 };
enum { MOPTIX_BRDF_NORMAL = 0, MOPTIX_BRDF_GLASS = 1 };   /* Structures.h:49 BrdfType   */
enum { MOPTIX_LIGHT_SPHERE = 0, MOPTIX_LIGHT_QUAD = 1 };  /* Structures.h:68 LightShape */

/* Structures.h:51-66 DisneyParams */
typedef struct moptix_disney_params {
  int32_t albedoID;              /* 0 = RT_TEXTURE_ID_NULL, else an id from moptix_add_texture */
  moptix_float3 color, emission;
  float metallic, subsurface, specular, roughness, specularTint, anisotropic;
  float sheen, sheenTint, clearcoat, clearcoatGloss;
  int32_t brdfType;
} moptix_disney_params;

/* Structures.h:70-80 LightParams */
typedef struct moptix_light_params {
  moptix_float3 position, normal, emission, u, v;
  float area, radius;
  int32_t shape;
} moptix_light_params;

/* One material = createMaterial + setClosestHitProgram(0, kind) [+ setAnyHitProgram(1,
 * disneyAnyHit) for MOPTIX_MAT_DISNEY] + setUserData of the matching params
 * (Structures.h:35-47 Lambertian/Metal/Glass, :51 Disney, :70 Light.emission). */
typedef struct moptix_material {
  int32_t kind;
  moptix_float3 albedo;          /* LambertianParams / MetalParams / GlassParams .albedo */
  float fuzz;                    /* MetalParams.fuzz   */
  float refIdx;                  /* GlassParams.refIdx */
  moptix_float3 emission;        /* light material: LightParams.emission */
  moptix_disney_params disney;   /* MOPTIX_MAT_DISNEY */
} moptix_material;

/* Context variables (MinimalOptiX.cpp:136-151), miss bgColor (:165...), raygen
 * camParams (:256...) and the launch size (:546). */
typedef struct moptix_params {
  uint32_t width, height;        /* fixedWidth/fixedHeight, MinimalOptiX.h:82-83 */
  uint32_t rayMaxDepth;          /* 256   MinimalOptiX.h:85 */
  float rayMinIntensity;         /* 0.001 MinimalOptiX.h:88 */
  float rayEpsilonT;             /* 0.001 MinimalOptiX.h:89 */
  moptix_float3 bgColor;         /* staticMiss bgColor, miss.cu:7 */
  moptix_cam_params cam;
} moptix_params;

/* Counters of a counting launch (moptix_render_counted): the algorithmic-byte
 * model of SURVEY 8(d) is evaluated from these. */
typedef struct moptix_stats {
  uint64_t samples;              /* camera samples traced                                 */
  uint64_t primaryRays, bounceRays, shadowRays;
  uint64_t nodeFetches;          /* four-child BVH nodes fetched: 64-byte records where get_option "node_format_used" says 64
                                    (variant 4 on scenes whose paths favour the quantised form), else 128-byte ones */
  uint64_t triTests;             /* 48-byte triangle records tested                       */
  uint64_t closestHits;          /* closest-hit shading fetches                           */
  uint64_t lightLoads;           /* LightParams records read for NEE                      */
  uint64_t analyticTests;        /* sphere/quad records tested (brute-force lists)        */
  uint64_t traversalSteps;       /* wave-level loop iterations (x64 lanes = lane slots)   */
  uint64_t activeLaneSteps;      /* lanes doing useful work summed over those iterations  */
  uint64_t shadeBatches;         /* shading / regeneration batches run (queue kernels)     */
  uint64_t shadeBatchLanes;      /* slots processed by those batches                       */
} moptix_stats;

typedef struct moptix_accel_info {
  uint32_t nTriangles, nNodes, maxLeafSize, treeDepth;
  float buildMs;                 /* wall time of the last acceleration build between two HIP events on the launch stream: its kernels plus the
                                    host round trips between them (the binned-SAH builder reads one node count per level back) */
  uint64_t nodeBytes, triBytes;  /* nodeBytes: both forms of the nodes, nNodes x (128 + 64) */
} moptix_accel_info;

/* ---- lifecycle: Context::create()/setRayTypeCount/setEntryPointCount/setStackSize
 *      (MinimalOptiX.cpp:131-134) --------------------------------------------- */
int moptix_create(moptix_context* out, int device);
int moptix_destroy(moptix_context ctx);
const char* moptix_last_error(moptix_context ctx);           /* ctx may be NULL: last global error */
const char* moptix_version(void);
/* run every launch on this hipStream_t (default: a stream the context owns) */
int moptix_set_stream(moptix_context ctx, void* hipStream);

/* ---- parameters: context[...]->set* (MinimalOptiX.cpp:136-151,165,256) ---- */
int moptix_set_params(moptix_context ctx, const moptix_params* p);

/* ---- scene upload (MinimalOptiX.cpp:168-249, 362-537, 780-843) ------------ */
int moptix_clear_scene(moptix_context ctx);
/* createTextureSampler + createBuffer(RT_BUFFER_INPUT, RT_FORMAT_FLOAT4, w, h) + sampler->getId()
 * (MinimalOptiX.cpp:444-479): RT_WRAP_REPEAT, normalized coordinates, RT_FILTER_LINEAR.  rgba holds
 * 4*w*h floats, row 0 = texture coordinate v 0 (the caller has already flipped the image as
 * MinimalOptiX.cpp:464 does).  *outTexId >= 1 goes into DisneyParams.albedoID; add textures before the
 * materials that name them. */
int moptix_add_texture(moptix_context ctx, const float* rgba, int32_t width, int32_t height, int32_t* outTexId);
int moptix_add_material(moptix_context ctx, const moptix_material* m, int32_t* outMatId);
/* createGeometry+sphereIntersect/sphereBBox+createGeometryInstance (MinimalOptiX.cpp:177-208,796-843) */
int moptix_add_spheres(moptix_context ctx, const moptix_sphere_params* s, const int32_t* matIds, int32_t n);
/* ...quadIntersect/quadBBox (MinimalOptiX.cpp:210-240, 505-510, 780-794) */
int moptix_add_quads(moptix_context ctx, const moptix_quad_params* q, const int32_t* matIds, int32_t n);
/* one tinyobj shape = one Geometry with six buffers (MinimalOptiX.cpp:392-441, Geometry.cu:114-119).
 * normals/texcoords may be NULL (count 0); nIdx/tIdx may be NULL or hold -1. */
int moptix_add_mesh(moptix_context ctx,
                    const float* positions, int32_t nVerts,
                    const float* normals, int32_t nNormals,
                    const float* texcoords, int32_t nTexcoords,
                    const int32_t* vIdx, const int32_t* nIdx, const int32_t* tIdx, int32_t nFaces,
                    int32_t matId);
/* context["lights"] buffer of LightParams (MinimalOptiX.cpp:523-531) */
int moptix_set_lights(moptix_context ctx, const moptix_light_params* lights, int32_t n);
/* rewrite sphere i..i+n (updateVideo, MinimalOptiX.cpp:763-764) */
int moptix_update_spheres(moptix_context ctx, int32_t first, const moptix_sphere_params* s, int32_t n);

/* setAcceleration("NoAccel" | "Trbvh") (MinimalOptiX.cpp:248,378,494,534,748).
 * "Trbvh" -> built on the device over all triangles: Morton order, then (option "builder" = 1, the default) a binned
 * surface-area-heuristic topology over that order, or (= 0) the Karras radix tree; four-wide 128-byte nodes either way.
 * Analytic primitives always stay in brute-force lists. "NoAccel" with triangles present
 * is rejected (MOPTIX_ERR_INVALID). */
int moptix_build_accel(moptix_context ctx, const char* kind);
int moptix_get_accel_info(moptix_context ctx, moptix_accel_info* out);
/* context->validate() (MinimalOptiX.cpp:542) */
int moptix_validate(moptix_context ctx);

/* ---- launch --------------------------------------------------------------- */
/* context["randSeed"]->setInt(seed); context->launch(0,W,H)  (MinimalOptiX.cpp:545-546,774-775):
 * one sample per pixel, accuBuffer += clamp(color,0,1); blocking. */
int moptix_launch(moptix_context ctx, int32_t randSeed);
/* the same nSeeds launches fused into ONE kernel (per pixel the samples are still
 * added in seed order, so the result is bit-identical to nSeeds moptix_launch calls). */
int moptix_render(moptix_context ctx, const int32_t* seeds, int32_t nSeeds);
/* non-blocking variant for stream overlap; moptix_sync() waits. */
int moptix_render_async(moptix_context ctx, const int32_t* seeds, int32_t nSeeds);
int moptix_sync(moptix_context ctx);
/* same work with the in-kernel counters enabled (slower; not for timing) */
int moptix_render_counted(moptix_context ctx, const int32_t* seeds, int32_t nSeeds, moptix_stats* out);

/* Multi-GPU tile split (new): this context renders one 8x8-pixel tile of every
 * group of nRanks tiles (raster order): tile g * nRanks + (rank + g) % nRanks of group g (the deal
 * rotates from group to group so that no rank owns whole columns).  Default (0,1) = whole frame. */
int moptix_set_partition(moptix_context ctx, int32_t rank, int32_t nRanks);

/* Multi-GPU collectives (new; SURVEY 8e, north_star "RCCL gather over xGMI"): one process per GPU, the context owns an
 * RCCL communicator.  One rank calls moptix_comm_unique_id (ncclGetUniqueId) and hands the 128 bytes to the others by
 * whatever the host has (a file, MPI, torch.distributed, a socket); every rank then calls moptix_comm_init
 * (collective: returns when all nRanks have called it).  Where the library offers it the communicator is NON-BLOCKING
 * (ncclCommInitRankConfig with config.blocking = 0; option "comm_blocking" = 1 asks for plain ncclCommInitRank): every RCCL call
 * then returns at once -- ncclInProgress while the library is still at work on the host, e.g. bringing a peer's connections up --
 * and this layer polls the communicator's state against "comm_timeout_ms".  With a blocking communicator a peer that is alive but
 * never makes its call holds the caller INSIDE ncclSend / ncclGroupEnd, where no deadline can reach.
 *   moptix_gather_tiles : tile split (moptix_set_partition(rank, nRanks) as the communicator's) -- every other rank packs its
 *                         tiles and ncclSend()s them to dstRank, which receives them in one group and writes them into its
 *                         accuBuffer on the device: dstRank then holds the whole frame, bit-identical to a one-GPU render.
 *   moptix_reduce_frame : sample split -- ncclReduce(sum) of the accuBuffers into dstRank's.
 * Both block until the data has landed -- at most "comm_timeout_ms" (option; default 120 s), host side (non-blocking communicator:
 * the call has not settled) and device side (its kernels have not finished) alike: a collective that has not completed by then (a peer
 * died or never called), or whose communicator reports an asynchronous error, is ABORTED (ncclCommAbort) and the call returns
 * MOPTIX_ERR_COMM; the host should exit.  The context stays usable as a one-rank context -- unless the dead collective's kernels
 * never leave its stream (a library without ncclCommAbort: the communicator is abandoned, not destroyed, and if the stream is still
 * busy after 10 s every later call on the context returns MOPTIX_ERR_COMM).  With a communicator of one rank they are no-ops.
 * moptix_pack_tiles / moptix_unpack_tiles are the device-side halves of the gather (rank r's tiles of the accuBuffer <->
 * a dense buffer of moptix_packed_tile_floats(nRanks) floats in work-item order), exposed so that the partition can be
 * tested on one GPU. */
#define MOPTIX_COMM_ID_BYTES 128
int moptix_comm_unique_id(uint8_t* id128);
int moptix_comm_init(moptix_context ctx, const uint8_t* id128, int32_t rank, int32_t nRanks);
int moptix_comm_destroy(moptix_context ctx);
int moptix_gather_tiles(moptix_context ctx, int32_t dstRank);
int moptix_reduce_frame(moptix_context ctx, int32_t dstRank);
int moptix_packed_tile_floats(moptix_context ctx, int32_t nRanks, uint64_t* outFloats);
int moptix_pack_tiles(moptix_context ctx, int32_t rank, int32_t nRanks, float* dstDevice);
int moptix_unpack_tiles(moptix_context ctx, int32_t rank, int32_t nRanks, const float* srcDevice);

/* The one option that is NOT a tuning knob:
 *   "shadow_rule"      1 (default): a shadow ray is decided by its NEAREST any-hit surface -- an opaque Disney surface gives (0,0,0), a
 *                      Disney GLASS surface gives its colour and nothing behind it is looked at.  This is DEVIATION D5' from SURVEY A2
 *                      ("an opaque surface anywhere on the segment blocks, every glass surface crossed multiplies"): it models what
 *                      OptiX does with disneyAnyHit (Material.cu:225-232 accepts the glass hit, which ends the ray's interval) under a
 *                      front-to-back traversal, and it is FITTED, not pinned: the evidence is the floor round the machine in
 *                      demo/coffee.png with a stand-in for the missing pot; real Trbvh traversal is not strictly front
 *                      to back.  0: SURVEY A2's order-independent rule (the oracle's switch shadow_any_opaque_blocks).  The two differ
 *                      only in scenes with a Disney GLASS material (the benchmark scene has none); takes effect at the next render.
 * tuning knobs.  The scheduler knobs (kernel_variant, slots_in_use, aux_depth, analytic_queue, tile_major, blocks_per_cu, swap_lanes,
 * starve_lanes, exit_threshold, leaf_threshold, sample_buffer_mb) never change a bit of the image.  The tree-shaping knobs (builder,
 * leaf_size, node_format) are bit-stable with ONE documented exception: the reference's float triangle test (Geometry.cu:121-160) can
 * accept a grazing hit on a needle triangle at a point outside that triangle's own bounding box, and whether any traversal ever tests
 * that triangle then depends on the boxes around it -- 1 pixel-sample in 3,600 fuzz cases (profiles/r04_fuzz.txt).
 * BOUND of that exception, as observed and as tests/test_gpu_parity.py::test_the_known_grazing_hit_is_tree_dependent_and_nothing_else replays it:
 * ONE pixel-sample of a frame takes another path (one closest hit more or less), every other pixel-sample keeps its bits; the frame's RMSE
 * against the oracle stays below north_star's 1e-3 (8e-4 on a 200x112 frame at 2 spp, i.e. below 1e-5 at any benchmark size).  It needs a
 * needle triangle and a ray within ~2e-4 of its plane; the benchmark scenes have shown none in 7,200 + 2,400 fuzz cases.
 *   "kernel_variant"   0 per-lane kernel, 3 path slots and queues shared by the workgroup (variants 1 and 2 of rounds 1-2 are gone),
 *                      4 = 3 with one shading visit per bounce (pt_packet.h; scenes with <= 3 lights, else 3 runs),
 *                      -1 (default) = the library's choice per launch: 4 for launches of >= 1e6 samples and >= 16 seeds
 *                      ("auto_packet" = 0 turns that off), else 3; scenes without triangles: see "analytic_queue" (they run a lean
 *                      instantiation of 3, four workgroups per CU).  get_option returns -1 while the choice is the library's
 *   "builder"          1 binned-SAH topology over the Morton order (default), 0 Morton radix tree
 *   "node_format"      variant 4: the node record the trace kernel fetches -- 128 = four child boxes in binary32 (one L2 line), 64 =
 *                      the same boxes on a 256-step grid over the node's box, rounded outwards (half a line: 4 L1 look-ups per
 *                      node step instead of 7, boxes up to a grid step larger), 0 (default) = whichever is cheaper for this scene
 *                      seen from this camera: decided at the first render after a build by walking one path per pixel of a
 *                      128-pixel-wide grid under both (get_option "node_format_used" tells the verdict)
 *   "slots_in_use"     path slots of a workgroup's pool (get_option "path_slots": 576 for variant 4, 512 for variant 3) that carry a
 *                      path (-1 = chosen per launch: 7/8 of them for variant 4 launches under 1e8 samples, else all); the others
 *                      are what deep paths borrow, see "aux_depth"
 *   "aux_depth"        variant 4: a path this deep (default 16; 0 = never) traces the shadow rays of each hit in slots
 *                      borrowed from finished paths, at the same time as the continuation ray (DESIGN.md "Borrowed slots")
 *   "analytic_queue"   scenes without triangles: 1 = through the queue kernel, 0 = per-lane kernel, -1 (default) = queue
 *                      kernel from 64 primitives on
 *   "leaf_size"        1..8 triangles per BVH leaf (default 4; takes effect at the next build_accel)
 *   "tile_major"       hand-out order of the (pixel, sample) work items: 0 = sample-major, 1 = all samples of an
 *                      8x8 tile back to back, tiles with the deepest paths of earlier launches first,
 *                      2 = as 1 with one pixel's samples per wave, 3 = all samples of a pixel back to back, pixels with
 *                      the deepest paths of earlier launches first (default)
 *   "blocks_per_cu"    resident workgroups per CU (default 3)
 *   "swap_lanes", "starve_lanes"   node-loop swap / starvation thresholds of variants 3 and 4
 *   "exit_threshold", "leaf_threshold" (variant 0)
 *   "sample_buffer_mb" budget of the per-sample buffer (default 16384); larger batches run in passes
 *   "fast_shading"     0 (default): disneyPdf / disneyEval in correctly rounded binary32, bit-parity with the oracle;
 *                      1: hardware reciprocal / square-root approximations there (the reference itself is built with
 *                      -use_fast_math, utils_host.cpp:30-32): same rays, BRDF weights within ~1e-6, default kernel only
 *   "watchdog_ms"      wall-clock bound of one render kernel (default 600000); a pass cut short is not accumulated
 *   "comm_timeout_ms"  deadline of moptix_comm_init / moptix_gather_tiles / moptix_reduce_frame (default 120000): when the call has not
 *                      settled on the host or its kernels have not completed by then, or the communicator reports an asynchronous
 *                      error, the communicator is aborted and the call returns MOPTIX_ERR_COMM instead of blocking the rank for good
 *   "forget_history"   (write-only, any value) drop the per-pixel depth history that orders the work items ("tile_major"): the next launch
 *                      is ordered like the first one of a context -- what a single-frame render sees
 *   "comm_blocking"    1 = moptix_comm_init makes a blocking communicator even where a non-blocking one is available (default 0)
 *   "drain_below"      variant 4: a workgroup of the trace kernel that is down to this many paths (default 64, 0 = never) hands them
 *                      to the drain kernel (csrc/drainkernel.hip: a wave per up to 16 paths, all lanes on one frontier) at their next
 *                      packet boundary and leaves; same image bits, ray and hit counts either way -- the node / triangle-test counts
 *                      of a counted launch then vary a little from run to run (the frontier's visiting order follows its atomics).  The one
 *                      documented exception to order independence -- a grazing hit outside its triangle's own box, below -- can in principle
 *                      make a drained path's result depend on that order; no such case in 3,300 fuzz cases with the drain kernel on
 * read-only (get_option): "kernel_variant_used", "node_format_used", "path_slots", "num_cus", "comm_ranks" (size of the context's
 *   communicator, 0 without one), "comm_nonblocking_used" (1: that communicator is non-blocking), and after moptix_render_counted
 *   "counted_span_us" (first wave in -> last wave out of the trace kernel) / "counted_tail_us" (the part of it after the last
 *   work item was handed out: the launch's drain; -1 for the per-lane kernel)
 * Unknown names -> MOPTIX_ERR_INVALID. */
int moptix_set_option(moptix_context ctx, const char* name, int32_t value);
int moptix_get_option(moptix_context ctx, const char* name, int32_t* value);

/* ---- output buffer: createBuffer(RT_BUFFER_INPUT_OUTPUT, FLOAT3, W, H) + map()/unmap()
 *      (MinimalOptiX.cpp:144-147, 44-60).  float3 row-major, row 0 = bottom row. -- */
int moptix_accum_read(moptix_context ctx, float* dstHost);          /* W*H*3 floats */
int moptix_accum_clear(moptix_context ctx);
int moptix_accum_device_ptr(moptix_context ctx, void** devPtr);     /* for device-side gather/reduce */
/* render into caller-owned device memory (e.g. a torch tensor) instead; NULL restores the own buffer */
int moptix_accum_bind(moptix_context ctx, void* devPtr);
/* updateContent (MinimalOptiX.cpp:43-66) on the device: out[(H-1-y)*W+x] = u8(clamp(accu/n,0,1)*255),
 * optionally clearing the accumulator. dstHost: W*H*3 bytes, row 0 = top. */
int moptix_resolve_rgb8(moptix_context ctx, float nAccumulation, int clearBuffer, uint8_t* dstHost);

/* ---- measurement ----------------------------------------------------------- */
/* device time (HIP events on the launch stream) of the trace kernel -- the dominant kernel --
 * and the number of its launches since the last reset */
int moptix_kernel_time(moptix_context ctx, double* totalMs, uint64_t* nLaunches, int reset);
/* device time of the ordered sample reductions that followed those launches */
int moptix_reduce_time(moptix_context ctx, double* totalMs);

/* debug/validation: copy the built BVH to host (nodes: nNodes*128 B four-child nodes, tris: nTriangles*48 B,
 * triPrimIds: nTriangles int32 = original face index of each record). Any pointer may be NULL. */
int moptix_debug_read_accel(moptix_context ctx, void* nodes, void* tris, int32_t* triPrimIds);
/* the same nodes in the 64-byte form the trace kernels fetch (nNodes*64 B: corner, grid steps, 24 plane bytes, the four
 * child references -- csrc/pt_types.h Node64); node i of this array stands for node i of moptix_debug_read_accel's. */
int moptix_debug_read_nodes64(moptix_context ctx, void* nodes64);
/* nearest-hit query for n rays (BVH-vs-brute-force tests): rays = n x {ox,oy,oz,dx,dy,dz,tmin,tmax};
 * outT[n], outPrim[n] (prim id: spheres, quads, triangles; -1 = miss). */
int moptix_debug_trace(moptix_context ctx, const float* rays, int32_t n, float* outT, int32_t* outPrim);

#ifdef __cplusplus
}
#endif
#endif /* MOPTIX_H */
