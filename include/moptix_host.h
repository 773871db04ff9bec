/*
 * moptix_host.h -- C entry points of the host-side library (libmoptix_host.so): scene
 * ingest (.scene/.obj), the built-in scene builders and the MinimalOptiX::renderScene
 * entry, for callers that cannot use the C++ class of minimaloptix_amd/host/minimal_optix.h
 * directly (the Python tests/bench, or a C program).
 *
 * Replaces, on the reference side: MinimalOptiX::setupScene()/setupScene(name)/setUpVideo
 * (MinimalOptiX.cpp:154-538, 607-759), Scene (scene.cpp:5-124), tinyobj::LoadObj and
 * MinimalOptiX::renderScene (MinimalOptiX.cpp:540-560).
 */
#ifndef MOPTIX_HOST_H
#define MOPTIX_HOST_H

#include "moptix.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mohost_scene_t* mohost_scene;

typedef struct mohost_scene_sizes {
  int32_t nMaterials, nSpheres, nQuads, nLights, nVerts, nNormals, nTexcoords, nFaces, nMeshes, nWarnings;
  int32_t nTextures;
} mohost_scene_sizes;

const char* mohost_last_error(void);

/* kind: "spheres" (farg = aperture; MinimalOptiX.cpp:156-257)
 *       "file:<name>"  coffee|bedroom|diningroom|stormtrooper|spaceship|cornell|hyperion|dragon
 *                      (setupScene(name) + camera of MinimalOptiX.cpp:258-353; baseFolder = "scenes/")
 *       "random_spheres" (iarg = nSpheres; setUpVideo, MinimalOptiX.cpp:607-759)
 *       "cornell_quads" | "dining_standin" (iarg = copies) | "million_standin" (iarg = triangles)
 *       "coffee_pot_standin" (coffee + a lathe stand-in for the missing glass pot Mesh010.obj) */
int mohost_scene_build(const char* kind, const char* baseFolder, uint32_t width, uint32_t height,
                       int32_t iarg, float farg, int skipMissing, mohost_scene* out);
void mohost_scene_free(mohost_scene s);
int mohost_scene_get_sizes(mohost_scene s, mohost_scene_sizes* out);
int mohost_scene_get_params(mohost_scene s, moptix_params* out, float aabbMin[3], float aabbMax[3], char accel[16]);
const char* mohost_scene_warning(mohost_scene s, int32_t i);
/* Flattened copy (meshes concatenated into one vertex/normal pool, indices rebased;
 * nIdx = -1 where a face has no normals).  Any destination may be NULL. */
int mohost_scene_copy(mohost_scene s, moptix_material* materials,
                      moptix_sphere_params* spheres, int32_t* sphereMat,
                      moptix_quad_params* quads, int32_t* quadMat, moptix_light_params* lights,
                      float* positions, float* normals, int32_t* vIdx, int32_t* nIdx, int32_t* faceMat);
/* texcoords (2*nTexcoords floats, meshes concatenated) and the rebased per-face indices (3*nFaces, -1 = none) */
int mohost_scene_copy_texcoords(mohost_scene s, float* texcoords, int32_t* tIdx);
/* texture i (< nTextures; DisneyParams.albedoID - 1): size and, if rgba != NULL, its 4*w*h floats as uploaded
 * (MinimalOptiX.cpp:459-472: row 0 = bottom image row, alpha 1) */
int mohost_scene_texture(mohost_scene s, int32_t i, int32_t* width, int32_t* height, float* rgba);
/* QImage(path) stand-in used for albedoTex files: PNG (non-interlaced), baseline JPEG and binary PNM -> 8-bit RGB, row 0 = top.
 * rgb may be NULL to query the size. */
int mohost_read_image(const char* path, int32_t* width, int32_t* height, uint8_t* rgb, uint64_t rgbCapacity);
/* clear_scene + set_params + add_texture + add_* + set_lights + build_accel on ctx */
int mohost_scene_upload(mohost_scene s, moptix_context ctx);

/* utils_host.cpp:67-99 */
void mohost_set_quad_params(const float anchor[3], const float v1[3], const float v2[3], moptix_quad_params* out);
void mohost_set_cam_params(const float lookFrom[3], const float lookAt[3], const float up[3],
                           float vFoV, float aspect, float aperture, float focus, moptix_cam_params* out);

/* animation of the spheres scene (MinimalOptiX::move/animate, MinimalOptiX.cpp:562-592): advances the n spheres
 * by `time` and *angle by time*5; mohost_video_camera is the orbit camera of updateVideo (:766-767) */
void mohost_animate_spheres(moptix_sphere_params* spheres, int32_t n, float time, float* angle);
void mohost_video_camera(float angle, float aspect, moptix_cam_params* out);

/* .obj ingest on its own (for loader tests): returns face count or <0 */
int mohost_obj_stats(const char* path, int32_t* nVerts, int32_t* nNormals, int32_t* nTexcoords, int32_t* nShapes);

/* MinimalOptiX::renderScene(autoSave, prefix) for one scene id on `device`.
 * sceneId: the enum of minimal_optix.h.  canvasRGB8 (W*H*3, row 0 = top) may be NULL. */
typedef struct mohost_render_result {
  double renderMs; float bvhBuildMs; uint64_t nVertices, nFaces; uint32_t nNodes, treeDepth;
} mohost_render_result;
int mohost_render_scene(int device, int sceneId, const char* baseFolder, uint32_t width, uint32_t height,
                        uint32_t nSuperSampling, uint32_t baseSeed, int autoSave, const char* fileNamePrefix,
                        const char* outputDir, uint8_t* canvasRGB8, mohost_render_result* result);

#ifdef __cplusplus
}
#endif
#endif
