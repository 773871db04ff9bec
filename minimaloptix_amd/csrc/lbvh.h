// lbvh.h -- device LBVH builder interface (implementation: lbvh.hip)
#pragma once
#include <hip/hip_runtime.h>
#include "pt_types.h"

namespace pt {

struct LbvhResult {
  Node128* nodes = nullptr;    // device, nNodes (four-wide)
  Node64* nodes64 = nullptr;   // device, nNodes: the same nodes compressed (pt_lbvh.h compress_node), or nullptr if a node is too wide for the grid
  Tri48* tris = nullptr;       // device, nTris (sorted / leaf order)
  TriShade* shade = nullptr;   // device, nTris
  int nTris = 0, nNodes = 0, rootRef = kEmptyRef, depth = 0, leafSize = 0;   // depth: levels of four-wide nodes
  int stackBound = 0;          // most entries a traversal stack can hold (3 per level + 1)
  float buildMs = 0.f;
};

// facePos: 9 floats per face (p0 p1 p2) in upload order; faceNrm: 9 per face; dFaceMat: face_mat_word(material id, SHADOW_*)
// per face (pt_types.h); device pointers.
// Allocates the result arrays with hipMalloc (caller frees with lbvh_free).
// builder: 0 = Morton radix tree (Karras 2012), 1 = binned-SAH topology over the Morton order (pt_lbvh.h)
hipError_t lbvh_build(hipStream_t stream, const float* dFacePos, const float* dFaceNrm, const int* dFaceHasNrm,
                      const int* dFaceMat, int nFaces, int leafSize, int builder, LbvhResult* out);
void lbvh_free(LbvhResult* r);

}  // namespace pt
