// obj_dump.cpp -- TEST INFRASTRUCTURE.  Loads one .obj and writes what the loader returned as a flat binary dump, so
// that the repo's own loader (minimaloptix_amd/host/obj_loader.cpp, namespace mobj) can be compared with the
// reference's vendored tiny_obj_loader.h v1.4.0 (MinimalOptiX/tiny_obj_loader.h, header-only, MIT) on the same files.
//   -DUSE_REFERENCE_TINYOBJ : compiled against the header where it lies under /root/reference (oracle/Makefile target
//                             _ref/tinyobj_dump; only in the build container), output binary under oracle/_ref/
//   otherwise               : compiled against obj_loader.h (oracle/mobj_dump)
// Dump: int32 counts {nVerts*3, nNormals*3, nTexcoords*2, nShapes}, the three float arrays, then per shape
// {nIndices, nFaces} followed by (vertex, normal, texcoord) index triples and the per-face vertex counts.
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>
#if defined(USE_REFERENCE_TINYOBJ)
#define TINYOBJLOADER_IMPLEMENTATION
#include "tiny_obj_loader.h"
namespace L = tinyobj;
#else
#include "../minimaloptix_amd/host/obj_loader.h"
namespace L = mobj;
#endif

int main(int argc, char** argv) {
  if (argc < 3) { fprintf(stderr, "usage: %s in.obj out.bin\n", argv[0]); return 2; }
  L::attrib_t attrib; std::vector<L::shape_t> shapes; std::vector<L::material_t> materials; std::string warn, err;
  const bool ok = L::LoadObj(&attrib, &shapes, &materials, &warn, &err, argv[1]);   // MinimalOptiX.cpp:385
  if (!ok || !err.empty()) { fprintf(stderr, "load failed: %s\n", err.c_str()); return 1; }
  FILE* f = fopen(argv[2], "wb");
  if (!f) return 3;
  const int32_t head[4] = { (int32_t)attrib.vertices.size(), (int32_t)attrib.normals.size(), (int32_t)attrib.texcoords.size(), (int32_t)shapes.size() };
  fwrite(head, 4, 4, f);
  fwrite(attrib.vertices.data(), 4, attrib.vertices.size(), f);
  fwrite(attrib.normals.data(), 4, attrib.normals.size(), f);
  fwrite(attrib.texcoords.data(), 4, attrib.texcoords.size(), f);
  for (const auto& s : shapes) {
    const int32_t h2[2] = { (int32_t)s.mesh.indices.size(), (int32_t)s.mesh.num_face_vertices.size() };
    fwrite(h2, 4, 2, f);
    for (const auto& i : s.mesh.indices) { const int32_t t[3] = { i.vertex_index, i.normal_index, i.texcoord_index }; fwrite(t, 4, 3, f); }
    for (unsigned char c : s.mesh.num_face_vertices) { const int32_t v = c; fwrite(&v, 4, 1, f); }
  }
  fclose(f);
  return 0;
}
