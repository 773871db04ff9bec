// obj_loader.h -- Wavefront .obj ingest for the MinimalOptiX file scenes.
//
// The reference vendors tiny_obj_loader.h v1.4.0 and consumes exactly this part of its
// interface (MinimalOptiX.cpp:380-442): attrib_t{vertices,normals,texcoords},
// shape_t.mesh.{indices[].{vertex,normal,texcoord}_index, num_face_vertices} and
// LoadObj(&attrib,&shapes,&materials,&warn,&err,filename) with default triangulation.
// This is an independent, smaller implementation of that contract (geometry statements
// only: v, vn, vt, f, g, o; materials are parsed past), so the ingest code above it
// reads like the reference's.
#pragma once
#include <string>
#include <vector>

namespace mobj {

typedef float real_t;

struct index_t { int vertex_index; int normal_index; int texcoord_index; };   // -1 when absent

struct mesh_t {
  std::vector<index_t> indices;                 // 3 per triangle after triangulation
  std::vector<unsigned char> num_face_vertices; // always 3 when triangulate == true
};

struct shape_t { std::string name; mesh_t mesh; };

struct attrib_t {
  std::vector<real_t> vertices;   // xyz
  std::vector<real_t> normals;    // xyz
  std::vector<real_t> texcoords;  // uv
};

struct material_t { std::string name; };        // accepted and ignored (the reference never reads it)

// Returns false (and fills *err) when the file cannot be opened or a face is malformed.
bool LoadObj(attrib_t* attrib, std::vector<shape_t>* shapes, std::vector<material_t>* materials,
             std::string* warn, std::string* err, const char* filename, bool triangulate = true);

}  // namespace mobj
