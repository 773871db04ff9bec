#include "obj_loader.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace mobj {
namespace {

inline const char* skip_ws(const char* p) { while (*p == ' ' || *p == '\t') ++p; return p; }
inline bool at_eol(const char* p) { return *p == '\0' || *p == '\n' || *p == '\r' || *p == '#'; }

// tiny_obj parses into double and narrows to real_t; do the same with the C library.
inline real_t parse_real(const char*& p, real_t dflt = 0.0f) {
  p = skip_ws(p);
  if (at_eol(p)) return dflt;
  char* end = nullptr;
  double v = strtod(p, &end);
  if (end == p) return dflt;
  p = end;
  return static_cast<real_t>(v);
}

// OBJ indices are 1-based; negative = relative to the current end of the array.
inline bool fix_index(int idx, int n, int* out) {
  if (idx > 0) { *out = idx - 1; return true; }
  if (idx < 0) { *out = n + idx; return *out >= 0; }
  return false;   // 0 is not allowed
}

// one "v", "v/vt", "v//vn" or "v/vt/vn" corner
inline bool parse_corner(const char*& p, int nv, int nvn, int nvt, index_t* out) {
  out->vertex_index = out->normal_index = out->texcoord_index = -1;
  char* end = nullptr;
  long v = strtol(p, &end, 10);
  if (end == p || !fix_index((int)v, nv, &out->vertex_index)) return false;
  p = end;
  if (*p != '/') return true;
  ++p;
  if (*p == '/') {                       // v//vn
    ++p;
    long n = strtol(p, &end, 10);
    if (end == p || !fix_index((int)n, nvn, &out->normal_index)) return false;
    p = end;
    return true;
  }
  long t = strtol(p, &end, 10);           // v/vt[/vn]
  if (end == p || !fix_index((int)t, nvt, &out->texcoord_index)) return false;
  p = end;
  if (*p != '/') return true;
  ++p;
  long n = strtol(p, &end, 10);
  if (end == p || !fix_index((int)n, nvn, &out->normal_index)) return false;
  p = end;
  return true;
}

inline std::string rest_of_line(const char* p) {
  p = skip_ws(p);
  std::string s(p);
  while (!s.empty() && (s.back() == '\n' || s.back() == '\r' || s.back() == ' ' || s.back() == '\t')) s.pop_back();
  return s;
}

}  // namespace

bool LoadObj(attrib_t* attrib, std::vector<shape_t>* shapes, std::vector<material_t>* materials,
             std::string* warn, std::string* err, const char* filename, bool triangulate) {
  attrib->vertices.clear(); attrib->normals.clear(); attrib->texcoords.clear();
  shapes->clear();
  if (materials) materials->clear();
  if (warn) warn->clear();
  if (err) err->clear();

  FILE* fp = fopen(filename, "rb");
  if (!fp) {
    if (err) *err = std::string("Cannot open file [") + filename + "]";
    return false;
  }

  shape_t cur;
  auto flush_shape = [&]() {
    if (!cur.mesh.indices.empty()) shapes->push_back(cur);
    cur = shape_t();
  };

  std::vector<index_t> corners;
  std::string line;
  char buf[4096];
  bool ok = true;
  size_t lineNo = 0;
  while (ok && fgets(buf, sizeof(buf), fp)) {
    line.assign(buf);
    while (!line.empty() && line.back() != '\n' && fgets(buf, sizeof(buf), fp)) line.append(buf);   // long lines
    ++lineNo;
    const char* p = skip_ws(line.c_str());
    if (at_eol(p)) continue;

    if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
      p += 2;
      real_t x = parse_real(p), y = parse_real(p), z = parse_real(p);
      attrib->vertices.push_back(x); attrib->vertices.push_back(y); attrib->vertices.push_back(z);
    } else if (p[0] == 'v' && p[1] == 'n' && (p[2] == ' ' || p[2] == '\t')) {
      p += 3;
      real_t x = parse_real(p), y = parse_real(p), z = parse_real(p);
      attrib->normals.push_back(x); attrib->normals.push_back(y); attrib->normals.push_back(z);
    } else if (p[0] == 'v' && p[1] == 't' && (p[2] == ' ' || p[2] == '\t')) {
      p += 3;
      real_t u = parse_real(p), v = parse_real(p);
      attrib->texcoords.push_back(u); attrib->texcoords.push_back(v);
    } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
      p += 2;
      corners.clear();
      const int nv = (int)(attrib->vertices.size() / 3), nvn = (int)(attrib->normals.size() / 3), nvt = (int)(attrib->texcoords.size() / 2);
      for (;;) {
        p = skip_ws(p);
        if (at_eol(p)) break;
        index_t c;
        if (!parse_corner(p, nv, nvn, nvt, &c)) {
          if (err) *err = std::string("Malformed face at line ") + std::to_string(lineNo) + " of " + filename;
          ok = false;
          break;
        }
        corners.push_back(c);
      }
      if (!ok) break;
      if (corners.size() < 3) continue;    // degenerate statement: skipped, as tiny_obj does
      if (triangulate) {                   // fan: (0, k-1, k)
        for (size_t k = 2; k < corners.size(); ++k) {
          cur.mesh.indices.push_back(corners[0]);
          cur.mesh.indices.push_back(corners[k - 1]);
          cur.mesh.indices.push_back(corners[k]);
          cur.mesh.num_face_vertices.push_back(3);
        }
      } else {
        for (const index_t& c : corners) cur.mesh.indices.push_back(c);
        cur.mesh.num_face_vertices.push_back((unsigned char)corners.size());
      }
    } else if ((p[0] == 'g' || p[0] == 'o') && (p[1] == ' ' || p[1] == '\t' || at_eol(p + 1))) {
      flush_shape();                       // a group/object statement closes the running shape
      cur.name = rest_of_line(p + 1);
    } else if (!strncmp(p, "usemtl", 6) && materials) {
      // material assignment per face is not consumed by the render path
    } else if (!strncmp(p, "mtllib", 6)) {
      if (warn && warn->empty()) *warn = "material libraries are ignored";
    }
    // everything else (s, l, p, curves, ...) is ignored
  }
  fclose(fp);
  flush_shape();
  return ok;
}

}  // namespace mobj
