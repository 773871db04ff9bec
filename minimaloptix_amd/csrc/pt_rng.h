// pt_rng.h -- per-path RNG of the reference (utils_device.h:8-52): TEA-16 seeding and a
// 24-bit LCG.  Integer arithmetic: bit-exact with the oracle by construction.
#pragma once
#include "pt_math.h"

namespace pt {

// utils_device.h:8-22  tea<16>
PT_HD uint32_t tea16(uint32_t val0, uint32_t val1) {
  uint32_t v0 = val0, v1 = val1, s0 = 0;
#pragma unroll
  for (int n = 0; n < 16; n++) {
    s0 += 0x9e3779b9u;
    v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
    v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
  }
  return v0;
}
// utils_device.h:24-34  lcg / rand : state is the path's `randSeed`
PT_HD uint32_t lcg(uint32_t& s) { s = 1664525u * s + 1013904223u; return s & 0x00FFFFFFu; }
PT_HD float rnd(uint32_t& s) { return (float)lcg(s) / (float)0x01000000; }
// utils_device.h:36-43 (draw order x,y,z; always at least one attempt)
PT_HD v3 rand_in_unit_sphere(uint32_t& s) {
  v3 res;
  do {
    float a = rnd(s); float b = rnd(s); float c = rnd(s);
    res = mk3(a, b, c) * 2.0f - mk3(1.f, 1.f, 1.f);
  } while (length(res) >= 1.0f);
  return res;
}
// utils_device.h:45-52
PT_HD v3 rand_in_unit_disk(uint32_t& s) {
  v3 res;
  do {
    float a = rnd(s); float b = rnd(s);
    res = mk3(a, b, 0.f) * 2.0f - mk3(1.f, 1.f, 0.f);
  } while (length(res) >= 1.0f);
  return res;
}
// utils_device.h:192-198 folkPayload: child.randSeed = tea<16>(parent.randSeed, parent.depth+1)
PT_HD uint32_t fork_seed(uint32_t parentSeed, int childDepth) { return tea16(parentSeed, (uint32_t)childDepth); }

}  // namespace pt
