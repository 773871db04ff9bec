// pt_packet.h -- the path state machine with one shading visit per BOUNCE instead of one per ray.
//
// The reference's disney program (Material.cu:170-221) interleaves, per hit: for every light {draw a point, trace the
// shadow ray, add its contribution}, then sample the BRDF and trace the continuation ray.  pt_path.h follows that order
// ray by ray, so a path of depth k is a chain of up to 4k dependent rays, each with its own shading visit.  Nothing in
// that chain needs the shadow results early: the traces consume no random numbers (the shadow payload's seed is a fork
// that is never read, Material.cu:191), and the light-side weight of a shadow ray (MIS weight x BRDF x emission / pdf)
// is known before it is traced.  So here a visit does everything that does not depend on a trace --
//     all light draws in light order, the facing tests, disneyPdf / disneyEval per facing light, the BRDF sample,
//     its pdf / eval and the seed fork
// -- and leaves a PACKET of up to kPacketShadows shadow rays plus the continuation ray, all from one origin.  The next
// visit first folds the shadow results into the radiance in light order with the throughput the hit had (same float
// operations in the same order as pt_path.h's on_result for shadow rays), then applies the continuation's weight and
// shades its hit.  Random draws, decisions and every floating-point operation on the path's values are those of
// pt_path.h; only the moment at which the traces happen differs.  Used by kernel variant 4 (packetkernel.hip); scenes with
// more than kPacketShadows lights run on variant 3.
#pragma once
#include "pt_path.h"

namespace pt {

constexpr int kPacketShadows = 3;

struct Packet {
  int nShadow;                         // shadow rays of this packet, in light order
  v3 sd[kPacketShadows]; float stmax[kPacketShadows];      // their directions and tmax (origin: ps.o, tmin: epsT)
  v3 pendW[kPacketShadows]; float pendInv[kPacketShadows]; // weight of each: c = (pendW * attenuation) * pendInv
  int hasBounce;                       // a continuation ray follows (ps.o, ps.d, radiance ray)
  int hasScale;                        // its weight is still to be applied: thr = (thr * bscale) * binv
  v3 bscale; float binv;
};
PT_HD void packet_clear(Packet& pk) {
  pk.nShadow = 0; pk.hasBounce = 0; pk.hasScale = 0; pk.bscale = mk3(1.f, 1.f, 1.f); pk.binv = 1.f;
  for (int i = 0; i < kPacketShadows; i++) { pk.sd[i] = mk3(0.f, 0.f, 1.f); pk.stmax[i] = 0.f; pk.pendW[i] = mk3(0.f, 0.f, 0.f); pk.pendInv[i] = 0.f; }
}

// slot j of the packet, without indexing the arrays by a run-time value (they are meant to live in registers)
PT_HD void packet_set_shadow(Packet& pk, int j, v3 d, float tmax, v3 w, float inv) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int k = 0; k < kPacketShadows; k++)
    if (k == j) { pk.sd[k] = d; pk.stmax[k] = tmax; pk.pendW[k] = w; pk.pendInv[k] = inv; }
}

// Where a new packet's shadow rays go: into the Packet (host mirror, tests) ...
struct PacketSink {
  Packet& pk;
  PT_HD void shadow(int j, v3 d, float tmax, v3 w, float inv) const { packet_set_shadow(pk, j, d, tmax, w, inv); }
};
// ... or wherever the caller keeps them (packetkernel.hip writes each one to the slot record as soon as it is known,
// so that the three of them are not held in registers across the BRDF evaluations that follow).

// Material.cu:172-221 for one hit, without the traces (ps.N, ps.V, ps.mat, ps.o set by on_result; ps.light == 0).
template <bool CNT, bool FAST = false, class Sink = PacketSink>
PT_HD void on_lights_packet(const SceneView& sc, PathState& ps, Packet& pk, Counters& ct, const Sink& sink) {
  const DevMaterial m = load_const(at32(sc.mats, ps.mat));
  packet_clear(pk);
  const Onb onb = make_onb(ps.N);
  const DisneyView dv = disney_view<FAST>(m, onb, ps.V);      // shared by the evaluations of this hit (up to three lights + the bounce)
  v3 Cdlin = m.Cdlin, Cspec0 = m.Cspec0, Csheen = m.Csheen;
  if (m.albedoTex != 0) {
    Cdlin = ps.cdlin;
    disney_color_constants(Cdlin, m.specular, m.specularTint, m.sheenTint, m.metallic, Cspec0, Csheen);
  }
  for (int li = 0; li < sc.nLights; li++) {                 // li is uniform across the wave: the record is a scalar load
    const DevLight ltv = load_uniform(sc.lights + li);
    const DevLight* lt = &ltv;
    cnt<CNT>(ct.lightLoads); census<CNT>(ct, CR_LIGHT_DRAW);
    v3 pointOnLight, normalOnLight;
    if (lt->shape == LIGHT_SPHERE) {
      pointOnLight = lt->position + rand_in_unit_sphere(ps.seed) * lt->radius;
      normalOnLight = normalize(pointOnLight - lt->position);
    } else {
      const float r1 = rnd(ps.seed); const float r2 = rnd(ps.seed);
      pointOnLight = (lt->position + lt->u * r1) + lt->v * r2;
      normalOnLight = lt->normal;
    }
    v3 L = pointOnLight - ps.o;
    const float lightDst = length(L);
    L = normalize(L);
    if (dot(L, ps.N) > 0.f && dot(L, normalOnLight) < 0.f && pk.nShadow < kPacketShadows) {
      census<CNT>(ct, li < 2 ? CR_LIGHT0 + li : CR_LIGHT2);
      const v3 H = normalize(L + ps.V);
      const float lightPdf = lightDst * lightDst / lt->area / dot(normalOnLight, -L);
      const float pdf = disney_pdf<FAST>(m, ps.N, L, H);
      const v3 brdf = disney_eval<FAST>(m, Cdlin, Cspec0, Csheen, ps.N, dv, L, H);
      v3 w = mk3(0.f, 0.f, 0.f); float inv = 0.f;
      if (lightPdf > 0 && pdf > 0) {
        w = (brdf * powerHeuristic(lightPdf, pdf)) * lt->emission;
        inv = 1.0f / fmaxf_(0.001f, lightPdf);
      }
      sink.shadow(pk.nShadow, L, lightDst - sc.epsT, w, inv);
      pk.nShadow++;
      cnt<CNT>(ct.shadowRays);
    }
  }
  v3 L, H;
  census<CNT>(ct, CR_BOUNCE_SAMPLE);
  disney_sample(ps.seed, m, onb, ps.V, L, H);
  if (dot(ps.N, L) > 0.0f && dot(ps.N, ps.V) > 0.0f) {
    census<CNT>(ct, CR_BOUNCE_EVAL);
    const float pdf = disney_pdf<FAST>(m, ps.N, L, H);
    const v3 brdf = disney_eval<FAST>(m, Cdlin, Cspec0, Csheen, ps.N, dv, L, H);
    const uint32_t childSeed = fork_seed(ps.seed, ps.depth + 1);
    if (pdf > 0) {
      pk.bscale = brdf; pk.binv = 1.0f / pdf; pk.hasScale = 1; pk.hasBounce = 1;
      cnt<CNT>(ct.bounceRays);
      bounce(sc, ps, ps.o, L, childSeed);
    }
  }
  if (pk.nShadow == 0 && !pk.hasBounce) { end_sample(ps); return; }
  ps.mode = M_TRACE;
}

// The packet has been traced: att[i] = attenuation of shadow ray i (disneyAnyHit), tv = nearest hit of the continuation.
// Runs until the path owns a new packet (ps.mode == M_TRACE) or the sample has ended (M_NEW_SAMPLE).
template <bool CNT, bool FAST = false, class Sink = PacketSink>
PT_HD void on_result_packet(const SceneView& sc, PathState& ps, Packet& pk, const Trav& tv, const v3 att[kPacketShadows], Counters& ct,
                            const Sink& sink) {
  census<CNT>(ct, CR_RESULT);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
  for (int i = 0; i < kPacketShadows; i++) {                      // Material.cu:193-201, light order
    if (i < pk.nShadow && pk.pendInv[i] != 0.f && length_is_nonzero(att[i])) {
      census<CNT>(ct, CR_SHADOW_FOLD);
      const v3 c = (pk.pendW[i] * att[i]) * pk.pendInv[i];
      ps.rad = ps.rad + ps.thr * c;
    }
  }
  if (!pk.hasBounce) { end_sample(ps); return; }
  if (pk.hasScale) ps.thr = (ps.thr * pk.bscale) * pk.binv;      // Material.cu:217-219 indirect = brdf * child / pdf
  ps.kind = RK_RADIANCE;
  on_result<CNT>(sc, ps, tv, ct);
  if (ps.mode == M_LIGHTS) { on_lights_packet<CNT, FAST, Sink>(sc, ps, pk, ct, sink); return; }
  if (ps.mode == M_TRACE) { packet_clear(pk); pk.hasBounce = 1; }       // glass / lambertian / metal: the continuation only
}

// first packet of a sample: the camera ray
PT_HD void packet_primary(Packet& pk) { packet_clear(pk); pk.hasBounce = 1; }

}  // namespace pt
