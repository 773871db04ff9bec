// jpeg_read.cpp -- JPEG decoder (8-bit, Huffman; sequential and progressive) behind readImage(): QImage reads albedoTex
// files through libjpeg, so this follows libjpeg's decoder arithmetic -- the "islow" integer IDCT, triangle
// ("fancy") chroma upsampling and the fixed-point YCbCr->RGB tables -- and reproduces its pixels exactly
// (tests compare with libjpeg-turbo through PIL).  A progressive file (SOF2; ITU-T T.81 Annex G) is a sequence of scans
// that each deliver a band of coefficients (spectral selection Ss..Se) at some precision (successive approximation
// Ah / Al): the coefficients of every block are kept until the last scan and transformed once, which is what libjpeg
// does with a complete file (its block smoothing only applies while refinement scans are still missing).
// Arithmetic-coded, 12-bit and CMYK files are reported as unsupported.
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace moptix {
namespace {

struct HuffTable {
  bool present = false;
  uint8_t bits[17] = { 0 }; uint8_t vals[256] = { 0 };
  int mincode[17], maxcode[18], valptr[17];
  void build() {
    int code = 0, k = 0;
    for (int l = 1; l <= 16; l++) {
      valptr[l] = k; mincode[l] = code;
      code += bits[l]; k += bits[l];
      maxcode[l] = bits[l] ? code - 1 : -1;
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
  }
};

struct Component { int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0; int wBlocks = 0, hBlocks = 0, dw = 0, dh = 0; int pred = 0; std::vector<uint8_t> px; int stride = 0;
                   std::vector<int16_t> coefs; };      // progressive: 64 coefficients (natural order) per block, wBlocks x hBlocks

struct BitSrc {
  const uint8_t* p; size_t n, pos; uint32_t buf = 0; int cnt = 0; bool hitMarker = false;
  int bit() {
    if (cnt == 0) {
      if (pos >= n) { hitMarker = true; return 0; }
      uint8_t b = p[pos];
      if (b == 0xff) {
        if (pos + 1 < n && p[pos + 1] == 0x00) pos += 2;
        else { hitMarker = true; return 0; }           // a marker: feed zeros (libjpeg does the same)
      } else pos++;
      buf = b; cnt = 8;
    }
    cnt--;
    return (int)((buf >> cnt) & 1u);
  }
  int bits(int k) { int v = 0; for (int i = 0; i < k; i++) v = (v << 1) | bit(); return v; }
  void reset() { buf = 0; cnt = 0; hitMarker = false; }
};

int decodeHuff(BitSrc& bs, const HuffTable& t) {
  int code = 0;
  for (int l = 1; l <= 16; l++) {
    code = (code << 1) | bs.bit();
    if (t.maxcode[l] >= 0 && code <= t.maxcode[l] && code >= t.mincode[l]) return t.vals[t.valptr[l] + code - t.mincode[l]];
  }
  return -1;
}
inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v - (1 << s) + 1 : v; }

const uint8_t kZigzag[64] = { 0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
                              35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63 };

// jidctint.c (accurate integer IDCT): CONST_BITS 13, PASS1_BITS 2
inline int32_t descale(int32_t x, int n) { return (x + (1 << (n - 1))) >> n; }
inline uint8_t clamp255(int32_t v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
void idctIslow(const int16_t* coef, const uint16_t* q, uint8_t* out, int stride) {
  const int32_t F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137, F1961 = 16069,
                F2053 = 16819, F2562 = 20995, F3072 = 25172;
  int32_t ws[64];
  for (int c = 0; c < 8; c++) {
    const int16_t* in = coef + c; const uint16_t* qq = q + c; int32_t* w = ws + c;
    if (!in[8] && !in[16] && !in[24] && !in[32] && !in[40] && !in[48] && !in[56]) {
      const int32_t dc = (int32_t)in[0] * qq[0] * 4;       // << PASS1_BITS
      for (int r = 0; r < 8; r++) w[8 * r] = dc;
      continue;
    }
    int32_t z2 = in[16] * qq[16], z3 = in[48] * qq[48];
    int32_t z1 = (z2 + z3) * F0541;
    int32_t tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
    z2 = in[0] * qq[0]; z3 = in[32] * qq[32];
    int32_t tmp0 = (z2 + z3) * 8192, tmp1 = (z2 - z3) * 8192;
    const int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[56] * qq[56]; tmp1 = in[40] * qq[40]; tmp2 = in[24] * qq[24]; tmp3 = in[8] * qq[8];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; int32_t z4 = tmp1 + tmp3;
    const int32_t z5 = (z3 + z4) * F1175;
    tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
    z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    w[0] = descale(tmp10 + tmp3, 11); w[56] = descale(tmp10 - tmp3, 11);
    w[8] = descale(tmp11 + tmp2, 11); w[48] = descale(tmp11 - tmp2, 11);
    w[16] = descale(tmp12 + tmp1, 11); w[40] = descale(tmp12 - tmp1, 11);
    w[24] = descale(tmp13 + tmp0, 11); w[32] = descale(tmp13 - tmp0, 11);
  }
  for (int r = 0; r < 8; r++) {
    const int32_t* w = ws + 8 * r; uint8_t* o = out + (size_t)r * stride;
    if (!w[1] && !w[2] && !w[3] && !w[4] && !w[5] && !w[6] && !w[7]) {
      const uint8_t dc = clamp255(descale(w[0], 5) + 128);
      for (int c = 0; c < 8; c++) o[c] = dc;
      continue;
    }
    int32_t z2 = w[2], z3 = w[6];
    int32_t z1 = (z2 + z3) * F0541;
    int32_t tmp2 = z1 + z3 * (-F1847), tmp3 = z1 + z2 * F0765;
    int32_t tmp0 = (w[0] + w[4]) * 8192, tmp1 = (w[0] - w[4]) * 8192;
    const int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; int32_t z4 = tmp1 + tmp3;
    const int32_t z5 = (z3 + z4) * F1175;
    tmp0 *= F0298; tmp1 *= F2053; tmp2 *= F3072; tmp3 *= F1501;
    z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    o[0] = clamp255(descale(tmp10 + tmp3, 18) + 128); o[7] = clamp255(descale(tmp10 - tmp3, 18) + 128);
    o[1] = clamp255(descale(tmp11 + tmp2, 18) + 128); o[6] = clamp255(descale(tmp11 - tmp2, 18) + 128);
    o[2] = clamp255(descale(tmp12 + tmp1, 18) + 128); o[5] = clamp255(descale(tmp12 - tmp1, 18) + 128);
    o[3] = clamp255(descale(tmp13 + tmp0, 18) + 128); o[4] = clamp255(descale(tmp13 - tmp0, 18) + 128);
  }
}

inline uint16_t be16(const uint8_t* p) { return (uint16_t)((p[0] << 8) | p[1]); }

// jdsample.c h2v1_fancy_upsample: one row of `n` samples -> 2n samples
void upsampleH2Fancy(const uint8_t* in, int n, uint8_t* out) {
  if (n == 1) { out[0] = out[1] = in[0]; return; }
  out[0] = in[0]; out[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
  for (int i = 1; i < n - 1; i++) {
    const int v = in[i] * 3;
    out[2 * i] = (uint8_t)((v + in[i - 1] + 1) >> 2); out[2 * i + 1] = (uint8_t)((v + in[i + 1] + 2) >> 2);
  }
  out[2 * n - 2] = (uint8_t)((in[n - 1] * 3 + in[n - 2] + 1) >> 2); out[2 * n - 1] = in[n - 1];
}
// jdsample.c h2v2_fancy_upsample: near row `a`, far row `b` (the neighbour above or below) -> one output row of 2n
void upsampleH2V2Fancy(const uint8_t* a, const uint8_t* b, int n, uint8_t* out) {
  if (n == 1) { const int t = a[0] * 3 + b[0]; out[0] = (uint8_t)((t * 4 + 8) >> 4); out[1] = (uint8_t)((t * 4 + 7) >> 4); return; }
  int thiscol = a[0] * 3 + b[0], nextcol = a[1] * 3 + b[1], lastcol;
  out[0] = (uint8_t)((thiscol * 4 + 8) >> 4); out[1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
  lastcol = thiscol; thiscol = nextcol;
  for (int i = 1; i < n - 1; i++) {
    nextcol = a[i + 1] * 3 + b[i + 1];
    out[2 * i] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4); out[2 * i + 1] = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
    lastcol = thiscol; thiscol = nextcol;
  }
  out[2 * n - 2] = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4); out[2 * n - 1] = (uint8_t)((thiscol * 4 + 7) >> 4);
}

// ---- progressive scans (T.81 G.1.2): one call per block of the scan; eobrun = blocks of this band still to be skipped ----
// first pass over the DC coefficient: the difference to the previous block's value, delivered with its low Al bits missing
bool progDcFirst(BitSrc& bs, const HuffTable& t, int al, int& pred, int16_t* blk) {
  const int s = decodeHuff(bs, t);
  if (s < 0 || s > 11) return false;
  pred += s ? extend(bs.bits(s), s) : 0;
  blk[0] = (int16_t)(pred * (1 << al));
  return true;
}
// a later pass over the DC coefficient: one more bit of it
void progDcRefine(BitSrc& bs, int al, int16_t* blk) { if (bs.bit()) blk[0] = (int16_t)(blk[0] | (1 << al)); }
// first pass over the band ss..se of the AC coefficients: run / size pairs as in a sequential scan, values scaled by 2^Al; a
// size of zero with a run r < 15 announces 2^r + (r more bits) blocks whose band holds nothing more, this one included
bool progAcFirst(BitSrc& bs, const HuffTable& t, int ss, int se, int al, int& eobrun, int16_t* blk) {
  if (eobrun > 0) { eobrun--; return true; }
  for (int k = ss; k <= se; k++) {
    const int rs = decodeHuff(bs, t);
    if (rs < 0) return false;
    const int r = rs >> 4, sz = rs & 15;
    if (sz) {
      k += r;
      if (k > se) return false;
      blk[kZigzag[k]] = (int16_t)(extend(bs.bits(sz), sz) * (1 << al));
    } else if (r == 15) {
      k += 15;
    } else {
      eobrun = (1 << r) - 1;
      if (r) eobrun += bs.bits(r);
      break;
    }
  }
  return true;
}
// a later pass over the band: every coefficient that is already non-zero gets one correction bit (its magnitude grows by 2^Al
// when the bit is set and that bit of it is still clear), and new coefficients of magnitude 2^Al appear after runs of r
// coefficients that STAY zero -- the correction bits of the non-zero ones passed on the way are interleaved with them
bool progAcRefine(BitSrc& bs, const HuffTable& t, int ss, int se, int al, int& eobrun, int16_t* blk) {
  const int plus = 1 << al, minus = -(1 << al);
  auto correct = [&](int16_t& c) { if (bs.bit() && (c & plus) == 0) c = (int16_t)(c + (c >= 0 ? plus : minus)); };
  int k = ss;
  if (eobrun == 0) {
    for (; k <= se; k++) {
      const int rs = decodeHuff(bs, t);
      if (rs < 0) return false;
      int r = rs >> 4; const int sz = rs & 15;
      int newVal = 0;
      if (sz) {
        if (sz != 1) return false;                       // a new coefficient of this pass has magnitude 2^Al, nothing else
        newVal = bs.bit() ? plus : minus;
      } else if (r != 15) {
        eobrun = 1 << r;
        if (r) eobrun += bs.bits(r);
        break;                                           // the rest of this block's band: correction bits only (below)
      }
      for (; k <= se; k++) {
        int16_t& c = blk[kZigzag[k]];
        if (c != 0) correct(c);
        else if (--r < 0) break;                         // the (r+1)-th zero: where the new coefficient goes (or, for ZRL, where the run of 16 ends)
      }
      if (newVal && k <= se) blk[kZigzag[k]] = (int16_t)newVal;
    }
  }
  if (eobrun > 0) {
    for (; k <= se; k++) { int16_t& c = blk[kZigzag[k]]; if (c != 0) correct(c); }
    eobrun--;
  }
  return true;
}

}  // namespace

bool decodeJPEG(const std::vector<uint8_t>& file, int& width, int& height, std::vector<uint8_t>& rgb, std::string& err) {
  uint16_t qt[4][64]; bool haveQ[4] = { false, false, false, false };
  HuffTable dcT[4], acT[4];
  std::vector<Component> comps;
  int W = 0, H = 0, hmax = 1, vmax = 1, restartInterval = 0;
  bool haveFrame = false, adobe = false, progressive = false; int adobeTransform = -1;
  size_t pos = 2;
  auto fail = [&](const char* m) { err = m; return false; };
  while (pos + 4 <= file.size()) {
    if (file[pos] != 0xff) { pos++; continue; }
    const int marker = file[pos + 1];
    if (marker == 0xff) { pos++; continue; }
    if (marker == 0xd8 || marker == 0x01 || (marker >= 0xd0 && marker <= 0xd7)) { pos += 2; continue; }
    if (marker == 0xd9) break;
    const size_t len = be16(&file[pos + 2]);
    if (len < 2 || pos + 2 + len > file.size()) return fail("truncated JPEG segment");
    const uint8_t* d = &file[pos + 4]; const size_t dl = len - 2;
    if (marker == 0xdb) {                                                       // DQT
      size_t i = 0;
      while (i < dl) {
        const int pq = d[i] >> 4, tq = d[i] & 15; i++;
        if (tq > 3 || i + (pq ? 128 : 64) > dl) return fail("bad JPEG quantisation table");
        for (int k = 0; k < 64; k++) { qt[tq][kZigzag[k]] = pq ? be16(d + i + 2 * k) : d[i + k]; }
        i += pq ? 128 : 64; haveQ[tq] = true;
      }
    } else if (marker == 0xc4) {                                                // DHT
      size_t i = 0;
      while (i + 17 <= dl) {
        const int tc = d[i] >> 4, th = d[i] & 15; i++;
        if (tc > 1 || th > 3) return fail("bad JPEG Huffman table");
        HuffTable& t = tc ? acT[th] : dcT[th];
        int total = 0;
        for (int l = 1; l <= 16; l++) { t.bits[l] = d[i + l - 1]; total += t.bits[l]; }
        i += 16;
        if (total > 256 || i + total > dl) return fail("bad JPEG Huffman table");
        memcpy(t.vals, d + i, total); i += total;
        t.build(); t.present = true;
      }
    } else if (marker == 0xc0 || marker == 0xc1 || marker == 0xc2) {            // SOF0 / SOF1 (8-bit sequential Huffman), SOF2 (progressive)
      progressive = marker == 0xc2;
      if (dl < 6 || d[0] != 8) return fail("only 8-bit JPEG is supported");
      H = be16(d + 1); W = be16(d + 3);
      const int nc = d[5];
      if (W <= 0 || H <= 0 || W > 32768 || H > 32768 || (nc != 1 && nc != 3) || dl < (size_t)(6 + 3 * nc)) return fail("unsupported JPEG frame (1 or 3 components)");
      comps.resize(nc);
      for (int c = 0; c < nc; c++) {
        comps[c].id = d[6 + 3 * c]; comps[c].h = d[7 + 3 * c] >> 4; comps[c].v = d[7 + 3 * c] & 15; comps[c].tq = d[8 + 3 * c] & 3;
        if (comps[c].h < 1 || comps[c].h > 2 || comps[c].v < 1 || comps[c].v > 2) return fail("unsupported JPEG sampling factors");
        hmax = comps[c].h > hmax ? comps[c].h : hmax; vmax = comps[c].v > vmax ? comps[c].v : vmax;
      }
      const int mcuW = 8 * hmax, mcuH = 8 * vmax, mcusX = (W + mcuW - 1) / mcuW, mcusY = (H + mcuH - 1) / mcuH;
      for (Component& c : comps) {
        c.wBlocks = mcusX * c.h; c.hBlocks = mcusY * c.v;
        c.dw = (W * c.h + hmax - 1) / hmax; c.dh = (H * c.v + vmax - 1) / vmax;
        c.stride = c.wBlocks * 8; c.px.assign((size_t)c.stride * c.hBlocks * 8, 0);
        if (progressive) c.coefs.assign((size_t)c.wBlocks * c.hBlocks * 64, 0);
      }
      haveFrame = true;
    } else if (marker >= 0xc3 && marker <= 0xcf && marker != 0xc4 && marker != 0xc8 && marker != 0xcc) { return fail("unsupported JPEG coding process");
    } else if (marker == 0xdd) { if (dl >= 2) restartInterval = be16(d);
    } else if (marker == 0xee) { if (dl >= 12 && !memcmp(d, "Adobe", 5)) { adobe = true; adobeTransform = d[11]; }
    } else if (marker == 0xda) {                                                // SOS + entropy-coded data
      if (!haveFrame) return fail("JPEG scan before frame header");
      const int ns = d[0];
      if (ns < 1 || ns > (int)comps.size() || dl < (size_t)(1 + 2 * ns + 3)) return fail("bad JPEG scan header");
      std::vector<Component*> sc;
      for (int i = 0; i < ns; i++) {
        Component* cp = nullptr;
        for (Component& c : comps) if (c.id == d[1 + 2 * i]) cp = &c;
        if (!cp) return fail("JPEG scan names an unknown component");
        cp->td = d[2 + 2 * i] >> 4; cp->ta = d[2 + 2 * i] & 15;
        sc.push_back(cp);
      }
      const int ss = d[1 + 2 * ns], se = d[2 + 2 * ns], ah = d[3 + 2 * ns] >> 4, al = d[3 + 2 * ns] & 15;
      for (Component* cp : sc) {                         // a progressive scan needs only the tables of the band it carries
        const bool needDc = !progressive || (ss == 0 && ah == 0), needAc = !progressive || ss > 0;
        if (cp->td > 3 || cp->ta > 3 || (needDc && !dcT[cp->td].present) || (needAc && !acT[cp->ta].present) || !haveQ[cp->tq]) return fail("JPEG scan uses a missing table");
      }
      if (progressive && (ss > se || se > 63 || al > 13 || (ss == 0 && se != 0) || (ss > 0 && ns != 1))) return fail("bad progressive JPEG scan parameters");
      BitSrc bs{ file.data(), file.size(), pos + 2 + len };
      for (Component& c : comps) c.pred = 0;
      const bool interleaved = ns > 1;
      const int mcuW = 8 * hmax, mcuH = 8 * vmax;
      int mcusX, mcusY;
      if (interleaved) { mcusX = (W + mcuW - 1) / mcuW; mcusY = (H + mcuH - 1) / mcuH; }
      else { mcusX = (sc[0]->dw + 7) / 8; mcusY = (sc[0]->dh + 7) / 8; }        // a single-component scan covers only real blocks
      int toRestart = restartInterval;
      int16_t coef[64];
      int eobrun = 0;
      for (int my = 0; my < mcusY; my++) for (int mx = 0; mx < mcusX; mx++) {
        if (restartInterval && toRestart == 0) {
          bs.reset();
          while (bs.pos + 1 < bs.n && !(bs.p[bs.pos] == 0xff && bs.p[bs.pos + 1] >= 0xd0 && bs.p[bs.pos + 1] <= 0xd7)) bs.pos++;
          bs.pos += 2;
          for (Component* c : sc) c->pred = 0;
          eobrun = 0;
          toRestart = restartInterval;
        }
        toRestart--;
        for (Component* c : sc) {
          const int bh = interleaved ? c->h : 1, bv = interleaved ? c->v : 1;
          for (int by = 0; by < bv; by++) for (int bx = 0; bx < bh; bx++) {
            if (progressive) {
              const int blockX = mx * bh + bx, blockY = my * bv + by;
              int16_t scratch[64];                      // a block outside the component's array (cannot happen for a well-formed scan) still consumes its bits
              int16_t* blk = (blockX < c->wBlocks && blockY < c->hBlocks) ? &c->coefs[((size_t)blockY * c->wBlocks + blockX) * 64] : (memset(scratch, 0, sizeof(scratch)), scratch);
              bool ok = true;
              if (ss == 0) { if (ah == 0) ok = progDcFirst(bs, dcT[c->td], al, c->pred, blk); else progDcRefine(bs, al, blk); }
              else if (ah == 0) ok = progAcFirst(bs, acT[c->ta], ss, se, al, eobrun, blk);
              else ok = progAcRefine(bs, acT[c->ta], ss, se, al, eobrun, blk);
              if (!ok) return fail("corrupt progressive JPEG data");
              continue;
            }
            memset(coef, 0, sizeof(coef));
            int s = decodeHuff(bs, dcT[c->td]);
            if (s < 0 || s > 11) return fail("corrupt JPEG data (DC)");
            c->pred += s ? extend(bs.bits(s), s) : 0;
            coef[0] = (int16_t)c->pred;
            for (int k = 1; k < 64;) {
              const int rs = decodeHuff(bs, acT[c->ta]);
              if (rs < 0) return fail("corrupt JPEG data (AC)");
              const int r = rs >> 4, sz = rs & 15;
              if (sz == 0) { if (r == 15) { k += 16; continue; } break; }
              k += r;
              if (k > 63) return fail("corrupt JPEG data (run)");
              coef[kZigzag[k]] = (int16_t)extend(bs.bits(sz), sz);
              k++;
            }
            const int blockX = mx * bh + bx, blockY = my * bv + by;
            if (blockX < c->wBlocks && blockY < c->hBlocks)
              idctIslow(coef, qt[c->tq], &c->px[(size_t)blockY * 8 * c->stride + (size_t)blockX * 8], c->stride);
          }
        }
      }
      pos = bs.pos;
      continue;
    }
    pos += 2 + len;
  }
  if (!haveFrame) return fail("no JPEG frame");
  if (progressive)                                     // every scan has been read: one inverse transform per block
    for (Component& c : comps) {
      if (!haveQ[c.tq]) return fail("JPEG component without a quantisation table");
      for (int by = 0; by < c.hBlocks; by++) for (int bx = 0; bx < c.wBlocks; bx++)
        idctIslow(&c.coefs[((size_t)by * c.wBlocks + bx) * 64], qt[c.tq], &c.px[(size_t)by * 8 * c.stride + (size_t)bx * 8], c.stride);
    }
  width = W; height = H;
  rgb.assign((size_t)W * H * 3, 0);
  if (comps.size() == 1) {
    for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
      const uint8_t v = comps[0].px[(size_t)y * comps[0].stride + x];
      uint8_t* o = &rgb[((size_t)y * W + x) * 3]; o[0] = o[1] = o[2] = v;
    }
    return true;
  }
  // upsample every component to full resolution (jdsample.c: fancy upsampling for 2:1 horizontally and 2:1 both ways)
  std::vector<std::vector<uint8_t>> full(3);
  for (int ci = 0; ci < 3; ci++) {
    Component& c = comps[ci];
    const int fw = c.dw * (hmax / c.h);
    full[ci].assign((size_t)(fw > W ? fw : W) * H + 16, 0);
    const int ostride = fw > W ? fw : W;
    if (c.h == hmax && c.v == vmax) {
      for (int y = 0; y < H; y++) memcpy(&full[ci][(size_t)y * ostride], &c.px[(size_t)y * c.stride], W);
    } else if (c.h * 2 == hmax && c.dw <= 2 && (c.v == vmax || c.v * 2 == vmax)) {
      // jdsample.c jinit_upsampler: the triangle filter is only used when downsampled_width > 2, else replication
      for (int y = 0; y < H; y++) {
        const uint8_t* in = &c.px[(size_t)(c.v == vmax ? y : y >> 1) * c.stride];
        for (int x = 0; x < 2 * c.dw; x++) full[ci][(size_t)y * ostride + x] = in[x >> 1];
      }
    } else if (c.h * 2 == hmax && c.v == vmax) {
      for (int y = 0; y < H; y++) upsampleH2Fancy(&c.px[(size_t)y * c.stride], c.dw, &full[ci][(size_t)y * ostride]);
    } else if (c.h * 2 == hmax && c.v * 2 == vmax) {
      for (int y = 0; y < H; y++) {
        const int r = y >> 1;
        int nb = (y & 1) ? r + 1 : r - 1;                       // the neighbour row below / above; edge rows are repeated
        if (nb < 0) nb = 0;
        if (nb > c.dh - 1) nb = c.dh - 1;
        upsampleH2V2Fancy(&c.px[(size_t)r * c.stride], &c.px[(size_t)nb * c.stride], c.dw, &full[ci][(size_t)y * ostride]);
      }
    } else return fail("unsupported JPEG chroma subsampling");
    c.stride = ostride;
  }
  const bool ycc = adobe ? adobeTransform == 1 : true;           // JFIF / no marker: YCbCr; Adobe transform 0: RGB
  // jdcolor.c build_ycc_rgb_table: SCALEBITS 16
  int crR[256], cbB[256]; int32_t crG[256], cbG[256];
  for (int i = 0; i < 256; i++) {
    const int x = i - 128;
    crR[i] = (int)((91881 * x + 32768) >> 16); cbB[i] = (int)((116130 * x + 32768) >> 16);
    crG[i] = -46802 * x; cbG[i] = -22554 * x + 32768;
  }
  for (int y = 0; y < H; y++) for (int x = 0; x < W; x++) {
    const int Y = full[0][(size_t)y * comps[0].stride + x], cb = full[1][(size_t)y * comps[1].stride + x], cr = full[2][(size_t)y * comps[2].stride + x];
    uint8_t* o = &rgb[((size_t)y * W + x) * 3];
    if (ycc) { o[0] = clamp255(Y + crR[cr]); o[1] = clamp255(Y + (int)((cbG[cb] + crG[cr]) >> 16)); o[2] = clamp255(Y + cbB[cb]); }
    else { o[0] = (uint8_t)Y; o[1] = (uint8_t)cb; o[2] = (uint8_t)cr; }
  }
  return true;
}

}  // namespace moptix
