// image_read.cpp -- decoder behind readImage(): what QImage(path) + pixelColor() give the reference's
// texture upload (MinimalOptiX.cpp:446-466).  The image has no zlib/libpng headers, so inflate
// (RFC 1951) and the PNG container (filters, bit depths, palette; plain and Adam7-interlaced) are
// implemented here; binary PNM (P5/P6) is read as well, baseline and progressive JPEG in jpeg_read.cpp.
#include "image_io.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace moptix {
namespace {

// ---------------------------------------------------------------- inflate ------------------
struct BitReader {
  const uint8_t* p; size_t n; size_t pos = 0; uint32_t buf = 0; int cnt = 0; bool eof = false;
  int bit() {
    if (cnt == 0) {
      if (pos >= n) { eof = true; return 0; }
      buf = p[pos++]; cnt = 8;
    }
    const int b = (int)(buf & 1u); buf >>= 1; cnt--;
    return b;
  }
  uint32_t bits(int k) { uint32_t v = 0; for (int i = 0; i < k; i++) v |= (uint32_t)bit() << i; return v; }
  void alignByte() { buf = 0; cnt = 0; }
};

// canonical Huffman code: count[len] codes of each length, symbols ordered by (length, value)
struct Huff { uint16_t count[16]; uint16_t symbol[288]; };

bool buildHuff(Huff& h, const uint8_t* lengths, int n) {
  memset(h.count, 0, sizeof(h.count));
  for (int i = 0; i < n; i++) h.count[lengths[i]]++;
  int left = 1;
  for (int len = 1; len < 16; len++) { left <<= 1; left -= h.count[len]; if (left < 0) return false; }   // over-subscribed
  uint16_t offs[16]; offs[1] = 0;
  for (int len = 1; len < 15; len++) offs[len + 1] = (uint16_t)(offs[len] + h.count[len]);
  for (int s = 0; s < n; s++) if (lengths[s]) h.symbol[offs[lengths[s]]++] = (uint16_t)s;
  return true;
}

int decodeSym(BitReader& br, const Huff& h) {
  int code = 0, first = 0, index = 0;
  for (int len = 1; len < 16; len++) {
    code |= br.bit();
    const int count = h.count[len];
    if (code - count < first) return h.symbol[index + (code - first)];
    index += count; first += count; first <<= 1; code <<= 1;
  }
  return -1;
}

const uint16_t kLenBase[29] = { 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258 };
const uint8_t kLenExtra[29] = { 0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0 };
const uint16_t kDistBase[30] = { 1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073,
                                 4097, 6145, 8193, 12289, 16385, 24577 };
const uint8_t kDistExtra[30] = { 0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13 };

bool inflateCodes(BitReader& br, const Huff& lit, const Huff& dist, std::vector<uint8_t>& out) {
  for (;;) {
    const int sym = decodeSym(br, lit);
    if (sym < 0 || br.eof) return false;
    if (sym < 256) { out.push_back((uint8_t)sym); continue; }
    if (sym == 256) return true;
    if (sym > 285) return false;
    const int len = kLenBase[sym - 257] + (int)br.bits(kLenExtra[sym - 257]);
    const int ds = decodeSym(br, dist);
    if (ds < 0 || ds > 29) return false;
    const size_t d = kDistBase[ds] + (size_t)br.bits(kDistExtra[ds]);
    if (d > out.size()) return false;
    for (int i = 0; i < len; i++) out.push_back(out[out.size() - d]);
  }
}

bool inflateRaw(const uint8_t* data, size_t n, std::vector<uint8_t>& out) {
  BitReader br{ data, n };
  for (;;) {
    const int last = br.bit();
    const uint32_t type = br.bits(2);
    if (br.eof) return false;
    if (type == 0) {                                                  // stored
      br.alignByte();
      if (br.pos + 4 > n) return false;
      const uint32_t len = data[br.pos] | (data[br.pos + 1] << 8), nlen = data[br.pos + 2] | (data[br.pos + 3] << 8);
      br.pos += 4;
      if ((len ^ 0xffffu) != nlen || br.pos + len > n) return false;
      out.insert(out.end(), data + br.pos, data + br.pos + len);
      br.pos += len;
    } else if (type == 1) {                                           // fixed codes
      uint8_t l[288];
      for (int i = 0; i < 144; i++) l[i] = 8;
      for (int i = 144; i < 256; i++) l[i] = 9;
      for (int i = 256; i < 280; i++) l[i] = 7;
      for (int i = 280; i < 288; i++) l[i] = 8;
      uint8_t dl[30]; for (int i = 0; i < 30; i++) dl[i] = 5;
      Huff lit, dist;
      if (!buildHuff(lit, l, 288) || !buildHuff(dist, dl, 30)) return false;
      if (!inflateCodes(br, lit, dist, out)) return false;
    } else if (type == 2) {                                           // dynamic codes
      const int nlen = (int)br.bits(5) + 257, ndist = (int)br.bits(5) + 1, ncode = (int)br.bits(4) + 4;
      if (nlen > 286 || ndist > 30) return false;
      static const uint8_t order[19] = { 16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15 };
      uint8_t cl[19] = { 0 };
      for (int i = 0; i < ncode; i++) cl[order[i]] = (uint8_t)br.bits(3);
      Huff lencode;
      if (!buildHuff(lencode, cl, 19)) return false;
      uint8_t lengths[286 + 30];
      int idx = 0;
      while (idx < nlen + ndist) {
        const int sym = decodeSym(br, lencode);
        if (sym < 0 || br.eof) return false;
        if (sym < 16) { lengths[idx++] = (uint8_t)sym; continue; }
        int rep; uint8_t val = 0;
        if (sym == 16) { if (idx == 0) return false; val = lengths[idx - 1]; rep = 3 + (int)br.bits(2); }
        else if (sym == 17) rep = 3 + (int)br.bits(3);
        else rep = 11 + (int)br.bits(7);
        if (idx + rep > nlen + ndist) return false;
        while (rep--) lengths[idx++] = val;
      }
      if (lengths[256] == 0) return false;
      Huff lit, dist;
      if (!buildHuff(lit, lengths, nlen) || !buildHuff(dist, lengths + nlen, ndist)) return false;
      if (!inflateCodes(br, lit, dist, out)) return false;
    } else {
      return false;
    }
    if (last) return true;
  }
}

// ---------------------------------------------------------------- PNG ----------------------
inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

bool decodePNG(const std::vector<uint8_t>& file, int& width, int& height, std::vector<uint8_t>& rgb, std::string& err) {
  size_t pos = 8;
  uint32_t w = 0, h = 0; int depth = 0, ctype = -1, interlace = 0;
  std::vector<uint8_t> idat, plte;
  bool end = false;
  while (!end && pos + 12 <= file.size()) {
    const uint32_t len = be32(&file[pos]);
    const uint8_t* type = &file[pos + 4];
    if (pos + 12 + (size_t)len > file.size()) { err = "truncated PNG chunk"; return false; }
    const uint8_t* d = &file[pos + 8];
    if (!memcmp(type, "IHDR", 4) && len >= 13) { w = be32(d); h = be32(d + 4); depth = d[8]; ctype = d[9]; interlace = d[12]; }
    else if (!memcmp(type, "PLTE", 4)) plte.assign(d, d + len);
    else if (!memcmp(type, "IDAT", 4)) idat.insert(idat.end(), d, d + len);
    else if (!memcmp(type, "IEND", 4)) end = true;
    pos += 12 + (size_t)len;
  }
  if (w == 0 || h == 0 || w > 32768 || h > 32768) { err = "bad PNG header"; return false; }
  if (interlace > 1) { err = "bad PNG interlace method"; return false; }
  int channels;
  switch (ctype) { case 0: channels = 1; break; case 2: channels = 3; break; case 3: channels = 1; break;
                   case 4: channels = 2; break; case 6: channels = 4; break; default: err = "bad PNG colour type"; return false; }
  if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4))) || (ctype == 3 && depth == 16)) {
    err = "bad PNG bit depth"; return false;
  }
  if (idat.size() < 2 || (idat[0] & 0x0f) != 8 || (idat[1] & 0x20)) { err = "bad zlib stream in PNG"; return false; }
  // The image as one pass (interlace method 0) or as the seven passes of Adam7 (method 1; PNG specification, section 8.2): pass p
  // holds the pixels (xs + i dx, ys + j dy) as an image of its own -- own scanlines, own filter bytes, filtering restarts with
  // a zero row -- and a pass without pixels has no bytes at all.
  struct Pass { uint32_t xs, ys, dx, dy; };
  static const Pass adam7[7] = { { 0, 0, 8, 8 }, { 4, 0, 8, 8 }, { 0, 4, 4, 8 }, { 2, 0, 4, 4 }, { 0, 2, 2, 4 }, { 1, 0, 2, 2 }, { 0, 1, 1, 2 } };
  static const Pass whole = { 0, 0, 1, 1 };
  const int nPass = interlace ? 7 : 1;
  size_t need = 0;
  for (int pi = 0; pi < nPass; pi++) {
    const Pass& ps = interlace ? adam7[pi] : whole;
    const uint32_t pw = w > ps.xs ? (w - ps.xs + ps.dx - 1) / ps.dx : 0, ph = h > ps.ys ? (h - ps.ys + ps.dy - 1) / ps.dy : 0;
    if (pw && ph) need += (((size_t)pw * channels * depth + 7) / 8 + 1) * ph;
  }
  std::vector<uint8_t> raw;
  raw.reserve(need);
  if (!inflateRaw(idat.data() + 2, idat.size() - 2, raw) || raw.size() < need) { err = "cannot inflate PNG data"; return false; }
  const int bpp = (channels * depth + 7) / 8;                            // filter unit
  rgb.assign((size_t)w * h * 3, 0);
  size_t at = 0;
  for (int pi = 0; pi < nPass; pi++) {
    const Pass& ps = interlace ? adam7[pi] : whole;
    const uint32_t pw = w > ps.xs ? (w - ps.xs + ps.dx - 1) / ps.dx : 0, ph = h > ps.ys ? (h - ps.ys + ps.dy - 1) / ps.dy : 0;
    if (!pw || !ph) continue;
    const size_t stride = ((size_t)pw * channels * depth + 7) / 8;
    std::vector<uint8_t> prev(stride, 0), cur(stride);
    for (uint32_t y = 0; y < ph; y++) {
      const uint8_t* row = &raw[at]; at += stride + 1;
      const int ft = row[0];
      for (size_t i = 0; i < stride; i++) {
        const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
        int v = row[1 + i];
        switch (ft) { case 0: break; case 1: v += a; break; case 2: v += b; break; case 3: v += (a + b) >> 1; break;
                      case 4: v += paeth(a, b, c); break; default: err = "bad PNG filter"; return false; }
        cur[i] = (uint8_t)v;
      }
      for (uint32_t x = 0; x < pw; x++) {
        uint8_t* o = &rgb[((size_t)(ps.ys + y * ps.dy) * w + (ps.xs + x * ps.dx)) * 3];
        // sample k of pixel x as an 8-bit value (16-bit samples keep their high byte, as png_set_strip_16 does)
        auto sample = [&](int k) -> int {
          if (depth == 8) return cur[(size_t)x * channels + k];
          if (depth == 16) return cur[((size_t)x * channels + k) * 2];
          const size_t bitpos = (size_t)x * depth;
          const int v = (cur[bitpos >> 3] >> (8 - depth - (int)(bitpos & 7))) & ((1 << depth) - 1);
          return ctype == 3 ? v : v * 255 / ((1 << depth) - 1);
        };
        if (ctype == 3) {
          const size_t idx = (size_t)sample(0);
          if (3 * idx + 2 < plte.size()) { o[0] = plte[3 * idx]; o[1] = plte[3 * idx + 1]; o[2] = plte[3 * idx + 2]; }
        } else if (channels <= 2) {
          o[0] = o[1] = o[2] = (uint8_t)sample(0);
        } else {
          o[0] = (uint8_t)sample(0); o[1] = (uint8_t)sample(1); o[2] = (uint8_t)sample(2);
        }
      }
      prev.swap(cur);
    }
  }
  width = (int)w; height = (int)h;
  return true;
}

// ---------------------------------------------------------------- PNM ----------------------
bool decodePNM(const std::vector<uint8_t>& file, int& width, int& height, std::vector<uint8_t>& rgb, std::string& err) {
  const int channels = file[1] == '6' ? 3 : 1;
  size_t pos = 2; long vals[3]; int got = 0;
  while (got < 3 && pos < file.size()) {
    if (file[pos] == '#') { while (pos < file.size() && file[pos] != '\n') pos++; continue; }
    if (file[pos] <= ' ') { pos++; continue; }
    long v = 0; bool any = false;
    while (pos < file.size() && file[pos] >= '0' && file[pos] <= '9') { v = v * 10 + (file[pos] - '0'); pos++; any = true; }
    if (!any) { err = "bad PNM header"; return false; }
    vals[got++] = v;
  }
  pos++;   // the single whitespace byte after maxval
  if (got < 3 || vals[0] <= 0 || vals[1] <= 0 || vals[2] <= 0 || vals[2] > 65535 || vals[0] > 32768 || vals[1] > 32768) { err = "bad PNM header"; return false; }
  const size_t bytes = vals[2] > 255 ? 2 : 1, need = (size_t)vals[0] * vals[1] * channels * bytes;
  if (pos + need > file.size()) { err = "truncated PNM data"; return false; }
  width = (int)vals[0]; height = (int)vals[1];
  rgb.resize((size_t)width * height * 3);
  for (size_t i = 0; i < (size_t)width * height; i++)
    for (int k = 0; k < 3; k++) {
      const size_t s = (i * channels + (channels == 3 ? k : 0)) * bytes;
      const long v = bytes == 2 ? ((long)file[pos + s] << 8 | file[pos + s + 1]) : file[pos + s];
      rgb[3 * i + k] = (uint8_t)(v * 255 / vals[2]);
    }
  return true;
}

}  // namespace

bool decodeJPEG(const std::vector<uint8_t>& file, int& width, int& height, std::vector<uint8_t>& rgb, std::string& err);   // jpeg_read.cpp

bool readImage(const std::string& path, int& width, int& height, std::vector<uint8_t>& rgbTopDown, std::string& err) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) { err = "cannot open image " + path; return false; }
  std::vector<uint8_t> file;
  uint8_t buf[65536]; size_t n;
  while ((n = fread(buf, 1, sizeof(buf), f)) > 0) file.insert(file.end(), buf, buf + n);
  fclose(f);
  static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
  bool ok;
  if (file.size() > 8 && !memcmp(file.data(), sig, 8)) ok = decodePNG(file, width, height, rgbTopDown, err);
  else if (file.size() > 2 && file[0] == 'P' && (file[1] == '5' || file[1] == '6')) ok = decodePNM(file, width, height, rgbTopDown, err);
  else if (file.size() > 3 && file[0] == 0xff && file[1] == 0xd8 && file[2] == 0xff) ok = decodeJPEG(file, width, height, rgbTopDown, err);
  else { err = "unsupported image format (PNG, JPEG and binary PNM are read)"; ok = false; }
  if (!ok) err = path + ": " + err;
  return ok;
}

// The RT_FORMAT_FLOAT4 buffer of MinimalOptiX.cpp:459-472: row j of the buffer is image row H-1-j,
// channels are QColor::redF() etc. (the 8-bit value widened to 16 bits, over 65535), alpha 1.
void imageToTextureRGBA(const std::vector<uint8_t>& rgbTopDown, int width, int height, std::vector<float>& rgba) {
  rgba.resize((size_t)width * height * 4);
  for (int j = 0; j < height; j++)
    for (int i = 0; i < width; i++) {
      const uint8_t* src = &rgbTopDown[((size_t)(height - j - 1) * width + i) * 3];
      float* dst = &rgba[4 * ((size_t)j * width + i)];
      for (int k = 0; k < 3; k++) dst[k] = (float)((double)(src[k] * 257) / 65535.0);
      dst[3] = 1.f;
    }
}

}  // namespace moptix
