#include "image_io.h"

#include <cstdio>
#include <cstring>

namespace moptix {
namespace {

uint32_t crcTable[256];
bool crcInit = false;
void initCrc() {
  for (uint32_t n = 0; n < 256; n++) {
    uint32_t c = n;
    for (int k = 0; k < 8; k++) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
    crcTable[n] = c;
  }
  crcInit = true;
}
uint32_t crc32(uint32_t crc, const uint8_t* buf, size_t len) {
  if (!crcInit) initCrc();
  crc = ~crc;
  for (size_t i = 0; i < len; i++) crc = crcTable[(crc ^ buf[i]) & 0xff] ^ (crc >> 8);
  return ~crc;
}
void put32(std::vector<uint8_t>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }
void chunk(FILE* fp, const char type[4], const std::vector<uint8_t>& data) {
  std::vector<uint8_t> hdr; put32(hdr, (uint32_t)data.size());
  fwrite(hdr.data(), 1, 4, fp);
  std::vector<uint8_t> body(type, type + 4);
  body.insert(body.end(), data.begin(), data.end());
  fwrite(body.data(), 1, body.size(), fp);
  std::vector<uint8_t> c; put32(c, crc32(0, body.data(), body.size()));
  fwrite(c.data(), 1, 4, fp);
}

}  // namespace

bool writePNG(const std::string& path, const uint8_t* rgb, uint32_t width, uint32_t height) {
  FILE* fp = fopen(path.c_str(), "wb");
  if (!fp) return false;
  static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
  fwrite(sig, 1, 8, fp);
  std::vector<uint8_t> ihdr; put32(ihdr, width); put32(ihdr, height);
  ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
  chunk(fp, "IHDR", ihdr);
  // raw scanlines: filter byte 0 + RGB
  const size_t stride = 1 + 3 * (size_t)width;
  std::vector<uint8_t> raw(stride * height);
  for (uint32_t y = 0; y < height; y++) {
    raw[y * stride] = 0;
    memcpy(&raw[y * stride + 1], rgb + 3 * (size_t)width * y, 3 * (size_t)width);
  }
  // zlib stream of stored blocks
  std::vector<uint8_t> z; z.push_back(0x78); z.push_back(0x01);
  uint32_t a = 1, b = 0;
  size_t pos = 0;
  while (pos < raw.size()) {
    const size_t n = std::min<size_t>(65535, raw.size() - pos);
    z.push_back(pos + n == raw.size() ? 1 : 0);
    z.push_back(n & 0xff); z.push_back(n >> 8); z.push_back(~n & 0xff); z.push_back((~n >> 8) & 0xff);
    z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
    for (size_t i = 0; i < n; i++) { a = (a + raw[pos + i]) % 65521u; b = (b + a) % 65521u; }
    pos += n;
  }
  put32(z, (b << 16) | a);
  chunk(fp, "IDAT", z);
  chunk(fp, "IEND", {});
  fclose(fp);
  return true;
}

bool writePPM(const std::string& path, const uint8_t* rgb, uint32_t width, uint32_t height) {
  FILE* fp = fopen(path.c_str(), "wb");
  if (!fp) return false;
  fprintf(fp, "P6\n%u %u\n255\n", width, height);
  fwrite(rgb, 1, 3 * (size_t)width * height, fp);
  fclose(fp);
  return true;
}

bool writePFM(const std::string& path, const float* rgbBottomUp, uint32_t width, uint32_t height) {
  FILE* fp = fopen(path.c_str(), "wb");
  if (!fp) return false;
  fprintf(fp, "PF\n%u %u\n-1.0\n", width, height);   // PFM rows are bottom-up, like accuBuffer
  fwrite(rgbBottomUp, sizeof(float), 3 * (size_t)width * height, fp);
  fclose(fp);
  return true;
}

}  // namespace moptix
