// image_io.h -- writes the 8-bit canvas (QImage::save in the reference, MinimalOptiX.cpp:68-84).
// No libpng/zlib headers exist in the image, so the PNG encoder emits stored (uncompressed)
// deflate blocks; PPM is available as well.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace moptix {
bool writePNG(const std::string& path, const uint8_t* rgb, uint32_t width, uint32_t height);
bool writePPM(const std::string& path, const uint8_t* rgb, uint32_t width, uint32_t height);
bool writePFM(const std::string& path, const float* rgbBottomUp, uint32_t width, uint32_t height);
}
