// image_io.h -- writes the 8-bit canvas (QImage::save in the reference, MinimalOptiX.cpp:68-84) and reads
// texture images (QImage(path), MinimalOptiX.cpp:446).
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

// QImage(path) of the texture upload (MinimalOptiX.cpp:446): 8-bit RGB, row 0 = top of the image.
// Reads PNG (all colour types and bit depths, non-interlaced), baseline JPEG (jpeg_read.cpp) and binary PNM.
bool readImage(const std::string& path, int& width, int& height, std::vector<uint8_t>& rgbTopDown, std::string& err);
// the float4 texture buffer the reference fills from it (MinimalOptiX.cpp:459-472): flipped, alpha 1
void imageToTextureRGBA(const std::vector<uint8_t>& rgbTopDown, int width, int height, std::vector<float>& rgba);
}
