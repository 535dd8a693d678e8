// rf_jpeg.hpp -- JPEG decoder for base-colour textures (see rf_jpeg.cpp).
#pragma once

#include <cstdint>
#include <span>
#include <vector>

namespace rf
{
struct Rgba8Image
{
    std::vector<uint8_t> rgba;
    uint32_t             width = 0, height = 0;
};

bool       looksLikeJpeg(std::span<const uint8_t> data);
// Baseline / extended sequential / progressive Huffman JPEG, 8 bit, 1 or 3 components -> RGBA8
// (alpha 255), the way stb_image returns it for req_comp = 4.  Throws std::runtime_error.
Rgba8Image decodeJpeg(std::span<const uint8_t> data);
} // namespace rf
