// cli_common.hpp -- helpers shared by the command-line tools (PNG/PFM writers, error exit).
#pragma once

#include <rayfinder_amd.h>
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

inline void rfCheck(int status, const char* what)
{
    if (status != RF_OK)
    {
        std::fprintf(stderr, "Exception occurred. %s (%s)\n", rf_last_error_message(), what);
        std::exit(1);
    }
}

// 8-bit RGBA PNG (filter 0 rows, one zlib stream).
inline bool writePngRgba(const std::string& path, const uint8_t* rgba, uint32_t w, uint32_t h)
{
    std::vector<uint8_t> raw(static_cast<size_t>(h) * (1 + 4 * static_cast<size_t>(w)));
    for (uint32_t y = 0; y < h; ++y)
    {
        raw[y * (1 + 4 * static_cast<size_t>(w))] = 0;
        std::memcpy(&raw[y * (1 + 4 * static_cast<size_t>(w)) + 1], rgba + static_cast<size_t>(y) * w * 4, static_cast<size_t>(w) * 4);
    }
    uLongf               zlen = compressBound(static_cast<uLong>(raw.size()));
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), static_cast<uLong>(raw.size()), 6) != Z_OK) return false;
    FILE* fp = std::fopen(path.c_str(), "wb");
    if (!fp) return false;
    const auto be = [](uint32_t v, uint8_t* p) {
        p[0] = static_cast<uint8_t>(v >> 24);
        p[1] = static_cast<uint8_t>(v >> 16);
        p[2] = static_cast<uint8_t>(v >> 8);
        p[3] = static_cast<uint8_t>(v);
    };
    const auto chunk = [&](const char* type, const uint8_t* data, uint32_t len) {
        uint8_t hdr[8];
        be(len, hdr);
        std::memcpy(hdr + 4, type, 4);
        std::fwrite(hdr, 1, 8, fp);
        if (len) std::fwrite(data, 1, len, fp);
        uLong crc = crc32(0L, reinterpret_cast<const Bytef*>(type), 4);
        if (len) crc = crc32(crc, data, len);
        uint8_t c[4];
        be(static_cast<uint32_t>(crc), c);
        std::fwrite(c, 1, 4, fp);
    };
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::fwrite(sig, 1, 8, fp);
    uint8_t ihdr[13];
    be(w, ihdr);
    be(h, ihdr + 4);
    ihdr[8] = 8;
    ihdr[9] = 6;
    ihdr[10] = ihdr[11] = ihdr[12] = 0;
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", z.data(), static_cast<uint32_t>(zlen));
    chunk("IEND", nullptr, 0);
    std::fclose(fp);
    return true;
}

// Little-endian RGB PFM (bottom row first), mean radiance.
inline bool writePfm(const std::string& path, const float* rgba, uint32_t w, uint32_t h, float scale)
{
    FILE* fp = std::fopen(path.c_str(), "wb");
    if (!fp) return false;
    std::fprintf(fp, "PF\n%u %u\n-1.0\n", w, h);
    std::vector<float> row(3 * static_cast<size_t>(w));
    for (uint32_t y = 0; y < h; ++y)
    {
        const float* src = rgba + 4 * static_cast<size_t>(h - 1 - y) * w;
        for (uint32_t x = 0; x < w; ++x)
            for (int c = 0; c < 3; ++c) row[3 * x + c] = src[4 * x + c] * scale;
        std::fwrite(row.data(), sizeof(float), row.size(), fp);
    }
    std::fclose(fp);
    return true;
}

inline bool endsWith(const std::string& s, const char* suffix)
{
    const size_t n = std::strlen(suffix);
    return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
}

inline rf_pt_format* loadScene(const std::string& path)
{
    rf_pt_format* pt = nullptr;
    if (endsWith(path, ".pt")) rfCheck(rf_pt_format_load(path.c_str(), &pt), "load .pt");
    else rfCheck(rf_pt_format_from_gltf(path.c_str(), &pt), "bake glTF");
    return pt;
}
