// rf_pt_format.hpp -- the ".pt" scene container ("PTFORMAT3").
// Byte layout and error behaviour: src/pt-format/pt_format.cpp:238-321, pt_format.hpp:18-43;
// error strings pinned by src/tests/pt_format.cpp:180-213.
//
//   "PTFORMAT3"                                  9 bytes, no terminator
//   array<BvhNode 48>  array<Positions 36>  array<PositionAttribute 48>  array<VertexAttributes 80>
//   array<vec4 16> positions  array<vec4 16> normals  array<vec2 8> texcoords  array<u32> indices
//   slices(positions) slices(normals) slices(texcoords) slices(indices)      (per raster mesh)
//   array<u32> modelBaseColorTextureIndices
//   u64 numTextures, each: {u32 width, u32 height} array<u32 BGRA>
// where array<T> = u64 count + count*sizeof(T) raw little-endian bytes and
// slices = u64 n + n x {u64 offset, u64 count}.
#pragma once

#include "rf_types.hpp"

#include <cstdint>
#include <string>
#include <vector>

namespace rf
{
struct Vec4
{
    float x, y, z, w;
};

struct Texture
{
    std::vector<uint32_t> pixels; // b | g<<8 | r<<16 | a<<24
    uint32_t              width = 0;
    uint32_t              height = 0;

    bool operator==(const Texture&) const = default;
};

struct Slice
{
    uint64_t offset = 0;
    uint64_t count = 0;

    bool operator==(const Slice&) const = default;
};

struct PtFormat
{
    std::vector<BvhNode>           bvhNodes;
    std::vector<Positions>         bvhPositionAttributes;
    std::vector<PositionAttribute> trianglePositionAttributes;
    std::vector<VertexAttributes>  triangleVertexAttributes;

    std::vector<Vec4>     vertexPositions;
    std::vector<Vec4>     vertexNormals;
    std::vector<Vec2>     vertexTexCoords;
    std::vector<uint32_t> vertexIndices;
    std::vector<Slice>    modelVertexPositions;
    std::vector<Slice>    modelVertexNormals;
    std::vector<Slice>    modelVertexTexCoords;
    std::vector<Slice>    modelVertexIndices;
    std::vector<uint32_t> modelBaseColorTextureIndices;

    std::vector<Texture> baseColorTextures;
};

// Both throw std::runtime_error.  deserialize reports a wrong version / foreign file with the
// reference's exact messages; a truncated stream raises "Unexpected end of PtFormat data."
// (the reference only asserts there).
std::vector<uint8_t> serializePt(const PtFormat& format);
void                 deserializePt(const uint8_t* data, std::size_t size, PtFormat& format);

void     writePtFile(const std::string& path, const PtFormat& format);
PtFormat readPtFile(const std::string& path);
} // namespace rf
