// rf_gltf.hpp -- glTF 2.0 / GLB ingest and the glTF -> PtFormat bake.
//
// Behavioural contract: src/common/gltf_model.cpp:29-72 (node transforms), :170-243 (base colour
// textures), :266-465 (mesh extraction, sort by texture), src/common/flattened_model.cpp:22-43,
// src/common/texture.cpp:12-65, src/pt-format/pt_format.cpp:20-151.
// The reference parses with cgltf 1.13 and decodes images with stb_image (both un-vendored third
// party code); this file carries its own GLB/JSON reader and PNG decoder (zlib inflate).
#pragma once

#include "rf_pt_format.hpp"

#include <span>
#include <string>
#include <vector>

namespace rf
{
struct GltfMesh
{
    std::vector<Vec3>     positions;
    std::vector<Vec3>     normals;
    std::vector<Vec2>     texCoords;
    std::vector<uint32_t> indices;
    std::size_t           baseColorTextureIndex = 0;
};

struct GltfModel
{
    std::vector<GltfMesh> meshes; // sorted by baseColorTextureIndex
    std::vector<Texture>  baseColorTextures;
};

GltfModel loadGltfModel(const std::string& path);

// Decode a PNG byte stream to BGRA pixels the way the reference's Texture::fromMemory does
// (4 channels forced, alpha forced to 255).
Texture textureFromMemory(std::span<const uint8_t> data);
// Texture::fromPixel (texture.cpp:56-65)
Texture textureFromPixel(float r, float g, float b, float a);

// BVH builder used by the two bakes below: -1 = host builder (default), >= 0 = GPU builder on that device
// (same node bytes; see rf_bvh_gpu.hpp for the order inside multi-triangle leaves).
void     setBakeBvhBuilder(int gpuDeviceOrMinusOne);
PtFormat ptFormatFromGltf(const std::string& path);
PtFormat ptFormatFromTriangles(std::span<const Positions> positions, std::span<const Normals> normals, std::span<const TexCoords> texCoords,
                               std::span<const uint32_t> textureIndices, std::vector<Texture> textures);
} // namespace rf
