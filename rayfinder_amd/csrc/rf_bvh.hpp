// rf_bvh.hpp -- binned-SAH BVH build emitting the reference's depth-first 48-byte node array.
// Behavioural contract: src/common/bvh.hpp:23-46 and src/common/bvh.cpp:81-291.
#pragma once

#include "rf_types.hpp"

#include <cstddef>
#include <span>
#include <vector>

namespace rf
{
struct Bvh
{
    std::vector<BvhNode> nodes;
    // triangleIndices[source index] = index in leaf order (bvh.hpp:26-30).
    std::vector<std::size_t> triangleIndices;
    int                      depth = 0; // root = 1
};

Bvh buildBvh(std::span<const Positions> triangles);

// Structural check of a flattened BVH + attribute arrays that come from outside (a .pt file, the C ABI).
// The reference leans on WGSL robust buffer access for malformed scenes; HIP has none, so a scene is
// validated once before it is uploaded.  Throws std::runtime_error naming the first violation:
//   interior node i:  i + 1 < secondChildOffset < numNodes (children strictly forward: no cycles), splitAxis <= 2
//   leaf node:        trianglesOffset + triangleCount <= numTriangles
//   textureIdx < max(numTextures, 1) for every triangle (the renderer supplies one white texel when there is none)
void validateScene(std::span<const BvhNode> nodes, std::size_t numTriangles, std::span<const VertexAttributes> vertexAttributes,
                   std::size_t numTextures);

template<typename T>
std::vector<T> reorderAttributes(std::span<const T> attributes, std::span<const std::size_t> triangleIndices)
{
    std::vector<T> out(attributes.size());
    for (std::size_t i = 0; i < attributes.size(); ++i) out[triangleIndices[i]] = attributes[i];
    return out;
}
} // namespace rf
