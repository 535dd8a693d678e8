// rf_types.hpp -- plain-old-data records of the .pt scene format and the flattened BVH.
// Byte layouts are the reference's (they are the drop-in surface):
//   BvhNode            48 B  src/common/bvh.hpp:14-21 (+ Aabb 32 B, src/common/aabb.hpp:12-27)
//   Positions          36 B  src/common/triangle_attributes.hpp:7-12
//   PositionAttribute  48 B  src/pt-format/vertex_attributes.hpp:7-15
//   VertexAttributes   80 B  src/pt-format/vertex_attributes.hpp:17-35
//   Camera             76 B  src/common/camera.hpp:10-21
#pragma once

#include "rf_math.hpp"

#include <cstddef>
#include <cstdint>

namespace rf
{
struct Aabb
{
    Vec3  min;
    float pad0;
    Vec3  max;
    float pad1;
};
static_assert(sizeof(Aabb) == 32);

struct BvhNode
{
    Aabb     aabb;
    uint32_t trianglesOffset;
    uint32_t secondChildOffset;
    uint32_t triangleCount;
    uint32_t splitAxis;
};
static_assert(sizeof(BvhNode) == 48);

struct Positions
{
    Vec3 v0, v1, v2;
};
static_assert(sizeof(Positions) == 36);

struct Normals
{
    Vec3 n0, n1, n2;
};

struct Vec2
{
    float x, y;
};

struct TexCoords
{
    Vec2 uv0, uv1, uv2;
};

struct PositionAttribute
{
    Vec3  p0;
    float pad0;
    Vec3  p1;
    float pad1;
    Vec3  p2;
    float pad2;
};
static_assert(sizeof(PositionAttribute) == 48);

struct VertexAttributes
{
    Vec3     n0;
    float    pad0;
    Vec3     n1;
    float    pad1;
    Vec3     n2;
    float    pad2;
    Vec2     uv0, uv1, uv2;
    uint32_t textureIdx;
    uint32_t pad3;
};
static_assert(sizeof(VertexAttributes) == 80);

struct Camera
{
    Vec3  origin;
    Vec3  lowerLeftCorner;
    Vec3  horizontal;
    Vec3  vertical;
    Vec3  up;
    Vec3  right;
    float lensRadius;
};
static_assert(sizeof(Camera) == 76);

// Texture descriptor as uploaded by the reference (src/pt/reference_path_tracer.cpp:211-214).
struct TextureDescriptor
{
    uint32_t width, height, offset;
};
static_assert(sizeof(TextureDescriptor) == 12);

// AlignedSkyState (src/pt/aligned_sky_state.hpp:34-41): 40 floats / 160 B.
struct SkyStateGpu
{
    float params[27];
    float skyRadiances[3];
    float solarRadiances[3];
    float padding1[3];
    float sunDirection[3];
    float padding2;
};
static_assert(sizeof(SkyStateGpu) == 160);
} // namespace rf
