// rf_query.hpp -- the CPU-side BVH query of the product: one ray (or a batch, or the bvh-visualizer's pixel grid) against a
// flattened 48-byte-node tree on the HOST, no GPU involved.
//
// Replaces  bool rayIntersectBvh(const Ray&, span<const BvhNode>, span<const Positions>, float tMax, Intersection&, BvhStats*)
// (src/common/ray_intersection.hpp:43-49, .cpp:138-213) -- used by the reference for focus picking (src/pt/main.cpp:214-225)
// and by its bvh-visualizer tool (src/bvh-visualizer/main.cpp:60-78) -- with the same results bit for bit: hit flag, t, the
// offset hit point p (offsetRay, .cpp:17-35) and BvhStats::nodesVisited.  A pure function of its arguments, re-entrant; the
// batch forms run it on std::threads over static blocks of rays / scanlines.
//
// Own construction (not the reference's loop): the walk keeps "the node to look at next" in a register and a pending list of
// far children that grows on demand (the reference's 32-entry array overruns silently past depth 32, .cpp:148,194), triangle
// vertices are read through a stride so that the 36-byte CPU records and the 48-byte GPU records of a .pt file traverse
// alike, and every index read from the node array is bounds-checked (a .pt file is untrusted input).
#pragma once

#include "rf_types.hpp"

#include <cstdint>
#include <span>

namespace rf
{
struct HostIntersection
{
    Vec3     p;        // offset hit point (the reference's Intersection::p)
    float    t;        // the reference's Intersection::t
    uint32_t triangle; // index into the triangle array (leaf order); 0xFFFFFFFF on a miss
    float    u, v;     // barycentrics of the hit (weights 1-u-v, u, v)
};

struct HostBvhStats
{
    uint32_t nodesVisited;  // the reference's BvhStats::nodesVisited
    uint32_t triangleTests; // Moeller-Trumbore evaluations
    uint32_t stackHigh;     // high-water mark of the pending list
};

// Triangle vertices with a byte stride: 36 (Positions) or 48 (PositionAttribute: p0 pad p1 pad p2 pad).
struct TriangleSpan
{
    const uint8_t* data = nullptr;
    uint64_t       count = 0;
    uint32_t       strideBytes = 36;
};

// Closest hit with t in (1e-5, tMax).  Returns false on a miss (`out` untouched except triangle = 0xFFFFFFFF).
// Throws std::runtime_error if the tree links outside the node / triangle arrays.
bool intersectBvh(Vec3 origin, Vec3 direction, std::span<const BvhNode> nodes, TriangleSpan triangles, float tMax, HostIntersection& out, HostBvhStats* stats);

// n rays (6 floats each: origin, direction) on `threads` host threads (0 = std::thread::hardware_concurrency()); any output
// pointer may be null.  hit[i] = 1 / 0.
void intersectBvhBatch(const float* rays6, uint64_t n, std::span<const BvhNode> nodes, TriangleSpan triangles, float tMax, uint32_t threads, uint8_t* hit,
                       HostIntersection* out, HostBvhStats* stats);

// The bvh-visualizer pixel loop (src/bvh-visualizer/main.cpp:60-78): pinhole ray through u = j/W, v = 1-(i+1)/H
// (generateCameraRay, src/common/camera.cpp:44-52), tMax = FLT_MAX; rows [rowBegin, rowEnd) of a width x height grid, outputs
// indexed by i * width + j over the WHOLE grid.  Any output pointer may be null.
void bvhVisualizerPass(const Camera& camera, uint32_t width, uint32_t height, uint32_t rowBegin, uint32_t rowEnd, std::span<const BvhNode> nodes, TriangleSpan triangles,
                       uint32_t threads, uint32_t* nodesVisited, uint8_t* hit, float* t, uint32_t* triangleTests);
} // namespace rf
