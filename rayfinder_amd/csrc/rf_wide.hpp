// rf_wide.hpp -- the render path's BVH layout: 64-byte "children in the parent" nodes.
//
// The reference visits one 48-byte node per step and decides with
//     hit(node) = P(node, ray) && tmin(node, ray) < rayTMax            (wgsl:447-475)
// where P (the two slab early-outs and `tmax > 0`) and tmin depend on the box and the ray only;
// rayTMax enters through the last comparison alone.  That makes the following re-arrangement
// decision-for-decision identical to the reference (same leaves entered in the same order, same
// triangles tested against the same rayTMax, hence bit-identical hits):
//
//   * one record per INTERIOR node holding BOTH children's boxes (16 dwords = 64 B, 64-B aligned:
//     one cache line, four dwordx4 loads, one dependent fetch per two box tests);
//   * the near child (reference order: dirNeg[splitAxis], wgsl:409-417) is handled at once; the
//     far child is pushed together with its tmin only if P holds, and when popped it is accepted
//     iff tmin < rayTMax *then* -- exactly the test the reference performs at pop time;
//   * a leaf is described by its parent's child word, so leaf nodes are never fetched.
//
// Child word: bit 31 = leaf.  Interior: index of the child's wide record.  Leaf: bits 30..28 =
// min(count-1, 7), bits 27..0 = first triangle (count <= 7) or index into the big-leaf table
// {first triangle, count} (count >= 8, or offsets >= 2^28).
//
// nodesVisited bookkeeping (the counting build): the reference counts a node when it is visited,
// i.e. root once, the near child at its parent's step, the far child when popped -- also when its
// box test then fails.  The counting build therefore pushes every far child (tmin = +inf when P
// fails) so that visit counts AND the stack high-water mark equal the reference's exactly.
#pragma once

#include "rf_device.hpp"

#include <vector>

namespace rf
{
constexpr uint32_t kWideLeafBit = 0x80000000u;
constexpr uint32_t kWideNone = 0xFFFFFFFFu; // scene.rootLeaf when the root is interior
constexpr int      kWideLdsStack = 12;      // (child word, tmin) pairs kept in LDS per lane
constexpr int      kWideSpillStack = 84;    // further pairs in scratch (total depth 96, as rf_device.hpp)

struct WideScene
{
    const float4* nodes;     // 4 float4 per interior node
    const uint2*  bigLeaves; // {first triangle, count}
    float4        rootLo;    // root box (w unused)
    float4        rootHi;
    uint32_t      rootLeaf;  // child word of the root if the whole tree is one leaf, else kWideNone
};

struct WideBuild
{
    std::vector<float4> nodes;
    std::vector<uint2>  bigLeaves;
    float4              rootLo, rootHi;
    uint32_t            rootLeaf = kWideNone;
};

// Host: 48-byte reference nodes -> wide records.
inline WideBuild buildWide(const BvhNode* nodes, size_t count)
{
    WideBuild             out;
    std::vector<uint32_t> wideIndex(count, 0);
    uint32_t              numInterior = 0;
    for (size_t i = 0; i < count; ++i)
        if (nodes[i].triangleCount == 0) wideIndex[i] = numInterior++;
    auto childWord = [&](size_t idx) -> uint32_t {
        const BvhNode& n = nodes[idx];
        if (n.triangleCount == 0) return wideIndex[idx];
        if (n.triangleCount <= 7 && n.trianglesOffset < (1u << 28)) return kWideLeafBit | ((n.triangleCount - 1) << 28) | n.trianglesOffset;
        out.bigLeaves.push_back(make_uint2(n.trianglesOffset, n.triangleCount));
        return kWideLeafBit | (7u << 28) | static_cast<uint32_t>(out.bigLeaves.size() - 1);
    };
    out.rootLo = make_float4(nodes[0].aabb.min.x, nodes[0].aabb.min.y, nodes[0].aabb.min.z, 0.0f);
    out.rootHi = make_float4(nodes[0].aabb.max.x, nodes[0].aabb.max.y, nodes[0].aabb.max.z, 0.0f);
    if (nodes[0].triangleCount > 0) out.rootLeaf = childWord(0);
    out.nodes.resize(4 * static_cast<size_t>(numInterior > 0 ? numInterior : 1));
    for (size_t i = 0; i < count; ++i)
    {
        const BvhNode& n = nodes[i];
        if (n.triangleCount > 0) continue;
        const size_t   c0 = i + 1, c1 = n.secondChildOffset;
        const BvhNode &a = nodes[c0], &b = nodes[c1];
        float4*        w = &out.nodes[4 * static_cast<size_t>(wideIndex[i])];
        w[0] = make_float4(a.aabb.min.x, a.aabb.min.y, a.aabb.min.z, bitsFloat(childWord(c0)));
        w[1] = make_float4(a.aabb.max.x, a.aabb.max.y, a.aabb.max.z, bitsFloat(n.splitAxis & 3u));
        w[2] = make_float4(b.aabb.min.x, b.aabb.min.y, b.aabb.min.z, bitsFloat(childWord(c1)));
        w[3] = make_float4(b.aabb.max.x, b.aabb.max.y, b.aabb.max.z, 0.0f);
    }
    if (out.bigLeaves.empty()) out.bigLeaves.push_back(make_uint2(0, 0));
    return out;
}

#if defined(__HIPCC__)
// The slab test split into its ray-only part: returns P and writes tmin.  Arithmetic and
// comparison order are those of slabTest() in rf_device.hpp.
__device__ __forceinline__ bool slabBounds(const RayPrep& r, float4 lo, float4 hi, float& tminOut)
{
    float       tmin = ((r.negX ? hi.x : lo.x) - r.origin.x) * r.invDir.x;
    float       tmax = ((r.negX ? lo.x : hi.x) - r.origin.x) * r.invDir.x;
    const float tymin = ((r.negY ? hi.y : lo.y) - r.origin.y) * r.invDir.y;
    const float tymax = ((r.negY ? lo.y : hi.y) - r.origin.y) * r.invDir.y;
    if ((tmin > tymax) || (tymin > tmax)) return false;
    tmin = maxf(tymin, tmin);
    tmax = minf(tymax, tmax);
    const float tzmin = ((r.negZ ? hi.z : lo.z) - r.origin.z) * r.invDir.z;
    const float tzmax = ((r.negZ ? lo.z : hi.z) - r.origin.z) * r.invDir.z;
    if ((tmin > tzmax) || (tzmin > tmax)) return false;
    tmin = maxf(tzmin, tmin);
    tmax = minf(tzmax, tmax);
    tminOut = tmin;
    return tmax > 0.0f;
}

#endif
} // namespace rf
