// rf_bvh.cpp -- host BVH builder.
//
// Produces, byte for byte, the node array the reference's recursive builder produces
// (src/common/bvh.cpp:81-260): same split decisions (12-bucket SAH, traversal cost 0.5,
// intersection cost 1, forced split above 255 primitives, median split for 2 primitives, leaf on
// zero surface area / degenerate centroid extent / single primitive), same depth-first preorder
// numbering with the first child at index+1.  The formulation here is iterative with an explicit
// work stack (no recursion depth limit on degenerate inputs); the right child's index is patched
// into its parent when the right task is popped.
//
// std::partition and std::nth_element are the same libstdc++ algorithms the reference gets when
// built on Linux, so even the (otherwise library-defined) triangle order inside multi-triangle
// leaves matches.
#include "rf_bvh.hpp"

#include "rf_aabb.hpp"

#include <algorithm>
#include <limits>
#include <stdexcept>
#include <string>

namespace rf
{
namespace
{
struct Primitive
{
    Box         bounds;
    Vec3        center;
    std::size_t source;
};

constexpr std::size_t kBuckets = 12;
constexpr std::size_t kMaxLeaf = 255;
constexpr float       kTraversalCost = 0.5f;
constexpr float       kIntersectionCost = 1.0f;

inline std::size_t bucketIndex(const Primitive& p, int axis, const Box& centers)
{
    // bvh.cpp:152-155 -- float(12) * (c - min) / (max - min), evaluated left to right
    const float       q = static_cast<float>(kBuckets) * (p.center[axis] - centers.lo[axis]) / (centers.hi[axis] - centers.lo[axis]);
    const std::size_t b = static_cast<std::size_t>(q);
    return std::min(b, kBuckets - 1);
}

struct Task
{
    std::size_t first, count;  // primitive range
    std::size_t leafOffset;    // where this subtree's triangles start in leaf order
    std::size_t parent;        // node to patch when this is a right child
    bool        isRight;
    int         depth;
};
} // namespace

Bvh buildBvh(std::span<const Positions> triangles)
{
    Bvh               out;
    const std::size_t n = triangles.size();
    if (n == 0) return out;

    std::vector<Primitive> prims(n);
    for (std::size_t i = 0; i < n; ++i)
    {
        prims[i].bounds = boundsOf(triangles[i]);
        prims[i].center = centroid(prims[i].bounds);
        prims[i].source = i;
    }
    out.triangleIndices.resize(n);
    out.nodes.reserve(2 * n);

    std::vector<Task> work;
    work.push_back(Task{0, n, 0, 0, false, 1});

    while (!work.empty())
    {
        const Task task = work.back();
        work.pop_back();
        out.depth = std::max(out.depth, task.depth);

        const std::size_t nodeIdx = out.nodes.size();
        out.nodes.emplace_back();
        if (task.isRight) out.nodes[task.parent].secondChildOffset = static_cast<uint32_t>(nodeIdx);

        Primitive* const  p = prims.data() + task.first;
        const std::size_t count = task.count;

        Box nodeBox, centerBox;
        for (std::size_t i = 0; i < count; ++i)
        {
            nodeBox = merge(nodeBox, p[i].bounds);
            centerBox = merge(centerBox, p[i].center);
        }
        const int axis = maxDimension(centerBox);

        auto makeLeaf = [&]() {
            for (std::size_t i = 0; i < count; ++i) out.triangleIndices[p[i].source] = task.leafOffset + i;
            BvhNode& node = out.nodes[nodeIdx];
            node.aabb = toAabb(nodeBox);
            node.trianglesOffset = static_cast<uint32_t>(task.leafOffset);
            node.secondChildOffset = 0;
            node.triangleCount = static_cast<uint32_t>(count);
            node.splitAxis = static_cast<uint32_t>(-1);
        };

        if (surfaceArea(nodeBox) == 0.0f || centerBox.lo[axis] == centerBox.hi[axis] || count == 1)
        {
            makeLeaf();
            continue;
        }

        std::size_t split;
        if (count < 3)
        {
            split = count / 2;
            std::nth_element(p, p + split, p + count, [axis](const Primitive& a, const Primitive& b) {
                return a.center[axis] < b.center[axis];
            });
        }
        else
        {
            std::size_t bucketCount[kBuckets] = {};
            Box         bucketBox[kBuckets];
            for (std::size_t i = 0; i < count; ++i)
            {
                const std::size_t b = bucketIndex(p[i], axis, centerBox);
                bucketCount[b]++;
                bucketBox[b] = merge(bucketBox[b], p[i].bounds);
            }

            constexpr std::size_t kSplits = kBuckets - 1;
            float                 cost[kSplits] = {};
            {
                std::size_t below = 0;
                Box         box;
                for (std::size_t i = 0; i < kSplits; ++i)
                {
                    below += bucketCount[i];
                    box = merge(box, bucketBox[i]);
                    cost[i] += kIntersectionCost * static_cast<float>(below) * surfaceArea(box);
                }
                std::size_t above = 0;
                Box         boxAbove;
                for (std::size_t i = kSplits; i > 0; --i)
                {
                    above += bucketCount[i];
                    boxAbove = merge(boxAbove, bucketBox[i]);
                    cost[i - 1] += kIntersectionCost * static_cast<float>(above) * surfaceArea(boxAbove);
                }
            }
            float       best = std::numeric_limits<float>::max();
            std::size_t bestBucket = static_cast<std::size_t>(-1);
            for (std::size_t i = 0; i < kSplits; ++i)
            {
                if (cost[i] < best)
                {
                    best = cost[i];
                    bestBucket = i;
                }
            }
            const float leafCost = kIntersectionCost * static_cast<float>(count);
            const float splitCost = kTraversalCost + best / surfaceArea(nodeBox);
            if (!(count > kMaxLeaf || splitCost < leafCost))
            {
                makeLeaf();
                continue;
            }
            // Every SAH cost non-finite (e.g. coordinates around 1e20: surface areas overflow to inf, inf - inf
            // and 0 * inf give NaN, and no cost passes `<`): the reference asserts 0 < split < count
            // (bvh.cpp:215-216) and would recurse forever in a release build.  Defined behaviour here, in the GPU
            // builder: such a node becomes a leaf, whatever its size.
            if (bestBucket >= kSplits)
            {
                makeLeaf();
                continue;
            }
            Primitive* mid = std::partition(p, p + count, [&](const Primitive& q) {
                return bucketIndex(q, axis, centerBox) <= bestBucket;
            });
            split = static_cast<std::size_t>(mid - p);
            if (split == 0 || split == count) // cannot happen with a finite best cost (both sides of its split are non-empty)
            {
                makeLeaf();
                continue;
            }
        }

        BvhNode& node = out.nodes[nodeIdx];
        node.aabb = toAabb(nodeBox);
        node.trianglesOffset = 0;
        node.secondChildOffset = 0; // patched when the right child is numbered
        node.triangleCount = 0;
        node.splitAxis = static_cast<uint32_t>(axis);

        // LIFO: push right first so the left subtree is numbered first (preorder).
        work.push_back(Task{task.first + split, count - split, task.leafOffset + split, nodeIdx, true, task.depth + 1});
        work.push_back(Task{task.first, split, task.leafOffset, nodeIdx, false, task.depth + 1});
    }
    return out;
}

void validateScene(std::span<const BvhNode> nodes, std::size_t numTriangles, std::span<const VertexAttributes> vertexAttributes,
                   std::size_t numTextures)
{
    const std::size_t count = nodes.size();
    if (count == 0) throw std::runtime_error("scene has no BVH nodes");
    if (count > 0xFFFFFFFFull) throw std::runtime_error("more than 2^32 BVH nodes");
    for (std::size_t i = 0; i < count; ++i)
    {
        const BvhNode& n = nodes[i];
        if (n.triangleCount > 0)
        {
            if (static_cast<uint64_t>(n.trianglesOffset) + n.triangleCount > numTriangles)
                throw std::runtime_error("BVH leaf " + std::to_string(i) + " references triangles past the end of the triangle array");
        }
        else
        {
            if (n.splitAxis > 2) throw std::runtime_error("interior BVH node " + std::to_string(i) + " has an invalid split axis");
            if (!(i + 1 < n.secondChildOffset && n.secondChildOffset < count))
                throw std::runtime_error("interior BVH node " + std::to_string(i) + " has a second child that is not strictly ahead of its first child");
        }
    }
    if (!vertexAttributes.empty() && vertexAttributes.size() != numTriangles) throw std::runtime_error("position and vertex attribute counts differ");
    const std::size_t textures = std::max<std::size_t>(numTextures, 1);
    for (std::size_t i = 0; i < vertexAttributes.size(); ++i)
        if (vertexAttributes[i].textureIdx >= textures)
            throw std::runtime_error("triangle " + std::to_string(i) + " references texture " + std::to_string(vertexAttributes[i].textureIdx) + " of " +
                                     std::to_string(numTextures));
}
} // namespace rf
