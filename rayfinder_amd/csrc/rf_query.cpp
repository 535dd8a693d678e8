// rf_query.cpp -- host-side BVH query (see rf_query.hpp).  IEEE f32, no FMA contraction (-ffp-contract=off), glm's
// operation order through rf_math.hpp: results are bit-identical to the reference's rayIntersectBvh and to the GPU kernels.
#include "rf_query.hpp"

#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cstring>
#include <exception>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace rf
{
namespace
{
// A ray with everything the slab test needs that does not depend on the box.  `nearOfs[a]` / `farOfs[a]` are the float
// offsets inside an Aabb (min.x = 0 .. max.z = 6, pads at 3 and 7) of the plane the reference's bounds[dirNeg[a]] /
// bounds[1 - dirNeg[a]] select on axis a (ray_intersection.cpp:92-99,103-108).
struct PreparedRay
{
    Vec3     origin, direction, invDir;
    uint32_t nearOfs[3], farOfs[3];
    uint32_t negative[3];
};

PreparedRay prepare(Vec3 o, Vec3 d)
{
    PreparedRay r;
    r.origin = o;
    r.direction = d;
    r.invDir = vec3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    const float inv[3] = {r.invDir.x, r.invDir.y, r.invDir.z};
    for (uint32_t a = 0; a < 3; ++a)
    {
        r.negative[a] = inv[a] < 0.0f ? 1u : 0u;
        r.nearOfs[a] = (r.negative[a] ? 4u : 0u) + a;
        r.farOfs[a] = (r.negative[a] ? 0u : 4u) + a;
    }
    return r;
}

// rayIntersectAabb (ray_intersection.cpp:101-136): same products, same comparison order, std::max / std::min as the
// ternaries they are (NaN slabs of axis-parallel rays fall the same way).
bool boxHit(const PreparedRay& r, const Aabb& box, float rayTMax)
{
    const float* f = &box.min.x;
    float        tEnter = (f[r.nearOfs[0]] - r.origin.x) * r.invDir.x;
    float        tLeave = (f[r.farOfs[0]] - r.origin.x) * r.invDir.x;
    const float  yEnter = (f[r.nearOfs[1]] - r.origin.y) * r.invDir.y;
    const float  yLeave = (f[r.farOfs[1]] - r.origin.y) * r.invDir.y;
    if ((tEnter > yLeave) || (yEnter > tLeave)) return false;
    tEnter = maxf(yEnter, tEnter);
    tLeave = minf(yLeave, tLeave);
    const float zEnter = (f[r.nearOfs[2]] - r.origin.z) * r.invDir.z;
    const float zLeave = (f[r.farOfs[2]] - r.origin.z) * r.invDir.z;
    if ((tEnter > zLeave) || (zEnter > tLeave)) return false;
    tEnter = maxf(zEnter, tEnter);
    tLeave = minf(zLeave, tLeave);
    return (tEnter < rayTMax) && (tLeave > 0.0f);
}

// offsetRay (ray_intersection.cpp:17-35)
Vec3 nudgeAlongNormal(Vec3 p, Vec3 n)
{
    const auto one = [](float pc, float nc) {
        const int   step = static_cast<int>(256.0f * nc);
        int32_t     bits;
        std::memcpy(&bits, &pc, 4);
        bits += pc < 0 ? -step : step;
        float moved;
        std::memcpy(&moved, &bits, 4);
        return std::fabs(pc) < (1.0f / 32.0f) ? pc + (1.0f / 65536.0f) * nc : moved;
    };
    return vec3(one(p.x, n.x), one(p.y, n.y), one(p.z, n.z));
}

struct Candidate
{
    float t, u, v;
    Vec3  e1, e2, p0;
};

// rayIntersectTriangle (ray_intersection.cpp:38-90) up to the acceptance test; the hit point is formed once, for the final hit
// (it depends on the accepted triangle only: forming it per accepted candidate, as the reference does, yields the same value).
bool triangleHit(const PreparedRay& r, const float* p0f, const float* p1f, const float* p2f, float rayTMax, Candidate& c)
{
    constexpr float kEps = 0.00001f;
    const Vec3      p0 = vec3(p0f[0], p0f[1], p0f[2]), p1 = vec3(p1f[0], p1f[1], p1f[2]), p2 = vec3(p2f[0], p2f[1], p2f[2]);
    const Vec3      e1 = p1 - p0, e2 = p2 - p0;
    const Vec3      h = cross(r.direction, e2);
    const float     det = dot(e1, h);
    if (det > -kEps && det < kEps) return false;
    const float invDet = 1.0f / det;
    const Vec3  s = r.origin - p0;
    const float u = invDet * dot(s, h);
    if (u < 0.0f || u > 1.0f) return false;
    const Vec3  q = cross(s, e1);
    const float v = invDet * dot(r.direction, q);
    if (v < 0.0f || u + v > 1.0f) return false;
    const float t = invDet * dot(e2, q);
    if (!(t > kEps && t < rayTMax)) return false;
    c = Candidate{t, u, v, e1, e2, p0};
    return true;
}

[[noreturn]] void badTree(const char* what, uint64_t index) { throw std::runtime_error(std::string("malformed BVH: ") + what + " " + std::to_string(index)); }
} // namespace

bool intersectBvh(Vec3 origin, Vec3 direction, std::span<const BvhNode> nodes, TriangleSpan tris, float rayTMax, HostIntersection& out, HostBvhStats* stats)
{
    if (nodes.empty()) badTree("no nodes", 0);
    const PreparedRay ray = prepare(origin, direction);

    // pending far children: a small inline array, spilling into the heap for trees deeper than that
    constexpr uint32_t    kInline = 64;
    uint32_t              inlineStack[kInline];
    std::vector<uint32_t> spill;
    uint32_t              pending = 0, high = 0;
    const auto            pushPending = [&](uint32_t idx) {
        if (pending < kInline) inlineStack[pending] = idx;
        else
        {
            if (spill.size() <= pending - kInline) spill.resize(pending - kInline + 1);
            spill[pending - kInline] = idx;
        }
        ++pending;
        high = std::max(high, pending);
    };
    const auto popPending = [&]() -> uint32_t {
        --pending;
        return pending < kInline ? inlineStack[pending] : spill[pending - kInline];
    };

    uint32_t  visited = 0, tested = 0;
    uint32_t  best = 0xFFFFFFFFu;
    Candidate bestHit{};
    uint32_t  at = 0;
    const uint64_t numNodes = nodes.size();
    for (;;)
    {
        ++visited;
        const BvhNode& node = nodes[at];
        bool           descended = false;
        if (boxHit(ray, node.aabb, rayTMax))
        {
            if (node.triangleCount > 0)
            {
                if (static_cast<uint64_t>(node.trianglesOffset) + node.triangleCount > tris.count) badTree("leaf range past the triangle array at node", at);
                for (uint32_t k = 0; k < node.triangleCount; ++k)
                {
                    const uint32_t tri = node.trianglesOffset + k;
                    const uint8_t* rec = tris.data + static_cast<uint64_t>(tri) * tris.strideBytes;
                    const uint32_t vertexStride = tris.strideBytes / 3; // 12 (Positions) or 16 (PositionAttribute)
                    float          v[9];
                    std::memcpy(v, rec, 12);
                    std::memcpy(v + 3, rec + vertexStride, 12);
                    std::memcpy(v + 6, rec + 2 * vertexStride, 12);
                    ++tested;
                    Candidate c;
                    if (triangleHit(ray, v, v + 3, v + 6, rayTMax, c))
                    {
                        rayTMax = c.t;
                        bestHit = c;
                        best = tri;
                    }
                }
            }
            else
            {
                // near child first: the side the ray comes from along the node's split axis (dirNeg[splitAxis], .cpp:184-193)
                if (node.splitAxis > 2) badTree("interior node with split axis > 2 at node", at);
                const uint32_t first = at + 1, second = node.secondChildOffset;
                if (second <= first || second >= numNodes) badTree("child link out of range at node", at);
                const bool swapOrder = ray.negative[node.splitAxis] != 0;
                pushPending(swapOrder ? first : second);
                at = swapOrder ? second : first;
                descended = true;
            }
        }
        if (descended) continue;
        if (pending == 0) break;
        at = popPending();
    }

    if (stats)
    {
        stats->nodesVisited = visited;
        stats->triangleTests = tested;
        stats->stackHigh = high;
    }
    out.triangle = best;
    if (best == 0xFFFFFFFFu) return false;
    const Vec3 p = bestHit.p0 + bestHit.u * bestHit.e1 + bestHit.v * bestHit.e2;
    const Vec3 n = normalize(cross(bestHit.e1, bestHit.e2));
    out.p = nudgeAlongNormal(p, n);
    out.t = bestHit.t;
    out.u = bestHit.u;
    out.v = bestHit.v;
    return true;
}

namespace
{
// Static blocks of [0, n) over `threads` host threads; the first exception is rethrown on the caller's thread.
template<typename F>
void parallelBlocks(uint64_t n, uint32_t threads, F&& body)
{
    if (threads == 0) threads = std::max(1u, std::thread::hardware_concurrency());
    threads = static_cast<uint32_t>(std::min<uint64_t>(threads, std::max<uint64_t>(n, 1)));
    if (threads <= 1)
    {
        body(uint64_t{0}, n);
        return;
    }
    std::exception_ptr       failure;
    std::mutex               failureLock;
    std::vector<std::thread> pool;
    pool.reserve(threads);
    for (uint32_t w = 0; w < threads; ++w)
    {
        const uint64_t begin = n * w / threads, end = n * (w + 1) / threads;
        pool.emplace_back([&, begin, end] {
            try
            {
                body(begin, end);
            }
            catch (...)
            {
                std::lock_guard<std::mutex> g(failureLock);
                if (!failure) failure = std::current_exception();
            }
        });
    }
    for (std::thread& t : pool) t.join();
    if (failure) std::rethrow_exception(failure);
}
} // namespace

void intersectBvhBatch(const float* rays6, uint64_t n, std::span<const BvhNode> nodes, TriangleSpan triangles, float tMax, uint32_t threads, uint8_t* hit,
                       HostIntersection* out, HostBvhStats* stats)
{
    parallelBlocks(n, threads, [&](uint64_t begin, uint64_t end) {
        for (uint64_t i = begin; i < end; ++i)
        {
            const float*     r = rays6 + 6 * i;
            HostIntersection h{};
            HostBvhStats     st{};
            const bool       ok = intersectBvh(vec3(r[0], r[1], r[2]), vec3(r[3], r[4], r[5]), nodes, triangles, tMax, h, &st);
            if (hit) hit[i] = ok ? 1 : 0;
            if (out) out[i] = h;
            if (stats) stats[i] = st;
        }
    });
}

void bvhVisualizerPass(const Camera& camera, uint32_t width, uint32_t height, uint32_t rowBegin, uint32_t rowEnd, std::span<const BvhNode> nodes, TriangleSpan triangles,
                       uint32_t threads, uint32_t* nodesVisited, uint8_t* hit, float* t, uint32_t* triangleTests)
{
    rowEnd = std::min(rowEnd, height);
    if (rowBegin >= rowEnd || width == 0) return;
    // interleaved scanlines would balance better, but static row blocks are what SURVEY.md 8(d) specifies for the baseline
    parallelBlocks(rowEnd - rowBegin, threads, [&](uint64_t begin, uint64_t end) {
        for (uint64_t row = rowBegin + begin; row < rowBegin + end; ++row)
            for (uint32_t col = 0; col < width; ++col)
            {
                const float u = static_cast<float>(col) / static_cast<float>(width);
                const float v = 1.0f - static_cast<float>(row + 1) / static_cast<float>(height);
                // generateCameraRay (camera.cpp:44-52)
                const Vec3       d = normalize(camera.lowerLeftCorner + camera.horizontal * u + camera.vertical * v - camera.origin);
                HostIntersection h{};
                HostBvhStats     st{};
                const bool       ok = intersectBvh(camera.origin, d, nodes, triangles, FLT_MAX, h, &st);
                const size_t     idx = static_cast<size_t>(row) * width + col;
                if (nodesVisited) nodesVisited[idx] = st.nodesVisited;
                if (hit) hit[idx] = ok ? 1 : 0;
                if (t) t[idx] = ok ? h.t : 0.0f;
                if (triangleTests) triangleTests[idx] = st.triangleTests;
            }
    });
}
} // namespace rf
