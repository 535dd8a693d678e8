// rf_bvh_gpu.hip -- binned-SAH BVH build on the GPU (gfx950), emitting the SAME depth-first 48-byte
// node array as the host builder rf_bvh.cpp and therefore as the reference's recursive builder
// (src/common/bvh.cpp:81-260; contract src/common/bvh.hpp:23-46).
//
// Why the node bytes can be reproduced in parallel: every quantity a node stores or a split decision
// reads is a function of the SET of triangles in the node, never of their order --
//   node / centroid / bucket boxes   component-wise min/max (exact, associative, commutative)
//   bucket of a triangle             12.0f * (c - min) / (max - min), per triangle
//   SAH sweep, split test            a fixed sequence over 12 buckets, evaluated by one thread per
//                                    node with the host builder's own expressions (rf_aabb.hpp)
// so the work is reorganised for the machine instead of recursing:
//
//   large phase   level-synchronous over all nodes with more than kSmallMax triangles: one pass over the
//                 triangle array per level accumulates per-node bucket counts / bounds / centroid
//                 bounds with integer atomics on order-preserving float encodings (aggregated per
//                 workgroup or per wave in LDS when the lanes share a node), one thread per node runs
//                 the SAH sweep and derives both children's boxes from the bucket boxes, and a global
//                 exclusive scan turns the split predicate into the permutation libstdc++'s
//                 std::partition performs (see "partition order" below; deterministic: the output
//                 does not depend on scheduling).
//   small phase   one wave per subtree of at most 64 triangles: a triangle per lane in registers,
//                 bucket reductions through wave-private LDS atomics, the same partition permutation
//                 from ballot ranks, explicit LIFO so that the subtree comes out in preorder.
//   numbering     subtree sizes bottom-up, preorder indices top-down (first child = index + 1, second
//                 child = index + 1 + size(first)), then every node is written to its final place.
//
// Partition order.  Closest-hit ties in t go to the first triangle tested (`t < tmax` is strict,
// wgsl:508), so the order of triangles INSIDE a multi-triangle leaf is observable when coincident
// triangles carry different attributes.  The host builder (rf_bvh.cpp) calls std::partition, i.e. on
// Linux libstdc++'s bidirectional algorithm: scan forward to the first element failing the predicate,
// backward to the last one passing it, swap, repeat.  Its net effect on a range with m passing
// elements is closed-form: the k-th failing element among the first m positions (ascending) trades
// places with the k-th passing element among the remaining positions (descending); everything else
// stays.  Both phases apply exactly that permutation (ranks from the scan / from ballots, the
// "k-th passing / failing element" looked up through the inverse of the stable order), and the
// 2-triangle case reproduces std::nth_element's insertion sort (swap iff the second centroid is
// smaller).  Result: triangleIndices -- and therefore a whole baked .pt -- equals the host builder's
// byte for byte (tests/test_gpu_bvh_build.py), coincident triangles included.
//
// Difference from the host builder that remains: the sign of a zero box coordinate when a node holds
// both -0.0f and +0.0f (the host keeps whichever its merge order met first).
#include "rf_bvh_gpu.hpp"

#include "rf_aabb.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <stdexcept>
#include <string>
#include <vector>

namespace rf
{
namespace
{
#define RF_HIP(expr)                                                                                          \
    do                                                                                                        \
    {                                                                                                         \
        const hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                                 \
            throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " in " #expr);      \
    } while (0)

constexpr uint32_t kBuckets = 12;
constexpr uint32_t kMaxLeaf = 255;       // bvh.cpp:203-206: more than this is always split
constexpr float    kTraversalCost = 0.5f;
constexpr uint32_t kSmallMax = 64;       // subtrees of at most this many triangles are built by one wave
constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr int      kThreads = 256;

// order-preserving float <-> uint (for atomicMin / atomicMax)
__host__ __device__ __forceinline__ uint32_t encodeFloat(float f)
{
    const uint32_t u = floatBits(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__host__ __device__ __forceinline__ float decodeFloat(uint32_t e)
{
    return bitsFloat((e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e);
}

struct EncBox
{
    uint32_t lo[3], hi[3];
};
__device__ __forceinline__ Box decodeBox(const EncBox& e)
{
    Box b;
    b.lo = vec3(decodeFloat(e.lo[0]), decodeFloat(e.lo[1]), decodeFloat(e.lo[2]));
    b.hi = vec3(decodeFloat(e.hi[0]), decodeFloat(e.hi[1]), decodeFloat(e.hi[2]));
    return b;
}
// A triangle in flight: bounds, centroid, source index.  Two float4 + one float2 streams, moved
// physically by every partition so that each pass reads them coalesced.
struct PrimStreams
{
    float4* a; // lo.xyz, centre.x
    float4* b; // hi.xyz, centre.y
    float2* c; // centre.z, source index (bits)
};

// One histogram cell per (node, bucket): count, triangle bounds, centroid bounds.
struct Bucket
{
    uint32_t count;
    EncBox   bounds, centers;
};
constexpr uint32_t kBucketWords = sizeof(Bucket) / 4; // 13

// A node of the large phase (more than kSmallMax triangles) at the current level.
struct LevelNode
{
    uint32_t first, count; // triangle range
    uint32_t node;         // index into the GNode array
    uint32_t depth;
    Box      box, centers;
    uint32_t axis;
    // filled by kSplitLarge
    uint32_t split;        // 1: partition by bucket <= bestBucket, 0: became a leaf
    uint32_t bestBucket, leftCount;
    uint32_t leftNext, rightNext; // index of the child in the next level's array, or kNone
};

// Node of the tree under construction (large phase: one each; small phase: the subtree root only).
struct GNode
{
    Box      box;
    uint32_t first, count; // leaf: triangle range
    uint32_t axis;         // kNone: leaf
    uint32_t left, right;  // GNode indices
    uint32_t size;         // nodes in the subtree
    uint32_t dfs;          // preorder index
    uint32_t smallPool;    // small subtree: offset of its nodes in the pool, else kNone
};

// A subtree handed to the small phase.
struct SmallTask
{
    uint32_t first, count, node, depth;
};

struct Counters
{
    uint32_t numNodes;   // GNodes allocated
    uint32_t nextCount;  // nodes of the next level
    uint32_t smallCount; // small tasks
    uint32_t poolUsed;   // nodes in the small pool
    uint32_t maxDepth;
    uint32_t pad[3];
};

__device__ __forceinline__ uint32_t bucketOf(float c, float cmin, float cmax)
{
    // bvh.cpp:152-155 -- float(12) * (c - min) / (max - min), evaluated left to right
    const float    q = static_cast<float>(kBuckets) * (c - cmin) / (cmax - cmin);
    const uint32_t b = static_cast<uint32_t>(q);
    return b < kBuckets - 1 ? b : kBuckets - 1;
}

// Leaf test of bvh.cpp:111-121 (single triangles are handled by the callers).
__device__ __forceinline__ bool mustBeLeaf(const Box& box, const Box& centers, int axis)
{
    return surfaceArea(box) == 0.0f || centers.lo[axis] == centers.hi[axis];
}

// The SAH sweep of bvh.cpp:162-206 over 12 filled buckets.  Returns false when the node becomes a leaf.
__device__ __forceinline__ bool chooseSplit(const uint32_t (&bucketCount)[kBuckets], const Box (&bucketBox)[kBuckets], const Box& nodeBox,
                                            uint32_t count, uint32_t& bestBucket)
{
    constexpr uint32_t kSplits = kBuckets - 1;
    float              cost[kSplits] = {};
    {
        uint32_t below = 0;
        Box      box;
        for (uint32_t i = 0; i < kSplits; ++i)
        {
            below += bucketCount[i];
            box = merge(box, bucketBox[i]);
            cost[i] += 1.0f * static_cast<float>(below) * surfaceArea(box);
        }
        uint32_t above = 0;
        Box      boxAbove;
        for (uint32_t i = kSplits; i > 0; --i)
        {
            above += bucketCount[i];
            boxAbove = merge(boxAbove, bucketBox[i]);
            cost[i - 1] += 1.0f * static_cast<float>(above) * surfaceArea(boxAbove);
        }
    }
    float best = FLT_MAX;
    bestBucket = kNone;
    for (uint32_t i = 0; i < kSplits; ++i)
    {
        if (cost[i] < best)
        {
            best = cost[i];
            bestBucket = i;
        }
    }
    // no finite cost (coordinates so large that the areas overflow): the reference asserts; defined here, in the host
    // builder (rf_bvh.cpp) as "the node becomes a leaf, whatever its size"
    if (bestBucket == kNone) return false;
    const float leafCost = 1.0f * static_cast<float>(count);
    const float splitCost = kTraversalCost + best / surfaceArea(nodeBox);
    return count > kMaxLeaf || splitCost < leafCost;
}

// LDS hand-over between the lanes of ONE wave (no other wave touches the data)
__device__ __forceinline__ void waveSync()
{
    // Compiler-level fence only: the LDS operations of one wave execute in program order, so nothing
    // more is needed for lane-to-lane hand-over.
    __atomic_signal_fence(__ATOMIC_SEQ_CST);
    __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------------
// setup
// ------------------------------------------------------------------------------------------------
__global__ void kSetup(const Positions* tris, uint32_t n, PrimStreams ps, int32_t* nodeOf, EncBox* rootBoxes)
{
    __shared__ uint32_t sBox[12];
    if (threadIdx.x < 12) sBox[threadIdx.x] = (threadIdx.x % 6) < 3 ? encodeFloat(FLT_MAX) : encodeFloat(-FLT_MAX);
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
    {
        const Box  b = boundsOf(tris[i]);
        const Vec3 c = centroid(b);
        ps.a[i] = make_float4(b.lo.x, b.lo.y, b.lo.z, c.x);
        ps.b[i] = make_float4(b.hi.x, b.hi.y, b.hi.z, c.y);
        ps.c[i] = make_float2(c.z, bitsFloat(i));
        nodeOf[i] = 0;
        const float lo[3] = {b.lo.x, b.lo.y, b.lo.z}, hi[3] = {b.hi.x, b.hi.y, b.hi.z}, cc[3] = {c.x, c.y, c.z};
        for (int k = 0; k < 3; ++k)
        {
            atomicMin(&sBox[k], encodeFloat(lo[k]));
            atomicMax(&sBox[3 + k], encodeFloat(hi[k]));
            atomicMin(&sBox[6 + k], encodeFloat(cc[k]));
            atomicMax(&sBox[9 + k], encodeFloat(cc[k]));
        }
    }
    __syncthreads();
    if (threadIdx.x < 12)
    {
        uint32_t* dst = reinterpret_cast<uint32_t*>(rootBoxes);
        if ((threadIdx.x % 6) < 3) atomicMin(dst + threadIdx.x, sBox[threadIdx.x]);
        else atomicMax(dst + threadIdx.x, sBox[threadIdx.x]);
    }
}

// The root: GNode 0, then either level 0 of the large phase, a small task, or a leaf.
__global__ void kRoot(uint32_t n, const EncBox* rootBoxes, GNode* nodes, LevelNode* level, SmallTask* small, Counters* ctr, int32_t* nodeOf)
{
    const Box box = decodeBox(rootBoxes[0]), centers = decodeBox(rootBoxes[1]);
    GNode&    g = nodes[0];
    g.box = box;
    g.first = 0;
    g.count = n;
    g.axis = kNone;
    g.left = g.right = kNone;
    g.size = 1;
    g.dfs = 0;
    g.smallPool = kNone;
    ctr->numNodes = 1;
    ctr->maxDepth = 1;
    if (n <= kSmallMax)
    {
        small[0] = SmallTask{0, n, 0, 1};
        ctr->smallCount = 1;
        ctr->nextCount = 0;
        return;
    }
    const int axis = maxDimension(centers);
    if (mustBeLeaf(box, centers, axis))
    {
        ctr->nextCount = 0; // one big leaf
        return;
    }
    LevelNode& l = level[0];
    l.first = 0;
    l.count = n;
    l.node = 0;
    l.depth = 1;
    l.box = box;
    l.centers = centers;
    l.axis = static_cast<uint32_t>(axis);
    ctr->nextCount = 1;
    (void)nodeOf;
}

// ------------------------------------------------------------------------------------------------
// large phase, per level
// ------------------------------------------------------------------------------------------------
__global__ void kClearBuckets(Bucket* buckets, uint32_t numNodes)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= numNodes * kBuckets) return;
    Bucket b;
    b.count = 0;
    for (int k = 0; k < 3; ++k)
    {
        b.bounds.lo[k] = b.centers.lo[k] = encodeFloat(FLT_MAX);
        b.bounds.hi[k] = b.centers.hi[k] = encodeFloat(-FLT_MAX);
    }
    buckets[i] = b;
}

__device__ __forceinline__ void accumulate(uint32_t* cell, const float (&lo)[3], const float (&hi)[3], const float (&cc)[3])
{
    atomicAdd(cell, 1u);
    for (int k = 0; k < 3; ++k)
    {
        atomicMin(cell + 1 + k, encodeFloat(lo[k]));
        atomicMax(cell + 4 + k, encodeFloat(hi[k]));
        atomicMin(cell + 7 + k, encodeFloat(cc[k]));
        atomicMax(cell + 10 + k, encodeFloat(cc[k]));
    }
}

// Bucket histogram of every active node: count, triangle bounds and centroid bounds per bucket.
// A workgroup whose 256 triangles all sit in one node (the common case on the upper levels, where
// one node spans thousands of triangles and global atomics on its 12 cells would serialise)
// accumulates in LDS and flushes 12 x 13 words once; mixed workgroups go to global memory directly.
__global__ __launch_bounds__(kThreads) void kHistogram(uint32_t n, PrimStreams ps, const int32_t* nodeOf, const LevelNode* level, Bucket* buckets)
{
    __shared__ uint32_t sCells[kBuckets * kBucketWords];
    __shared__ int32_t  sNode;
    __shared__ int32_t  sMixed;
    const uint32_t      i = blockIdx.x * blockDim.x + threadIdx.x;
    const int32_t       nd = i < n ? nodeOf[i] : -2;
    if (threadIdx.x == 0)
    {
        sNode = nd;
        sMixed = 0;
    }
    __syncthreads();
    if (nd != sNode && nd != -2) sMixed = 1; // benign race: any writer stores 1
    for (uint32_t k = threadIdx.x; k < kBuckets * kBucketWords; k += blockDim.x)
    {
        const uint32_t w = k % kBucketWords;
        sCells[k] = w == 0 ? 0u : ((w - 1) % 6 < 3 ? encodeFloat(FLT_MAX) : encodeFloat(-FLT_MAX));
    }
    __syncthreads();
    const bool uniform = sMixed == 0 && sNode >= 0;
    if (nd >= 0)
    {
        const LevelNode& l = level[nd];
        const float4     a = ps.a[i], b = ps.b[i];
        const float2     c = ps.c[i];
        const float      lo[3] = {a.x, a.y, a.z}, hi[3] = {b.x, b.y, b.z}, cc[3] = {a.w, b.w, c.x};
        const uint32_t   axis = l.axis;
        const uint32_t   bk = bucketOf(cc[axis], l.centers.lo[axis], l.centers.hi[axis]);
        uint32_t*        cell = uniform ? &sCells[bk * kBucketWords] : reinterpret_cast<uint32_t*>(&buckets[static_cast<size_t>(nd) * kBuckets + bk]);
        accumulate(cell, lo, hi, cc);
    }
    __syncthreads();
    if (uniform)
    {
        uint32_t* dst = reinterpret_cast<uint32_t*>(&buckets[static_cast<size_t>(sNode) * kBuckets]);
        for (uint32_t k = threadIdx.x; k < kBuckets * kBucketWords; k += blockDim.x)
        {
            const uint32_t w = k % kBucketWords, v = sCells[k];
            if (w == 0)
            {
                if (v) atomicAdd(dst + k, v);
            }
            else if ((w - 1) % 6 < 3)
            {
                if (v != encodeFloat(FLT_MAX)) atomicMin(dst + k, v);
            }
            else if (v != encodeFloat(-FLT_MAX)) atomicMax(dst + k, v);
        }
    }
}

// Creates the GNode of a child and routes it: next level (large), small task, or leaf.
__device__ __forceinline__ uint32_t makeChild(uint32_t gIdx, uint32_t first, uint32_t count, uint32_t depth, const Box& box, const Box& centers,
                                              GNode* nodes, LevelNode* nextLevel, SmallTask* small, Counters* ctr)
{
    GNode& g = nodes[gIdx];
    g.box = box;
    g.first = first;
    g.count = count;
    g.axis = kNone;
    g.left = g.right = kNone;
    g.size = 1;
    g.dfs = 0;
    g.smallPool = kNone;
    atomicMax(&ctr->maxDepth, depth);
    if (count <= kSmallMax)
    {
        small[atomicAdd(&ctr->smallCount, 1u)] = SmallTask{first, count, gIdx, depth};
        return kNone;
    }
    const int axis = maxDimension(centers);
    if (mustBeLeaf(box, centers, axis)) return kNone; // stays a (large) leaf
    const uint32_t slot = atomicAdd(&ctr->nextCount, 1u);
    LevelNode&     l = nextLevel[slot];
    l.first = first;
    l.count = count;
    l.node = gIdx;
    l.depth = depth;
    l.box = box;
    l.centers = centers;
    l.axis = static_cast<uint32_t>(axis);
    return slot;
}

// One thread per active node: SAH sweep, split decision, children (their boxes are unions of bucket boxes).
__global__ void kSplitLarge(uint32_t numLevelNodes, LevelNode* level, const Bucket* buckets, GNode* nodes, LevelNode* nextLevel, SmallTask* small,
                            Counters* ctr)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= numLevelNodes) return;
    LevelNode& l = level[i];
    uint32_t   bucketCount[kBuckets];
    Box        bucketBox[kBuckets];
    for (uint32_t k = 0; k < kBuckets; ++k)
    {
        const Bucket& bk = buckets[static_cast<size_t>(i) * kBuckets + k];
        bucketCount[k] = bk.count;
        bucketBox[k] = decodeBox(bk.bounds);
    }
    uint32_t best;
    if (!chooseSplit(bucketCount, bucketBox, l.box, l.count, best))
    {
        l.split = 0;
        l.leftNext = l.rightNext = kNone;
        return; // GNode keeps axis = kNone: leaf over [first, first + count)
    }
    uint32_t leftCount = 0;
    Box      leftBox, rightBox, leftCenters, rightCenters;
    for (uint32_t k = 0; k < kBuckets; ++k)
    {
        if (bucketCount[k] == 0) continue; // an empty bucket's boxes are the merge identity
        const Box cb = decodeBox(buckets[static_cast<size_t>(i) * kBuckets + k].centers);
        if (k <= best)
        {
            leftCount += bucketCount[k];
            leftBox = merge(leftBox, bucketBox[k]);
            leftCenters = merge(leftCenters, cb);
        }
        else
        {
            rightBox = merge(rightBox, bucketBox[k]);
            rightCenters = merge(rightCenters, cb);
        }
    }
    l.split = 1;
    l.bestBucket = best;
    l.leftCount = leftCount;
    const uint32_t base = atomicAdd(&ctr->numNodes, 2u);
    GNode&         g = nodes[l.node];
    g.axis = l.axis;
    g.left = base;
    g.right = base + 1;
    l.leftNext = makeChild(base, l.first, leftCount, l.depth + 1, leftBox, leftCenters, nodes, nextLevel, small, ctr);
    l.rightNext = makeChild(base + 1, l.first + leftCount, l.count - leftCount, l.depth + 1, rightBox, rightCenters, nodes, nextLevel, small, ctr);
}

// flag = 1 for triangles that go to the left child of a splitting node
__global__ void kFlags(uint32_t n, PrimStreams ps, const int32_t* nodeOf, const LevelNode* level, uint32_t* flags)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t nd = nodeOf[i];
    uint32_t      f = 0;
    if (nd >= 0 && level[nd].split)
    {
        const LevelNode& l = level[nd];
        const uint32_t   axis = l.axis;
        const float      c = axis == 0 ? ps.a[i].w : (axis == 1 ? ps.b[i].w : ps.c[i].x);
        f = bucketOf(c, l.centers.lo[axis], l.centers.hi[axis]) <= l.bestBucket ? 1u : 0u;
    }
    flags[i] = f;
}

// exclusive scan of `flags` (n <= 2^32): per-block scan + block sums, scan of the sums by one block, add.
constexpr int kScanItems = 4;
__global__ __launch_bounds__(kThreads) void kScanBlocks(const uint32_t* in, uint32_t* out, uint32_t n, uint32_t* blockSums)
{
    __shared__ uint32_t sWave[kThreads / 64];
    const uint32_t      base = (blockIdx.x * kThreads + threadIdx.x) * kScanItems;
    uint32_t            v[kScanItems], sum = 0;
    for (int k = 0; k < kScanItems; ++k)
    {
        v[k] = base + k < n ? in[base + k] : 0u;
        sum += v[k];
    }
    // inclusive wave scan of the per-thread sums
    uint32_t       incl = sum;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (int off = 1; off < 64; off <<= 1)
    {
        const uint32_t t = __shfl_up(incl, off);
        if (lane >= static_cast<uint32_t>(off)) incl += t;
    }
    if (lane == 63) sWave[wave] = incl;
    __syncthreads();
    uint32_t waveBase = 0;
    for (uint32_t w = 0; w < wave; ++w) waveBase += sWave[w];
    uint32_t run = waveBase + incl - sum;
    for (int k = 0; k < kScanItems; ++k)
    {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
    if (threadIdx.x == kThreads - 1) blockSums[blockIdx.x] = waveBase + incl;
}
__global__ __launch_bounds__(1024) void kScanSums(uint32_t* sums, uint32_t count)
{
    __shared__ uint32_t sWave[16];
    __shared__ uint32_t sCarry;
    if (threadIdx.x == 0) sCarry = 0;
    __syncthreads();
    for (uint32_t start = 0; start < count; start += 1024)
    {
        const uint32_t i = start + threadIdx.x;
        const uint32_t v = i < count ? sums[i] : 0u;
        uint32_t       incl = v;
        const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
        for (int off = 1; off < 64; off <<= 1)
        {
            const uint32_t t = __shfl_up(incl, off);
            if (lane >= static_cast<uint32_t>(off)) incl += t;
        }
        if (lane == 63) sWave[wave] = incl;
        __syncthreads();
        uint32_t waveBase = sCarry;
        for (uint32_t w = 0; w < wave; ++w) waveBase += sWave[w];
        if (i < count) sums[i] = waveBase + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) sCarry = waveBase + incl;
        __syncthreads();
    }
}
__global__ __launch_bounds__(kThreads) void kScanAdd(uint32_t* out, uint32_t n, const uint32_t* blockSums)
{
    const uint32_t base = (blockIdx.x * kThreads + threadIdx.x) * kScanItems;
    const uint32_t add = blockSums[blockIdx.x];
    for (int k = 0; k < kScanItems; ++k)
        if (base + k < n) out[base + k] += add;
}

// inv[stable-partition destination of i] = i for every triangle of a splitting node: position first + r holds the
// left-going triangle with r left-going ones before it, first + leftCount + r the right-going one of rank r
__global__ void kInverse(uint32_t n, const int32_t* nodeOf, const LevelNode* level, const uint32_t* flags, const uint32_t* scan, uint32_t* inv)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t nd = nodeOf[i];
    if (nd < 0 || !level[nd].split) return;
    const LevelNode& l = level[nd];
    const uint32_t   leftBefore = scan[i] - scan[l.first];
    inv[flags[i] ? l.first + leftBefore : l.first + l.leftCount + ((i - l.first) - leftBefore)] = i;
}

// libstdc++'s std::partition permutation of every splitting node (see the file header) + routing of the
// triangles to the next level's nodes.
__global__ void kScatter(uint32_t n, PrimStreams src, PrimStreams dst, const int32_t* nodeOf, int32_t* nodeOfNext, const LevelNode* level,
                         const uint32_t* flags, const uint32_t* scan, const uint32_t* inv)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int32_t nd = nodeOf[i];
    uint32_t      to = i;
    int32_t       next = -1;
    if (nd >= 0 && level[nd].split)
    {
        const LevelNode& l = level[nd];
        const uint32_t   leftBefore = scan[i] - scan[l.first];
        const uint32_t   p = i - l.first, m = l.leftCount;
        if (flags[i])
        {
            // a left-going triangle in the right part swaps with the k-th right-going one of the left part,
            // k = left-going triangles behind it
            if (p >= m) to = inv[l.first + m + (m - 1u - leftBefore)];
            next = l.leftNext == kNone ? -1 : static_cast<int32_t>(l.leftNext);
        }
        else
        {
            // a right-going triangle in the left part swaps with the k-th left-going one counted from the end,
            // k = right-going triangles before it
            if (p < m) to = inv[l.first + (m - 1u - (p - leftBefore))];
            next = l.rightNext == kNone ? -1 : static_cast<int32_t>(l.rightNext);
        }
    }
    dst.a[to] = src.a[i];
    dst.b[to] = src.b[i];
    dst.c[to] = src.c[i];
    nodeOfNext[to] = next;
}

// ------------------------------------------------------------------------------------------------
// small phase: one wave per subtree of at most 64 triangles
// ------------------------------------------------------------------------------------------------
struct LocalTask
{
    uint32_t first, count;
    uint32_t parent;          // local index of the node to patch when this is a right child
    uint32_t isRightAndDepth; // bit 15: right child; low bits: depth below the subtree root
};

constexpr int kSmallWaves = 4;

__global__ __launch_bounds__(64 * kSmallWaves) void kSmall(uint32_t numTasks, const SmallTask* tasks, PrimStreams ps, GNode* nodes, BvhNode* pool,
                                                           Counters* ctr, uint64_t* triangleIndices)
{
    __shared__ uint32_t  sCells[kSmallWaves][kBuckets * 7 + 12]; // per bucket: count + bounds; then node box + centre box
    __shared__ LocalTask sStack[kSmallWaves][kSmallMax];
    __shared__ float     sStage[kSmallWaves][10][kSmallMax];
    __shared__ uint32_t  sInv[kSmallWaves][kSmallMax];
    const uint32_t       wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    const uint32_t       t = blockIdx.x * kSmallWaves + wave;
    if (t >= numTasks) return;
    const SmallTask task = tasks[t];
    uint32_t*       cells = sCells[wave];
    LocalTask*      stack = sStack[wave];

    // my triangle
    float    lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0}, cc[3] = {0, 0, 0};
    uint32_t source = 0;
    if (lane < task.count)
    {
        const float4 a = ps.a[task.first + lane], b = ps.b[task.first + lane];
        const float2 c = ps.c[task.first + lane];
        lo[0] = a.x, lo[1] = a.y, lo[2] = a.z;
        hi[0] = b.x, hi[1] = b.y, hi[2] = b.z;
        cc[0] = a.w, cc[1] = b.w, cc[2] = c.x;
        source = floatBits(c.y);
    }

    // this subtree's nodes go to a private stretch of the pool, in preorder
    uint32_t poolBase = 0;
    if (lane == 0) poolBase = atomicAdd(&ctr->poolUsed, 2u * task.count - 1u);
    poolBase = __shfl(poolBase, 0);
    uint32_t numLocal = 0, maxDepth = 0;

    int sp = 0;
    if (lane == 0) stack[0] = LocalTask{0u, task.count, 0u, 0u};
    sp = 1;
    waveSync();

    while (sp > 0)
    {
        --sp;
        const LocalTask lt = stack[sp];
        const uint32_t  first = lt.first, count = lt.count, depth = lt.isRightAndDepth & 0x7FFFu;
        const bool      isRight = (lt.isRightAndDepth & 0x8000u) != 0;
        const uint32_t  me = numLocal++;
        maxDepth = max(maxDepth, depth);
        const bool in = lane >= first && lane < first + count;

        // node box and centroid box of the range (wave-private LDS atomics)
        if (lane < 12) cells[kBuckets * 7 + lane] = (lane % 6) < 3 ? encodeFloat(FLT_MAX) : encodeFloat(-FLT_MAX);
        for (uint32_t k = lane; k < kBuckets * 7; k += 64) cells[k] = (k % 7) == 0 ? 0u : (((k % 7) - 1) < 3 ? encodeFloat(FLT_MAX) : encodeFloat(-FLT_MAX));
        waveSync();
        if (in)
        {
            for (int k = 0; k < 3; ++k)
            {
                atomicMin(&cells[kBuckets * 7 + k], encodeFloat(lo[k]));
                atomicMax(&cells[kBuckets * 7 + 3 + k], encodeFloat(hi[k]));
                atomicMin(&cells[kBuckets * 7 + 6 + k], encodeFloat(cc[k]));
                atomicMax(&cells[kBuckets * 7 + 9 + k], encodeFloat(cc[k]));
            }
        }
        waveSync();
        Box box, centers;
        box.lo = vec3(decodeFloat(cells[kBuckets * 7 + 0]), decodeFloat(cells[kBuckets * 7 + 1]), decodeFloat(cells[kBuckets * 7 + 2]));
        box.hi = vec3(decodeFloat(cells[kBuckets * 7 + 3]), decodeFloat(cells[kBuckets * 7 + 4]), decodeFloat(cells[kBuckets * 7 + 5]));
        centers.lo = vec3(decodeFloat(cells[kBuckets * 7 + 6]), decodeFloat(cells[kBuckets * 7 + 7]), decodeFloat(cells[kBuckets * 7 + 8]));
        centers.hi = vec3(decodeFloat(cells[kBuckets * 7 + 9]), decodeFloat(cells[kBuckets * 7 + 10]), decodeFloat(cells[kBuckets * 7 + 11]));
        const int axis = maxDimension(centers);

        BvhNode out;
        out.aabb = toAabb(box);
        out.secondChildOffset = 0;
        bool     leaf = count == 1 || mustBeLeaf(box, centers, axis);
        bool     left = false; // my side when the node splits
        uint32_t leftCount = 0;
        if (!leaf)
        {
            const float myC = cc[axis];
            if (count == 2)
            {
                // bvh.cpp:126-137: nth_element around the middle = the smaller centroid first (ties keep order)
                // (through LDS rather than __shfl with a non-constant lane index)
                float* bc = &sStage[wave][0][0];
                bc[lane] = myC;
                waveSync();
                const float c0 = bc[first], c1 = bc[first + 1];
                waveSync();
                const bool  swap = c1 < c0;
                left = in && ((lane == first) != swap);
                leftCount = 1;
            }
            else
            {
                const uint32_t bk = bucketOf(myC, centers.lo[axis], centers.hi[axis]);
                if (in)
                {
                    atomicAdd(&cells[bk * 7], 1u);
                    for (int k = 0; k < 3; ++k)
                    {
                        atomicMin(&cells[bk * 7 + 1 + k], encodeFloat(lo[k]));
                        atomicMax(&cells[bk * 7 + 4 + k], encodeFloat(hi[k]));
                    }
                }
                waveSync();
                uint32_t bucketCount[kBuckets];
                Box      bucketBox[kBuckets];
                for (uint32_t k = 0; k < kBuckets; ++k)
                {
                    bucketCount[k] = cells[k * 7];
                    bucketBox[k].lo = vec3(decodeFloat(cells[k * 7 + 1]), decodeFloat(cells[k * 7 + 2]), decodeFloat(cells[k * 7 + 3]));
                    bucketBox[k].hi = vec3(decodeFloat(cells[k * 7 + 4]), decodeFloat(cells[k * 7 + 5]), decodeFloat(cells[k * 7 + 6]));
                }
                uint32_t best;
                if (!chooseSplit(bucketCount, bucketBox, box, count, best)) leaf = true;
                else
                {
                    left = in && bk <= best;
                    for (uint32_t k = 0; k <= best; ++k) leftCount += bucketCount[k];
                }
            }
        }

        if (leaf)
        {
            out.trianglesOffset = task.first + first;
            out.triangleCount = count;
            out.splitAxis = 0xFFFFFFFFu;
        }
        else
        {
            out.trianglesOffset = 0;
            out.triangleCount = 0;
            out.splitAxis = static_cast<uint32_t>(axis);
            // in-wave std::partition permutation of the range (file header): ranks from ballots, the inverse of
            // the stable order and the data through LDS
            const unsigned long long leftMask = __ballot(left), inMask = __ballot(in);
            const unsigned long long below = (1ull << lane) - 1ull;
            const uint32_t           leftRank = __popcll(leftMask & below), rightRank = __popcll(inMask & ~leftMask & below);
            uint32_t*                inv = sInv[wave];
            if (in) inv[left ? first + leftRank : first + leftCount + rightRank] = lane;
            waveSync();
            uint32_t to = lane;
            if (in)
            {
                const uint32_t p = lane - first;
                if (left && p >= leftCount) to = inv[first + leftCount + (leftCount - 1u - leftRank)];
                else if (!left && p < leftCount) to = inv[first + (leftCount - 1u - rightRank)];
            }
            waveSync();
            float* stage = &sStage[wave][0][0]; // [value][lane]
            for (int k = 0; k < 3; ++k)
            {
                stage[k * kSmallMax + to] = lo[k];
                stage[(3 + k) * kSmallMax + to] = hi[k];
                stage[(6 + k) * kSmallMax + to] = cc[k];
            }
            stage[9 * kSmallMax + to] = bitsFloat(source);
            waveSync();
            for (int k = 0; k < 3; ++k)
            {
                lo[k] = stage[k * kSmallMax + lane];
                hi[k] = stage[(3 + k) * kSmallMax + lane];
                cc[k] = stage[(6 + k) * kSmallMax + lane];
            }
            source = floatBits(stage[9 * kSmallMax + lane]);
            waveSync();
            if (lane == 0)
            {
                // LIFO: right first, so that the left subtree is numbered first (preorder)
                stack[sp] = LocalTask{first + leftCount, count - leftCount, me, 0x8000u | (depth + 1)};
                stack[sp + 1] = LocalTask{first, leftCount, me, depth + 1};
            }
            sp += 2;
        }
        if (lane == 0)
        {
            // dword by dword: a plain 48-byte struct store here (under exec = lane 0, values from uniform LDS
            // reads) crashes ROCm 7.2's instruction selection (InstrEmitter::AddRegisterOperand) about 4 times in 5
            volatile uint32_t* dst = reinterpret_cast<volatile uint32_t*>(&pool[poolBase + me]);
            const uint32_t*    srcWords = reinterpret_cast<const uint32_t*>(&out);
            for (int k = 0; k < 12; ++k) dst[k] = srcWords[k];
            if (isRight) pool[poolBase + lt.parent].secondChildOffset = me; // local; rebased when the subtree is placed
        }
        waveSync();
    }

    if (lane < task.count) triangleIndices[source] = task.first + lane;
    if (lane == 0)
    {
        GNode& g = nodes[task.node];
        g.size = numLocal;
        g.smallPool = poolBase;
        atomicMax(&ctr->maxDepth, task.depth + maxDepth);
    }
}

// triangles of large leaves (and of a root leaf) keep their final position too
__global__ void kLargeLeafIndices(uint32_t n, PrimStreams ps, const uint8_t* inSmall, uint64_t* triangleIndices)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || inSmall[i]) return;
    triangleIndices[floatBits(ps.c[i].y)] = i;
}
__global__ void kMarkSmall(uint32_t numTasks, const SmallTask* tasks, uint8_t* inSmall)
{
    const uint32_t t = blockIdx.x;
    if (t >= numTasks) return;
    for (uint32_t k = threadIdx.x; k < tasks[t].count; k += blockDim.x) inSmall[tasks[t].first + k] = 1;
}

// ------------------------------------------------------------------------------------------------
// numbering and output
// ------------------------------------------------------------------------------------------------
// GNodes are allocated level by level, so children always have larger indices than their parent:
// sizes bottom-up = descending index ranges, preorder top-down = ascending ranges.
__global__ void kSizes(GNode* nodes, uint32_t begin, uint32_t end)
{
    const uint32_t i = begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= end) return;
    GNode& g = nodes[i];
    if (g.axis != kNone) g.size = 1 + nodes[g.left].size + nodes[g.right].size;
}
__global__ void kPreorder(GNode* nodes, uint32_t begin, uint32_t end)
{
    const uint32_t i = begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= end) return;
    const GNode& g = nodes[i];
    if (g.axis == kNone) return;
    nodes[g.left].dfs = g.dfs + 1;
    nodes[g.right].dfs = g.dfs + 1 + nodes[g.left].size;
}
__global__ void kEmit(const GNode* nodes, uint32_t numNodes, const BvhNode* pool, BvhNode* out)
{
    const uint32_t i = blockIdx.x;
    if (i >= numNodes) return;
    const GNode& g = nodes[i];
    if (g.smallPool != kNone)
    {
        for (uint32_t k = threadIdx.x; k < g.size; k += blockDim.x)
        {
            BvhNode nd = pool[g.smallPool + k];
            if (nd.triangleCount == 0) nd.secondChildOffset += g.dfs;
            out[g.dfs + k] = nd;
        }
        return;
    }
    if (threadIdx.x != 0) return;
    BvhNode nd;
    nd.aabb = toAabb(g.box);
    if (g.axis == kNone)
    {
        nd.trianglesOffset = g.first;
        nd.secondChildOffset = 0;
        nd.triangleCount = g.count;
        nd.splitAxis = 0xFFFFFFFFu;
    }
    else
    {
        nd.trianglesOffset = 0;
        nd.secondChildOffset = nodes[g.right].dfs;
        nd.triangleCount = 0;
        nd.splitAxis = g.axis;
    }
    out[g.dfs] = nd;
}

template<typename T>
struct Dev
{
    T*   p = nullptr;
    void alloc(size_t n) { RF_HIP(hipMalloc(reinterpret_cast<void**>(&p), std::max<size_t>(n, 1) * sizeof(T))); }
    ~Dev()
    {
        if (p) (void)hipFree(p);
    }
};
inline dim3 gridFor(uint64_t n, int threads = kThreads) { return dim3(static_cast<uint32_t>((n + threads - 1) / threads)); }
} // namespace

Bvh buildBvhGpu(std::span<const Positions> triangles, int deviceOrdinal, float* buildMsOut)
{
    Bvh            out;
    const uint64_t n64 = triangles.size();
    if (n64 == 0) return out;
    if (n64 >= (1ull << 31)) throw std::runtime_error("buildBvhGpu: too many triangles");
    const uint32_t n = static_cast<uint32_t>(n64);
    int              deviceCount = 0;
    const hipError_t countErr = hipGetDeviceCount(&deviceCount);
    if (countErr != hipSuccess || deviceCount == 0)
        throw std::runtime_error(std::string("rayfinder_amd: no HIP device available (buildBvhGpu has no CPU fallback; use buildBvh): ") +
                                 hipGetErrorString(countErr) + ", " + std::to_string(deviceCount) + " device(s)");
    RF_HIP(hipSetDevice(deviceOrdinal));

    // capacities: every large node has > kSmallMax triangles, so a level holds < n / kSmallMax of them
    const uint32_t maxLevelNodes = n / kSmallMax + 2;
    // (disjoint ranges); small tasks hold at least one triangle each; GNodes = large nodes + small roots
    const uint32_t maxSmall = n + 2;
    const uint32_t maxGNodes = 2 * n + 64;
    Dev<Positions> dTris;
    Dev<float4>    a0, b0, a1, b1;
    Dev<float2>    c0, c1;
    Dev<int32_t>   nodeOf0, nodeOf1;
    Dev<uint32_t>  flags, scan, blockSums, inv;
    Dev<EncBox>    rootBoxes;
    Dev<GNode>     gnodes;
    Dev<LevelNode> levelA, levelB;
    Dev<Bucket>    buckets;
    Dev<SmallTask> small;
    Dev<Counters>  ctr;
    Dev<BvhNode>   pool, nodesOut;
    Dev<uint64_t>  triIdx;
    Dev<uint8_t>   inSmall;
    dTris.alloc(n);
    a0.alloc(n), b0.alloc(n), c0.alloc(n), a1.alloc(n), b1.alloc(n), c1.alloc(n);
    nodeOf0.alloc(n), nodeOf1.alloc(n);
    flags.alloc(n), scan.alloc(n), inv.alloc(n);
    const uint32_t scanBlocks = (n + kThreads * kScanItems - 1) / (kThreads * kScanItems);
    blockSums.alloc(scanBlocks);
    rootBoxes.alloc(2);
    gnodes.alloc(maxGNodes);
    levelA.alloc(maxLevelNodes), levelB.alloc(maxLevelNodes);
    buckets.alloc(static_cast<size_t>(maxLevelNodes) * kBuckets);
    small.alloc(maxSmall);
    ctr.alloc(1);
    pool.alloc(2ull * n);
    nodesOut.alloc(2ull * n);
    triIdx.alloc(n);
    inSmall.alloc(n);

    hipStream_t stream = nullptr; // default stream: this is a one-shot, synchronous build
    hipEvent_t  e0, e1;
    RF_HIP(hipEventCreate(&e0));
    RF_HIP(hipEventCreate(&e1));
    RF_HIP(hipMemcpy(dTris.p, triangles.data(), n64 * sizeof(Positions), hipMemcpyHostToDevice));
    RF_HIP(hipEventRecord(e0, stream));
    {
        EncBox init[2];
        for (int k = 0; k < 3; ++k)
        {
            init[0].lo[k] = init[1].lo[k] = encodeFloat(FLT_MAX);
            init[0].hi[k] = init[1].hi[k] = encodeFloat(-FLT_MAX);
        }
        RF_HIP(hipMemcpyAsync(rootBoxes.p, init, sizeof init, hipMemcpyHostToDevice, stream));
        RF_HIP(hipMemsetAsync(ctr.p, 0, sizeof(Counters), stream));
        RF_HIP(hipMemsetAsync(inSmall.p, 0, n, stream));
    }
    PrimStreams cur{a0.p, b0.p, c0.p}, nxt{a1.p, b1.p, c1.p};
    int32_t *   nodeOf = nodeOf0.p, *nodeOfNext = nodeOf1.p;
    LevelNode * level = levelA.p, *nextLevel = levelB.p;

    hipLaunchKernelGGL(kSetup, gridFor(n), dim3(kThreads), 0, stream, dTris.p, n, cur, nodeOf, rootBoxes.p);
    hipLaunchKernelGGL(kRoot, dim3(1), dim3(1), 0, stream, n, rootBoxes.p, gnodes.p, level, small.p, ctr.p, nodeOf);

    // large phase
    std::vector<uint32_t> levelEnds; // GNode count after each level (levels are contiguous GNode index ranges)
    levelEnds.push_back(1);
    Counters h{};
    RF_HIP(hipMemcpyAsync(&h, ctr.p, sizeof h, hipMemcpyDeviceToHost, stream));
    RF_HIP(hipStreamSynchronize(stream));
    uint32_t active = h.nextCount;
    while (active > 0)
    {
        if (active > maxLevelNodes) throw std::runtime_error("buildBvhGpu: level overflow");
        hipLaunchKernelGGL(kClearBuckets, gridFor(static_cast<uint64_t>(active) * kBuckets), dim3(kThreads), 0, stream, buckets.p, active);
        hipLaunchKernelGGL(kHistogram, gridFor(n), dim3(kThreads), 0, stream, n, cur, nodeOf, level, buckets.p);
        RF_HIP(hipMemsetAsync(&ctr.p->nextCount, 0, sizeof(uint32_t), stream));
        hipLaunchKernelGGL(kSplitLarge, gridFor(active, 64), dim3(64), 0, stream, active, level, buckets.p, gnodes.p, nextLevel, small.p, ctr.p);
        hipLaunchKernelGGL(kFlags, gridFor(n), dim3(kThreads), 0, stream, n, cur, nodeOf, level, flags.p);
        hipLaunchKernelGGL(kScanBlocks, dim3(scanBlocks), dim3(kThreads), 0, stream, flags.p, scan.p, n, blockSums.p);
        hipLaunchKernelGGL(kScanSums, dim3(1), dim3(1024), 0, stream, blockSums.p, scanBlocks);
        hipLaunchKernelGGL(kScanAdd, dim3(scanBlocks), dim3(kThreads), 0, stream, scan.p, n, blockSums.p);
        hipLaunchKernelGGL(kInverse, gridFor(n), dim3(kThreads), 0, stream, n, nodeOf, level, flags.p, scan.p, inv.p);
        hipLaunchKernelGGL(kScatter, gridFor(n), dim3(kThreads), 0, stream, n, cur, nxt, nodeOf, nodeOfNext, level, flags.p, scan.p, inv.p);
        RF_HIP(hipMemcpyAsync(&h, ctr.p, sizeof h, hipMemcpyDeviceToHost, stream));
        RF_HIP(hipStreamSynchronize(stream));
        if (h.numNodes > maxGNodes || h.smallCount > maxSmall) throw std::runtime_error("buildBvhGpu: node pool overflow");
        levelEnds.push_back(h.numNodes);
        active = h.nextCount;
        std::swap(cur, nxt);
        std::swap(nodeOf, nodeOfNext);
        std::swap(level, nextLevel);
    }

    // small phase
    const uint32_t numSmall = h.smallCount;
    if (numSmall)
    {
        hipLaunchKernelGGL(kMarkSmall, dim3(numSmall), dim3(64), 0, stream, numSmall, small.p, inSmall.p);
        hipLaunchKernelGGL(kSmall, dim3((numSmall + kSmallWaves - 1) / kSmallWaves), dim3(64 * kSmallWaves), 0, stream, numSmall, small.p, cur, gnodes.p,
                           pool.p, ctr.p, triIdx.p);
    }
    hipLaunchKernelGGL(kLargeLeafIndices, gridFor(n), dim3(kThreads), 0, stream, n, cur, inSmall.p, triIdx.p);

    // numbering: sizes bottom-up, preorder top-down, over the level ranges of the GNode array
    const uint32_t numG = h.numNodes;
    for (size_t l = levelEnds.size(); l-- > 0;)
    {
        const uint32_t begin = l == 0 ? 0 : levelEnds[l - 1], end = levelEnds[l];
        if (end > begin) hipLaunchKernelGGL(kSizes, gridFor(end - begin), dim3(kThreads), 0, stream, gnodes.p, begin, end);
    }
    for (size_t l = 0; l < levelEnds.size(); ++l)
    {
        const uint32_t begin = l == 0 ? 0 : levelEnds[l - 1], end = levelEnds[l];
        if (end > begin) hipLaunchKernelGGL(kPreorder, gridFor(end - begin), dim3(kThreads), 0, stream, gnodes.p, begin, end);
    }
    hipLaunchKernelGGL(kEmit, dim3(numG), dim3(64), 0, stream, gnodes.p, numG, pool.p, nodesOut.p);
    RF_HIP(hipGetLastError());
    RF_HIP(hipEventRecord(e1, stream));

    GNode root{};
    RF_HIP(hipMemcpy(&root, gnodes.p, sizeof root, hipMemcpyDeviceToHost));
    RF_HIP(hipMemcpy(&h, ctr.p, sizeof h, hipMemcpyDeviceToHost));
    out.nodes.resize(root.size);
    RF_HIP(hipMemcpy(out.nodes.data(), nodesOut.p, static_cast<size_t>(root.size) * sizeof(BvhNode), hipMemcpyDeviceToHost));
    std::vector<uint64_t> idx(n);
    RF_HIP(hipMemcpy(idx.data(), triIdx.p, n64 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    out.triangleIndices.assign(idx.begin(), idx.end());
    out.depth = static_cast<int>(h.maxDepth);
    if (buildMsOut)
    {
        RF_HIP(hipEventSynchronize(e1));
        RF_HIP(hipEventElapsedTime(buildMsOut, e0, e1));
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return out;
}
} // namespace rf
