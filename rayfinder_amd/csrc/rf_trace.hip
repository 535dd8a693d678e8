// rf_trace.hip -- the traversal kernels of the wavefront path tracer (MI355X, gfx950): kTraceWide (persistent waves over the wide record layouts of
// rf_wide.hpp: closest hit wgsl:370-429, any hit wgsl:321-368), kShadowFirstLook (the occluder cache without the traversal around it), the one-ray-per-thread
// reference-ordered kernels (scalar fallback, bvh-visualizer pass, ray queries) and -- experiment builds -- the packet kernel.  Launched from
// rf_renderer.hip through the accessors at the end of this file (rf_kernels.hpp).
#include "rf_kernels.hpp"

namespace rf
{
namespace
{
template<bool COUNT>
__global__ __launch_bounds__(kBlock) void kTraceClosest(DeviceScene scene, PathStreams ps, const uint32_t* queue,
                                                         const uint32_t* queueCount, DeviceCounters* counters)
{
    __shared__ uint32_t sStack[kLdsStack * kBlock];
    const uint32_t      i = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t      count = *queueCount;
    if (blockIdx.x * kBlock >= count) return;
    TraversalCounters tc;
    if (i < count)
    {
        const Vec3 o = load3(ps.rayO + i); // path state of the ray sits at its queue position
        const Vec3 d = load3(ps.rayD + i);
        ClosestHit h;
        traverse<false, COUNT>(scene, o, d, kTMax, &sStack[threadIdx.x], h, tc);
        if (tc.abandoned) atomicAdd(&counters->abandonedRays, 1ull);
        ps.hit[i] = make_float4(__uint_as_float(h.triangle), h.u, h.v, 0.0f); // (kShade rebuilds the offset hit point from it)
    }
    if (COUNT)
    {
        const unsigned long long nv = waveSum(tc.nodesVisited), tt = waveSum(tc.triangleTests);
        const uint32_t           sh = waveMax(tc.stackHigh);
        if (__lane_id() == 0)
        {
            atomicAdd(&counters->closestNodeVisits, nv);
            atomicAdd(&counters->closestTriangleTests, tt);
            atomicMax(&counters->stackHigh, sh);
        }
    }
    if (i == 0) atomicAdd(&counters->closestRays, static_cast<unsigned long long>(count));
}

template<bool COUNT>
__global__ __launch_bounds__(kBlock) void kTraceShadow(DeviceScene scene, SkyStateGpu sky, SunBasis sunBasis, PathStreams ps, const uint32_t* queue,
                                                        const uint32_t* queueCount, DeviceCounters* counters, uint32_t firstBounce)
{
    __shared__ uint32_t sStack[kLdsStack * kBlock];
    const uint32_t      i = blockIdx.x * kBlock + threadIdx.x;
    const uint32_t      count = *queueCount;
    if (blockIdx.x * kBlock >= count) return;
    TraversalCounters tc;
    if (i < count)
    {
        const uint32_t slot = queue[i];
        const Vec3     o = load3(ps.rayO + i);
        const Vec3     nz = load3(ps.noiseOut + i);
        const Vec3     l = sunSample(sky, sunBasis, nz.x, nz.y, nz.z);
        ClosestHit     h;
        const bool     occluded = traverse<true, COUNT>(scene, o, l, kTMax, &sStack[threadIdx.x], h, tc);
        if (tc.abandoned) atomicAdd(&counters->abandonedRays, 1ull);
        const float    visibility = occluded ? 0.0f : 1.0f;
        const Vec3     pend = load3(ps.pending + i); // by queue position (written there by kShade)
        const Vec3     rad0 = firstBounce ? vec3(0.0f, 0.0f, 0.0f) : load3(ps.rad + slot); // bounce 1: radiance is still 0 (wgsl:183)
        // wgsl:203  radiance += ((throughput*L)*reflectance) * visibility * SOLAR_INV_PDF
        const Vec3 add = (pend * visibility) * __uint_as_float(kSolarInvPdfBits);
        const Vec3 radiance = rad0 + add;
        ps.rad[slot] = make_float4(radiance.x, radiance.y, radiance.z, 0.0f);
    }
    if (COUNT)
    {
        const unsigned long long nv = waveSum(tc.nodesVisited), tt = waveSum(tc.triangleTests);
        if (__lane_id() == 0)
        {
            atomicAdd(&counters->shadowNodeVisits, nv);
            atomicAdd(&counters->shadowTriangleTests, tt);
        }
    }
    if (i == 0) atomicAdd(&counters->shadowRays, static_cast<unsigned long long>(count));
}

// cell of a point -> table index.  (Blocks of 4 x 4 x 4 neighbouring cells sharing 1 KB of the table -- the block hashed, the cell's place inside it from its low
// coordinate bits, so that a wave's rays read neighbouring lines -- measured -0.7 %: profiles/r04_occluder/occ_blocks.log.)
__device__ __forceinline__ uint32_t occluderCellIndex(const WideScene& wide, float ox, float oy, float oz)
{
    const uint32_t cx = static_cast<uint32_t>(__float2int_rd((ox - wide.rootLo.x) * wide.occScale)), cy = static_cast<uint32_t>(__float2int_rd((oy - wide.rootLo.y) * wide.occScale)),
                   cz = static_cast<uint32_t>(__float2int_rd((oz - wide.rootLo.z) * wide.occScale));
    return ((cx * 73856093u) ^ (cy * 19349663u) ^ (cz * 83492791u)) & wide.occMask;
}
__device__ __forceinline__ void loadOccluderCell(const uint32_t* cell, uint32_t (&e)[kOccSlots])
{
    if constexpr (kOccSlots == 1) e[0] = *cell;
    else if constexpr (kOccSlots == 2)
    {
        const uint2 v = *reinterpret_cast<const uint2*>(cell);
        e[0] = v.x, e[1] = v.y;
    }
    else
    {
        const uint4 v = *reinterpret_cast<const uint4*>(cell);
        e[0] = v.x, e[1] = v.y, e[kOccSlots > 2 ? 2 : 0] = v.z, e[kOccSlots > 3 ? 3 : 0] = v.w;
    }
}
__device__ __forceinline__ void storeOccluderCell(uint32_t* cell, const uint32_t (&e)[kOccSlots])
{
    if constexpr (kOccSlots == 1) *cell = e[0];
    else if constexpr (kOccSlots == 2) *reinterpret_cast<uint2*>(cell) = make_uint2(e[0], e[1]);
    else *reinterpret_cast<uint4*>(cell) = make_uint4(e[0], e[1], e[kOccSlots > 2 ? 2 : 0], e[kOccSlots > 3 ? 3 : 0]);
}

// ------------------------------------------------------------------------------------------------
// kTraceWide: persistent traversal over the 64-byte children-in-parent layout (rf_wide.hpp).
// Scheduling: a wave is 64 independent rays whose trip counts differ by an order of magnitude, and
// most visits are interior nodes.  Waves are persistent (grid = resident blocks), claim `chunk`
// queue entries per atomic, refill lanes whose ray has finished, and park lanes that reach a leaf
// until fewer than `leafVote` lanes are still descending, so that the Moller-Trumbore code runs for
// many lanes at once.  None of this changes any ray's own visit order.
//
// One step = one record = both children of an accepted interior node.  With hit(c) = P(c) &&
// tmin(c) < rayTMax (rf_wide.hpp), near/far in the reference's order (dirNeg[splitAxis]):
//     near hit, far hit : go to near, push (far, tmin(far))     reference: push far, visit near
//     near hit only     : go to near                             far would be popped and rejected later:
//                                                                rayTMax only ever shrinks
//     far hit only      : go to far, no stack traffic            reference: near rejected, far popped at once
//                                                                and tested against the same rayTMax
//     none              : pop until an entry passes tmin < rayTMax (the reference's test at pop time)
// The stack holds (child word, tmin) pairs, kWideLdsStack per lane in LDS ([depth][lane], ds_*_b64).
// A ray that would need more, and any ray that is not "regular" (axis-parallel / denormal / NaN,
// rf_wide.hpp), is redone whole by the reference-ordered scalar traversal over the 32-byte nodes
// (rf_device.hpp) -- same result by construction, and rare enough not to matter.
// ------------------------------------------------------------------------------------------------
// NEAREST_FIRST (any-hit only): visit the child with the smaller slab tmin first instead of the
// reference's split-axis order.  A shadow ray's answer is "does ANY triangle of any reachable leaf
// intersect", and with the fixed rayTMax of shadowRay (wgsl:323-368) the set of reachable leaves
// does not depend on the visit order, so the visibility bit is identical while occluded rays
// terminate after fewer fetches.  (Closest-hit keeps the reference order: ties in t are resolved
// by visit order.)
//
// COUNT && !NEAREST_FIRST is the reference-bookkeeping build: every far child is pushed (tmin = +inf
// when its box is missed) and counted when popped, so nodesVisited and the stack high-water mark
// equal the reference's exactly; it trades occupancy for a deeper LDS stack.
template<bool COUNT, bool NEAREST_FIRST>
constexpr int wideStackDepth()
{
    return (COUNT && !NEAREST_FIRST) ? 28 : kWideLdsStack;
}

// (experiment builds, RF_EXP_ANYHIT_WAVES=n: the any-hit kernels -- whose stack holds child words only, 4 bytes per entry: half the LDS of the closest-hit kernels' -- at n waves per SIMD)
#if defined(RF_EXP_ANYHIT_WAVES)
constexpr int kWideWavesAnyHit = RF_EXP_ANYHIT_WAVES;
#else
constexpr int kWideWavesAnyHit = kWideWaves;
#endif
template<bool ANY_HIT, bool COUNT, bool NEAREST_FIRST = false, int COMPACT = 0, bool DENSE_LEAVES = false>
__global__ __launch_bounds__(kBlock, (COUNT && !NEAREST_FIRST) ? 2 : ((ANY_HIT && !COUNT) ? kWideWavesAnyHit : kWideWaves)) void kTraceWide(DeviceScene scene, WideScene wide, SkyStateGpu sky, SunBasis sunBasis, PathStreams ps,
                                                                                        const uint32_t* queue, const uint32_t* queueCount, uint32_t* cursor,
                                                                                        DeviceCounters* counters, uint32_t refillMin, uint32_t leafVote,
                                                                                        uint32_t chunkMax, float tMax, uint32_t flags)
{
    constexpr int  kDepth = wideStackDepth<COUNT, NEAREST_FIRST>();
    constexpr bool kRefCount = COUNT && !NEAREST_FIRST;
    static_assert(!(COMPACT != 0 && COUNT), "the compact-record and quad-record variants have no counting build");
    static_assert(COMPACT >= 0 && COMPACT <= 6, "0: 64-byte records, 1: compact-capable, 2: 32-byte, 3: quad, 4: half-precision quad, 5: local-grid quad, 6: local-grid oct (closest-hit)");
    static_assert(!(COMPACT == 6 && ANY_HIT), "the oct records serve the closest-hit launches (the any-hit launches start at occluder-cache entries that name quad records)");
#if !defined(RF_EXP_LEGACY_LAYOUTS)
    static_assert(COMPACT != 1 && COMPACT != 2, "the compact-capable and the 32-byte records are experiment-build layouts (make EXP=RF_EXP_LEGACY_LAYOUTS)");
#endif
    constexpr bool kConservative = COMPACT == 4 || COMPACT == 5 || COMPACT == 6; // interior tests accept a superset; every leaf's EXACT box is applied at the leaf
    // (an any-hit kernel's entries are child words only -- kStackWordsOnly below --: 4 bytes each, 12 KB per workgroup instead of 24)
    using StackEntry = std::conditional_t<ANY_HIT && !kRefCount, uint32_t, uint2>;
    __shared__ StackEntry sStack[kDepth * kBlock];
    const uint32_t   count = *queueCount;
    const uint32_t   lane = __lane_id();
    const bool       shadowDirFromStream = flags & kFlagShadowDirFromStream;
    const bool       firstBounce = flags & kFlagFirstBounce;
    const bool       uniformFetch = flags & kFlagUniformFetch, uniformTri = flags & kFlagUniformTri;
    // Occluder cache (any-hit launches on the conservative records).  A shadow ray is answered as soon as ONE triangle stops it, and the rays that leave the
    // same few centimetres of the scene towards the 0.27-degree sun disc are stopped by the same few triangles.  The launch therefore keeps a hash grid over
    // cells of the scene's space (WideScene::occGrid; kOccSlots leaf words per cell, most recent first): a finished ray records the leaf in which it found its
    // occluder, and a NEW ray visits the leaves of its origin's cell FIRST, with the root waiting below them on its stack -- if one of them stops it, it is done
    // after a leaf visit or two instead of a walk from the root (atrium: 11.2 -> 0.8 interior steps per shadow ray).  A ray that tried its cell's leaves and reached the sun
    // drops the cell's first entry, so lit regions stop paying for stale entries.
    // The visibility bit is the reference's by the argument that lets an any-hit ray choose its visit order (NEAREST_FIRST above): a leaf visit here applies the
    // leaf's EXACT box with the reference's formula before any triangle is tested (the COMPACT 4 / 5 leaf phase below); a leaf whose own box passes is reached by
    // the reference too, because its ancestors' boxes contain it and the slab arithmetic is monotone in the planes (rf_wide.hpp) -- so the reference either tests
    // the same triangle or has found another one before: occluded either way; and a leaf visited a second time in the regular walk answers as it did the first
    // time.  Entries are hints only: any leaf word of this scene is a valid first visit, so racing writers, hash collisions and entries left from another sun
    // position cost time, never the result (tests: test_occluder_cache_is_invisible).
    constexpr bool kOccluderCache = ANY_HIT && !COUNT && (COMPACT == 3 || COMPACT == 4 || COMPACT == 5);
    // (the exact quad records test a leaf's box at its parent's step, not at the leaf: a launch of theirs that uses the cache applies the box at the leaf too, as
    // the conservative layouts always do -- a second, identical test for the leaves reached by the walk, THE test for the ones visited first)
    const bool     leafBoxAtLeaf = kConservative || (COMPACT == 3 && kOccluderCache && (flags & kFlagOccluderCache) != 0u && wide.occGrid != nullptr);
    const bool     occluderCache = kOccluderCache && (flags & kFlagOccluderCache) != 0u && wide.occGrid != nullptr;
    const auto occluderCell = [&](float ox, float oy, float oz) -> uint32_t { return occluderCellIndex(wide, ox, oy, oz); };
    constexpr uint32_t kNegTriedHint = 16u; // negMask: the ray started at a hint

    // The queue is cut into kShards contiguous ranges with one cursor each; a wave starts on the
    // shard of its block and moves on round-robin when a shard is dry.
    // entries per cursor claim: `chunkMax`, halved until every wave gets at least 8 claims (a short queue -- a small frame, a deep
    // bounce of one rank's shard -- ends in a tail of half-empty waves otherwise), but not below 64: a claim is a wave-wide stall
    // of a few microseconds, so fewer, larger claims win as long as the tail stays balanced
    uint32_t chunk = chunkMax;
    while (chunk > 64u && static_cast<unsigned long long>(chunk) * 8ull * gridDim.x * (kBlock / 64) > count) chunk >>= 1;
    const uint32_t shardLen = ((count + kShards - 1) / kShards + chunk - 1) / chunk * chunk;
    uint32_t       shard = blockIdx.x % kShards, shardsTried = 0;
    uint32_t       chunkPos = 0, chunkEnd = 0;
    bool           exhausted = count == 0;

    uint32_t  node = kNodeIdle;
    uint32_t  slot = 0;
    uint32_t  resultIndex = 0; // queue position of the lane's ray
    Vec3      pendingTerm{};   // ANY_HIT: the ray's NEE term (pending[resultIndex])
    // COMPACT: t-values of the x planes of the node the lane is about to visit, in hand when it enters the node straight from its
    // parent's step (rf_wide.hpp, compact-capable records); a lane that arrives from the stack or starts at the root reads them
    float tOuterLo = 0.0f, tOuterHi = 0.0f;
    bool  haveOuter = false;
    // COMPACT == 2 (32-byte records): the t-values of all six planes of that node's box
    BoxT  own{};
    // COMPACT == 4 (half-precision quad records): b = -(o / d) per axis, the addend of t' = fma(plane', 1/d, b)
    float hbx = 0.0f, hby = 0.0f, hbz = 0.0f;
    uint32_t lselX = 0u, lselY = 0u, lselZ = 0u; // COMPACT == 5 (local-grid quad records): per-axis v_perm_b32 selectors (see localEntryBounds)
    uint32_t octKey = 0u; // COMPACT == 6 (oct records): bits 5..0 = 16 x the field of the record's order table this ray reads, bits 8.. = 0x7777 when its positions are flipped (WideBuild::oct)
    uint32_t hrot = 0u, hrotY = 0u, hrotZ = 0u; // ... and (1/d.x < 0) << 4, (1/d.y < 0) << 4, (1/d.z < 0) << 4: rotate amounts that bring a plane word's NEAR plane into its low half
                                                // (three registers, not three fields of one: the two shifts that took the fields apart ran in every step)
    PackedRay pr{};        // origin and 1/direction in the pairings of the record (rf_wide.hpp)
    Vec3      rayDir{};    // for the triangle tests
    uint32_t  negMask = 0; // bit a: 1/direction[a] < 0 (reference child order); bit 3: class B ray (rf_wide.hpp)
    float     rayTMax = tMax;
    // Closest-hit launches keep the stack top as a BYTE offset into sStack (lane * 8 + depth * kBlock * 8): a push is one ds_write + one add, no
    // shift-or for the address, and -- in the quad steps -- one bound check per step instead of one per push: closest-hit launches -1.5 % (round 4,
    // gpurun_out A/B in profiles/r04_lanes).  The any-hit launches measured +2.5 % with it and keep the plain depth, as do the counting builds
    // (they report it).
    // ---- Eager leaves (closest-hit launches of scenes without long leaves; round 5).  A closest-hit ray's stack is full of leaves: the far child of a step near the bottom
    // of the tree IS a leaf, and after a leaf phase a third of the lanes that were served hold the next leaf straight off their stack.  They used to sit through the next descend
    // loop -- at least one trip, usually several -- before the next leaf phase took them.  Now (a) the leaf phase repeats while kLeafRepeat or more lanes stand at a leaf, and
    // (b) a descend loop that fewer than `leafVote` lanes would enter is skipped when that many lanes wait at a leaf.  Scheduling only: every ray still visits its nodes and
    // leaves in its own order.  Closest-hit launches -4.4 % on the plain atrium, neutral on the out-of-cache one (profiles/r05_leafrep); the dense (lane, triangle) leaf phase
    // of the scenes with long leaves (DENSE_LEAVES) loses 6 % with it -- a phase of its kind wants many parked lanes -- and the any-hit launches stop at their first hit:
    // both keep the plain schedule.  (As a run-time threshold in the launch flags the loop cost 4 % by its presence; a compile-time constant costs nothing.)
    constexpr bool     kEagerLeaves = !ANY_HIT && !COUNT && !DENSE_LEAVES;
#if defined(RF_EXP_LEAF_REPEAT)
    constexpr uint32_t kLeafRepeat = RF_EXP_LEAF_REPEAT;
#else
    constexpr uint32_t kLeafRepeat = 8u;
#endif // (6 ... 12, and 2 ... 16 for the skipped descend loop alone, measure the same: profiles/r05_leafrep/ab_thresholds.log)
    constexpr bool kPtrStack = !COUNT && !ANY_HIT;
    const int     spBase = kPtrStack ? static_cast<int>(threadIdx.x * sizeof(uint2)) : 0;
    constexpr int kSpStep = kPtrStack ? static_cast<int>(kBlock * sizeof(uint2)) : 1;
    constexpr int kSpLimit = kPtrStack ? kDepth * static_cast<int>(kBlock * sizeof(uint2)) : kDepth; // (depth == kDepth <=> offset >= this: lane * 8 < kBlock * 8)
    int       stackSize = spBase;
    const auto stackAt = [&](int s) -> StackEntry& {
        if constexpr (kPtrStack) return *reinterpret_cast<StackEntry*>(reinterpret_cast<char*>(sStack) + s);
        else return sStack[s * kBlock + threadIdx.x];
    };
    bool      needScalar = false; // irregular ray or stack overflow: redo with the scalar traversal
    // An any-hit ray's rayTMax never changes, so an entry that passed `tmin < rayTMax` when it was pushed passes it when it is popped: such a
    // kernel keeps only the words on its stack (no tmin to select, store and compare) -- except the reference-bookkeeping build, which
    // pushes missed children with tmin = +inf to count them.
    constexpr bool kStackWordsOnly = ANY_HIT && !kRefCount;
    static_assert(std::is_same_v<StackEntry, uint32_t> == kStackWordsOnly, "the LDS stack's entry type follows kStackWordsOnly");
    // ---- Rays that need more than the LDS stack holds.  Until round 4 such a ray was redone whole by the scalar traversal (one lane, the
    // reference-ordered kernel over the 32-byte nodes): fine at 0.01 % of the rays (the plain atrium), a cliff at 2.6 % (the atrium with clutter, whose
    // long diagonal boxes keep many candidates alive: closest-hit launches 3.2 x longer than with the binary records, which push at most one entry per
    // step).  Now a full LDS stack EVICTS its kEvict oldest entries -- the ones needed last -- to a per-lane scratch array and moves the rest down; when the
    // LDS stack runs empty the youngest evicted block comes back.  Same entries, same order, nothing recomputed; only a ray that would need more than
    // kDepth + kEvict * kSpillBlocks pending entries still takes the scalar traversal.  The number of evicted entries rides in bits 8.. of negMask.
    constexpr bool kSpill = !kRefCount;
    constexpr int  kEvict = kDepth >= 9 ? 6 : (kDepth > 4 ? kDepth - 3 : 1), kSpillBlocks = 36 / kEvict; // (6 x 6 by default; the stress build with a 6-entry LDS stack -- make EXP=RF_EXP_STACK=6 -- evicts 3 at a time, all the time)
    static_assert(kEvict >= 3 && kEvict <= kDepth - 2, "a quad step checks the bound once (depth < kDepth - 2) and then pushes up to three entries: an eviction must make room for all three");
    using SpillEntry = std::conditional_t<kStackWordsOnly, uint32_t, uint2>;
    SpillEntry spillBuf[kSpill ? kEvict * kSpillBlocks : 1];
    const auto slotS = [&](int i) -> int { return kPtrStack ? spBase + i * kSpStep : i; };
    const auto evict = [&]() -> bool {
        if constexpr (!kSpill) return false;
        const uint32_t spilled = negMask >> 8;
        if (spilled + kEvict > static_cast<uint32_t>(kEvict * kSpillBlocks)) return false;
        for (int i = 0; i < kEvict; ++i)
        {
            if constexpr (kStackWordsOnly) spillBuf[spilled + i] = stackAt(slotS(i));
            else spillBuf[spilled + i] = stackAt(slotS(i));
        }
        const int depth = kPtrStack ? (stackSize - spBase) / kSpStep : stackSize;
        for (int i = kEvict; i < depth; ++i)
        {
            if constexpr (kStackWordsOnly) stackAt(slotS(i - kEvict)) = stackAt(slotS(i));
            else stackAt(slotS(i - kEvict)) = stackAt(slotS(i));
        }
        stackSize -= kEvict * kSpStep;
        negMask += static_cast<uint32_t>(kEvict) << 8;
        return true;
    };
    // (the LDS stack is empty and entries are waiting in scratch: the youngest block comes back.  popNext() does not look at the scratch area -- it is the
    // hot path -- so a lane whose LDS stack ran dry reports "done"; the write-back block below, which every finished lane passes once, sends a lane with
    // evicted entries back to work instead)
    const auto unspill = [&]() {
        negMask -= static_cast<uint32_t>(kEvict) << 8;
        const uint32_t spilled = negMask >> 8;
        for (int i = 0; i < kEvict; ++i)
        {
            if constexpr (kStackWordsOnly) stackAt(slotS(i)) = spillBuf[spilled + i];
            else stackAt(slotS(i)) = spillBuf[spilled + i];
        }
        stackSize = slotS(kEvict);
    };
    auto      push = [&](uint32_t word, float tmin) -> bool {
        if (stackSize >= kSpLimit && !evict()) return false;
        if constexpr (kStackWordsOnly) stackAt(stackSize) = word;
        else stackAt(stackSize) = make_uint2(word, __float_as_uint(tmin));
        stackSize += kSpStep;
        return true;
    };
    auto      pushUnchecked = [&](uint32_t word, float tmin) {
        if constexpr (kStackWordsOnly) stackAt(stackSize) = word;
        else stackAt(stackSize) = make_uint2(word, __float_as_uint(tmin));
        stackSize += kSpStep;
    };
    ClosestHit        best{};
    bool              occluded = false;
    TraversalCounters tc;                                           // COUNT: totals of this lane's finished rays
    uint32_t          rayNodes = 0, rayTris = 0, rayStackHigh = 0;  // COUNT: the ray in flight
    uint32_t          recordFetches = 0;
    uint32_t          wDescend = 0, wLeaf = 0, wLeafPhase = 0, wRefill = 0, wPop = 0, wOuter = 0; // COUNT: loop trips
#if defined(RF_EXP_PHASE)
    constexpr bool kPhase = true; // experiment build: the wave-trip / lane-trip counters of the COUNT build in EVERY kTraceWide (RF_DEBUG_COUNTERS prints them)
    uint32_t       phaseTris = 0, phaseLeafWave = 0, phaseOccTried = 0, phaseOccHit = 0, phaseOccluded = 0;
    uint32_t       phaseParked = 0, phaseIdle = 0, phaseLeafInterior = 0, phaseLeafIdle = 0;
    bool           phaseFromCache = false;
#else
    constexpr bool kPhase = COUNT;
#endif

    // Pop entries until one passes `tmin < rayTMax` (the reference's box test at pop time).
    auto popNext = [&]() {
        if (COMPACT != 0) haveOuter = false;
        node = kNodeDone;
        if constexpr (kStackWordsOnly)
        {
            if (stackSize > spBase)
            {
                stackSize -= kSpStep;
                node = stackAt(stackSize);
                if (COUNT) ++wPop;
            }
            return;
        }
        else
        {
            while (stackSize > spBase)
            {
                stackSize -= kSpStep;
                uint2 e = stackAt(stackSize);
                asm volatile("" : "+v"(e.x), "+v"(e.y)); // one ds_read_b64 (not tmin first, word after the loop)
                if (COUNT) ++wPop;
                if (kRefCount) ++rayNodes;
                if (__uint_as_float(e.y) < rayTMax)
                {
                    node = e.x;
                    break;
                }
            }
        }
    };

    for (;;)
    {
        if (kPhase) ++wOuter;
        // ---- refill idle lanes from the wave's chunk
        const unsigned long long idleMask = __ballot(node == kNodeIdle);
        const uint32_t           idleCount = __popcll(idleMask);
        if (!exhausted && idleCount >= refillMin)
        {
            if (kPhase) ++wRefill;
            // queue positions for the idle lanes, in lane order; a refill that reaches the end of the wave's chunk goes on in the
            // next one (it used to stop there and leave the remaining lanes idle until the next refill: one refill in three)
            const uint32_t rankInIdle = __popcll(idleMask & ((1ull << lane) - 1ull));
            uint32_t       assigned = 0, myPos = 0xFFFFFFFFu;
            while (assigned < idleCount)
            {
                while (chunkPos == chunkEnd && !exhausted)
                {
                    const uint32_t shardBegin = shard * shardLen, shardEnd = min(shardBegin + shardLen, count);
                    uint32_t       base = 0;
                    if (lane == 0) base = shardBegin < count ? atomicAdd(cursor + shard * kLineWords, chunk) : shardLen;
                    // (readfirstlane, not a shuffle: the claim state -- shard, chunkPos, chunkEnd, shardsTried, exhausted -- is wave-uniform, and the compiler can only
                    // keep it in SGPRs and run these loops on the scalar unit if it can SEE that: behind a shuffle it held four VGPRs and ran exec-mask loops.  Every lane
                    // is enabled here -- the refill's condition is wave-uniform -- so the first lane is lane 0.)
                    base = shardBegin + static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(base)));
                    if (base >= shardEnd)
                    {
                        shard = (shard + 1) % kShards;
                        if (++shardsTried == kShards) exhausted = true;
                    }
                    else
                    {
                        chunkPos = base;
                        chunkEnd = min(base + chunk, shardEnd);
                    }
                }
                if (chunkPos == chunkEnd) break; // the queue is dry
                const uint32_t take = min(idleCount - assigned, chunkEnd - chunkPos);
                if (rankInIdle - assigned < take) myPos = chunkPos + (rankInIdle - assigned); // (unsigned: false for ranks below `assigned`)
                chunkPos += take;
                assigned += take;
            }
            if (node == kNodeIdle && myPos != 0xFFFFFFFFu)
            {
                // the ray's state sits at its QUEUE position: the lanes of a refill read consecutive elements (coalesced), and
                // the closest-hit launch does not read the queue itself at all
                resultIndex = myPos;
                // (closest-hit launches: a list of queue positions to visit in ITS order -- today only the ray-query path's RF_DEBUG_QUERY_LIST, tools/gpu_sort_potential.py)
                if (!ANY_HIT && wide.rayList != nullptr) resultIndex = wide.rayList[myPos];
                bool triedCell = false;
                if constexpr (kOccluderCache)
                {
                    if (wide.rayList != nullptr)
                    {
                        const uint32_t e = wide.rayList[myPos]; // behind kShadowFirstLook: the rays it could not answer, by queue position
                        resultIndex = e & 0x7FFFFFFFu;
                        triedCell = (e >> 31) != 0u;
                    }
                }
                if (ANY_HIT) slot = loadQ(queue + resultIndex); // the radiance sum and the blue-noise pair are the path's: by slot
                // the NEE term this ray decides about: read with the rest of the ray (consecutive queue positions: coalesced) instead of
                // at write-back, where every finishing lane gathered its own 12 bytes and the wave waited for them
                if (ANY_HIT) pendingTerm = load3s(ps.pending + resultIndex);
                // (a pinhole camera's primary rays: one origin, a kernel argument -- 12 of the 40 bytes a path costs kRaygen, and the read back here)
                // (the direction is requested FIRST: behind the origin's wave-uniform branch the compiler waited for the origin before it asked for the direction -- two
                // memory round trips per refill of a closest-hit launch instead of one)
                Vec3 dir{};
                if (!(ANY_HIT && !shadowDirFromStream)) dir = load3s(ps.rayD + resultIndex);
                const Vec3 o = (!ANY_HIT && (flags & kFlagConstOrigin) != 0u) ? vec3(wide.constOriginX, wide.constOriginY, wide.constOriginZ) : load3s(ps.rayO + resultIndex);
                if (ANY_HIT && !shadowDirFromStream)
                {
                    const Vec3 nz = load3s(ps.noiseOut + resultIndex);
                    dir = sunSample(sky, sunBasis, nz.x, nz.y, nz.z);
                }
                // (round 6 measured 1 / direction written by kShade next to the direction and read here instead of the three divides: closest-hit launches +1 %, kShade +4 % --
                // the refill is not short of VALU slots, it waits for its loads: profiles/r06_lanes/README.md)
                const RayPrep ray = prepareRay(o, dir);
                pr = packRay(ray);
                rayDir = dir;
                rayTMax = tMax;
                stackSize = spBase;
#if defined(RF_EXP_PHASE)
                phaseFromCache = false;
#endif
                best.triangle = kMiss;
                occluded = false;
                if (COMPACT == 1) haveOuter = false;
                if (COMPACT == 2)
                {
                    // the root's own box is a kernel argument: no fetch for it
                    own = boxPlaneT(pr, make_float4(wide.rootLo.x, wide.rootLo.y, wide.rootHi.x, wide.rootHi.y), wide.rootLo.z, wide.rootHi.z);
                    haveOuter = true;
                }
                rayNodes = 1; // the root visit (wgsl:379-382)
                rayTris = 0;
                rayStackHigh = 0;
                // ---- classification (rf_wide.hpp) and, on the conservative records, the preconditions of their margin proofs.  Round 6: ONE test first that nearly every ray
                // passes -- all three 1 / d of ordinary magnitude (which excludes NaN, infinity, zero), the origin within the bound (which excludes NaN and, the bound being
                // below 1e30, anything classifyRay() would call non-finite): such a ray is class A, needs no scalar traversal, no replacement of infinities and (below) no root
                // test.  Only the others run the full classification -- the same decisions as before, moved off the path of the ordinary ray (nine compares instead of ~60
                // instructions of every refill trip).
                bool fastRay = false;
                if constexpr (kConservative && !COUNT)
                {
                    const float ax = fabsf(ray.invDir.x), ay = fabsf(ray.invDir.y), az = fabsf(ray.invDir.z);
                    fastRay = ax >= 1e-18f && ax <= 1e18f && ay >= 1e-18f && ay <= 1e18f && az >= 1e-18f && az <= 1e18f && fabsf(o.x) <= wide.originBound && fabsf(o.y) <= wide.originBound &&
                              fabsf(o.z) <= wide.originBound && wide.originBound < 1e30f;
                }
                uint32_t rayClass = kRayPlain;
                bool     rootOk = true;
                needScalar = false;
                if (__builtin_expect(!fastRay, 0))
                {
                    rayClass = classifyRay(ray);
                    needScalar = rayClass == kRayIrregular;
                    if constexpr (kConservative)
                    {
                        // the margin of the half-precision / local-grid planes covers origins within wide.originBound and 1/direction components of
                        // ordinary magnitude (or +-inf: those axes drop out as NaNs): anything else takes the scalar traversal
                        const auto ordinary = [](float inv) { const float a = fabsf(inv); return (a >= 1e-18f && a <= 1e18f) || a == __uint_as_float(0x7F800000u); };
                        const bool inside = fabsf(o.x) <= wide.originBound && fabsf(o.y) <= wide.originBound && fabsf(o.z) <= wide.originBound;
                        if (!(inside && ordinary(ray.invDir.x) && ordinary(ray.invDir.y) && ordinary(ray.invDir.z))) needScalar = true;
                        // An infinite 1/d (axis-parallel ray, class B) is replaced by +-1e30 IN THE CONSERVATIVE TESTS: the margin argument
                        // does not depend on the size of 1/d, so the ray is still accepted wherever the reference accepts it (strictly inside
                        // the slab: [-huge, +huge]; within the margin of a plane: accepted as well) and rejected when it is outside the
                        // conservative slab by more than rounding -- instead of being left unconstrained on that axis, which sent such rays
                        // through whole slices of the scene (and over the 12-entry stack: 150 x the scalar redos).  The leaf phase puts the
                        // infinity back for its exact test (a genuine |1/d| of 1e30 never gets here: see `ordinary`).
                        const float inf = __uint_as_float(0x7F800000u);
                        if (fabsf(pr.iXY.x) == inf) pr.iXY.x = __builtin_copysignf(1e30f, pr.iXY.x);
                        if (fabsf(pr.iXY.y) == inf) pr.iXY.y = __builtin_copysignf(1e30f, pr.iXY.y);
                        if (fabsf(pr.iZ) == inf) pr.iZ = __builtin_copysignf(1e30f, pr.iZ);
                    }
                    // The root's own test (wgsl:379-382).  On the conservative records a class A ray (no infinite 1/d: no 0 * inf product anywhere) does without it: its first step
                    // tests the root's grandchildren -- supersets of boxes that lie inside the root's --, every leaf applies its exact box, and a ray that misses the root's box
                    // misses every box inside it (the slab arithmetic is monotone in the planes: rf_wide.hpp): one wasted step for such a ray, ~25 instructions less in every
                    // refill.  Class B rays keep the test: the reference's NaN rules at the ROOT's planes are not seen by any leaf.  The counting builds keep it too.
                    if (!(kConservative && !COUNT) || rayClass != kRayPlain)
                    {
                        float rootTMin;
                        rootOk = slabBounds(ray, wide.rootLo, wide.rootHi, rootTMin) && rootTMin < rayTMax;
                    }
                }
                negMask = ray.negX | (ray.negY << 1) | (ray.negZ << 2) | (rayClass == kRayHasInf ? 8u : 0u) | (triedCell ? 16u : 0u);
                if constexpr (kConservative)
                {
                    hbx = -(o.x * pr.iXY.x);
                    hby = -(o.y * pr.iXY.y);
                    hbz = -(o.z * pr.iZ);
                    hrot = ray.negX << 4, hrotY = ray.negY << 4, hrotZ = ray.negZ << 4;
                    lselX = ray.negX ? 0x00040005u : 0x00050004u, lselY = ray.negY ? 0x00040005u : 0x00050004u, lselZ = ray.negZ ? 0x00040005u : 0x00050004u;
                    const uint32_t signXY = ray.negX | (ray.negY << 1);
                    octKey = ray.negZ ? ((16u * (3u - signXY)) | (0x7777u << 8)) : 16u * signXY;
                }
                node = (needScalar || !rootOk) ? kNodeDone : (wide.rootLeaf != kWideNone ? wide.rootLeaf : 0u);
                if constexpr (kOccluderCache)
                {
                    uint32_t hint = 0u;
                    uint32_t later[kOccSlots > 1 ? kOccSlots - 1 : 1] = {};
                    if (occluderCache && (flags & kFlagOccluderNoTry) == 0u)
                    {
                        uint32_t e[kOccSlots];
                        loadOccluderCell(wide.occGrid + kOccSlots * static_cast<size_t>(occluderCell(o.x, o.y, o.z)), e);
                        if (e[0] != 0u)
                        {
                            hint = e[0];
#pragma unroll
                            for (int k = 1; k < kOccSlots; ++k) later[k - 1] = e[k];
                        }
                    }
                    if (occluderCache && hint != 0u && node == 0u)
                    {
                        push(0u, 0.0f); // the root waits (an empty stack: always room for it and the cell's entries)
#pragma unroll
                        for (int k = kOccSlots - 1; k >= 1; --k)
                            if (later[k - 1] != 0u) push(later[k - 1], 0.0f);
                        node = hint;
                        negMask |= kNegTriedHint;
#if defined(RF_EXP_PHASE)
                        ++phaseOccTried, phaseFromCache = true;
#endif
                    }
                }
            }
        }
        if (__ballot(node != kNodeIdle) == 0ull)
        {
            if (exhausted) break;
            continue;
        }

        // ---- descend: one 64-byte record = both children of an accepted interior node
        // (kEagerLeaves: when fewer than `leafVote` lanes would descend and kLeafRepeat or more already stand at a leaf -- off their stacks, or fresh from a refill of a
        // one-leaf tree -- the leaf phase comes first: the thin descend trip that used to run in front of it is skipped)
        if (!kEagerLeaves || !(__popcll(__ballot(static_cast<int32_t>(node) >= 0)) < leafVote &&
                               __popcll(__ballot(node - kWideLeafBit < kNodeDone - kWideLeafBit)) >= kLeafRepeat))
        do
        {
            if (kPhase) ++wDescend;
#if defined(RF_EXP_PHASE)
            if (node >= kNodeDone) ++phaseIdle;                       // no ray (finished, or never filled)
            else if (static_cast<int32_t>(node) < 0) ++phaseParked;   // stands at a leaf
#endif
            if (static_cast<int32_t>(node) >= 0)
            {
                if (kPhase) ++recordFetches;
                if constexpr (COMPACT == 6)
                {
                    // ---- oct records (rf_wide.hpp, WideBuild::oct): the boxes of the node's (up to) eight GREAT-GRANDCHILDREN as 8-bit planes on the record's own
                    // grid -- three levels of the reference's tree per dependent fetch, seven loads from one 128-byte line.  CONSERVATIVE tests (the leaf phase
                    // applies the exact boxes).  No ordering network: the record tabulates the position at which each slot is visited for the ray's sign pattern;
                    // the slots that can still be hit go onto the stack AT THEIR PLACE in that order (a slot's place = the number of hit slots visited after it:
                    // one popcount of the hit mask in visit order), and the first one comes straight back off the top.
                    const uint4* n = wide.oct + 8 * static_cast<size_t>(node);
                    const uint4  v0 = n[0], v1 = n[1], vx = n[2], vy = n[3], vz = n[4], wa = n[5], wb = n[6];
                    const float  ax = __uint_as_float(v0.w) * pr.iXY.x, ay = __uint_as_float(v1.x) * pr.iXY.y, az = __uint_as_float(v1.y) * pr.iZ;
                    const float  bx = __builtin_fmaf(-1024.0f, ax, (__uint_as_float(v0.x) - pr.oXY.x) * pr.iXY.x), by = __builtin_fmaf(-1024.0f, ay, (__uint_as_float(v0.y) - pr.oXY.y) * pr.iXY.y),
                                bz = __builtin_fmaf(-1024.0f, az, (__uint_as_float(v0.z) - pr.oZ) * pr.iZ);
                    float tq[8], fq[8];
                    localEntryBounds<0>(vx.x, vy.x, vz.x, lselX, lselY, lselZ, ax, ay, az, bx, by, bz, tq[0], fq[0]);
                    localEntryBounds<1>(vx.x, vy.x, vz.x, lselX, lselY, lselZ, ax, ay, az, bx, by, bz, tq[1], fq[1]);
                    localEntryBounds<0>(vx.y, vy.y, vz.y, lselX, lselY, lselZ, ax, ay, az, bx, by, bz, tq[2], fq[2]);
                    localEntryBounds<1>(vx.y, vy.y, vz.y, lselX, lselY, lselZ, ax, ay, az, bx, by, bz, tq[3], fq[3]);
                    localEntryBounds<0>(vx.z, vy.z, vz.z, lselX, lselY, lselZ, ax, ay, az, bx, by, bz, tq[4], fq[4]);
                    localEntryBounds<1>(vx.z, vy.z, vz.z, lselX, lselY, lselZ, ax, ay, az, bx, by, bz, tq[5], fq[5]);
                    localEntryBounds<0>(vx.w, vy.w, vz.w, lselX, lselY, lselZ, ax, ay, az, bx, by, bz, tq[6], fq[6]);
                    localEntryBounds<1>(vx.w, vy.w, vz.w, lselX, lselY, lselZ, ax, ay, az, bx, by, bz, tq[7], fq[7]);
                    const uint32_t words[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
                    // visit positions of the eight slots for this ray's direction signs: four nibbles, slot (c, g, 0) at nibble 2 c + g, slot (c, g, 1) at that ^ 1
                    const unsigned long long table = (static_cast<unsigned long long>(v1.w) << 32) | v1.z;
                    const uint32_t           ord = static_cast<uint32_t>(table >> (octKey & 63u)) ^ (octKey >> 8);
                    // slot e can still be hit  <=>  near <= far && far > 0 && near < rayTMax  <=>  max(near, tiny) <= min(far, pred(rayTMax)): one subtraction whose SIGN
                    // is the answer (x - y of two different floats is never zero, denormals are kept), shifted straight into the miss mask at the slot's position
                    const float tiny = __uint_as_float(1u), predTMax = __uint_as_float(__float_as_uint(rayTMax) - 1u); // (rayTMax > 1e-5: a positive normal number)
                    uint32_t    pos[8], miss = 0u;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                    {
                        pos[2 * j] = (ord >> (4 * j)) & 7u;
                        pos[2 * j + 1] = pos[2 * j] ^ 1u;
                    }
                    float gap[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                    {
                        gap[e] = isaMin(fq[e], predTMax) - isaMax(tq[e], tiny);
                        miss |= (__float_as_uint(gap[e]) >> 31) << pos[e];
                    }
                    const uint32_t hits = ~miss & 0xFFu; // bit p: the slot visited p-th can still be hit
                    if (hits != 0u)
                    {
                        const int need = __popc(hits);
                        bool      room = true;
                        if constexpr (kPtrStack)
                        {
                            while (room && stackSize + need * kSpStep > kSpLimit + spBase) room = (stackSize - spBase) >= kEvict * kSpStep && evict();
                        }
                        else
                        {
                            while (room && stackSize + need > kSpLimit) room = stackSize >= kEvict && evict();
                        }
                        if (__builtin_expect(room, 1))
                        {
                            const uint32_t later = hits >> 1;
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                if (!(__float_as_uint(gap[e]) >> 31))
                                {
                                    const int rank = __popc(later >> pos[e]); // hit slots visited after this one: they lie below it
                                    if constexpr (kStackWordsOnly) stackAt(stackSize + rank * kSpStep) = words[e];
                                    else stackAt(stackSize + rank * kSpStep) = make_uint2(words[e], __float_as_uint(tq[e]));
                                }
                            stackSize += need * kSpStep;
                            popNext();
                        }
                        else
                        {
                            needScalar = true;
                            node = kNodeDone;
                        }
                    }
                    else popNext();
                }
                else if constexpr (COMPACT == 3 || COMPACT == 4 || COMPACT == 5)
                {
                    // ---- quad records (rf_wide.hpp): the boxes of the node's (up to) four grandchildren in ONE 128-byte record --
                    // two levels of the reference's tree per dependent fetch.  Entries 0,1 belong to the first child, 2,3 to the
                    // second; an entry passes iff P(entry) && tmin(entry) < rayTMax, which implies the same for the skipped child.
                    float    tq0, tq1, tq2, tq3;
                    bool     okq0, okq1, okq2, okq3, hasNaN = false;
                    uint32_t w0, w1, w2, w3;
                    if constexpr (COMPACT == 3)
                    {
                        const auto quadStep = [&](float4 a0, float4 a1, float4 a2, float4 a3, float4 a4, float4 a5) {
                            float f0, f1, f2, f3;
                            slabPairBounds(pr, a0, a1, a2, tq0, f0, tq1, f1);
                            slabPairBounds(pr, a3, a4, a5, tq2, f2, tq3, f3);
                            asm volatile("" : "+v"(tq0), "+v"(f0), "+v"(tq1), "+v"(f1), "+v"(tq2), "+v"(f2), "+v"(tq3), "+v"(f3)); // (min/max chains stay with their products: see slabStep)
                            if (__builtin_expect((negMask & 8u) != 0u, 0)) hasNaN = slabPairHasNaN(pr, a0, a1, a2) || slabPairHasNaN(pr, a3, a4, a5);
                            okq0 = tq0 <= f0 && f0 > 0.0f;
                            okq1 = tq1 <= f1 && f1 > 0.0f;
                            okq2 = tq2 <= f2 && f2 > 0.0f;
                            okq3 = tq3 <= f3 && f3 > 0.0f;
                        };
                        const uint32_t uNode = __builtin_amdgcn_readfirstlane(node);
                        if (uniformFetch && __ballot(node != uNode) == 0ull)
                        {
                            typedef uint32_t u16v __attribute__((ext_vector_type(16)));
                            typedef uint32_t u8v __attribute__((ext_vector_type(8)));
                            typedef uint32_t u4v __attribute__((ext_vector_type(4)));
                            const float4* un = wide.quad + 8 * static_cast<size_t>(uNode);
                            u16v          a;
                            u8v           b;
                            u4v           c;
                            asm volatile("s_load_dwordx16 %0, %3, 0x0\n\ts_load_dwordx8 %1, %3, 0x40\n\ts_load_dwordx4 %2, %3, 0x60\n\ts_waitcnt lgkmcnt(0)"
                                         : "=&s"(a), "=&s"(b), "=&s"(c)
                                         : "s"(un)
                                         : "memory");
                            const auto f4 = [](uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return make_float4(__uint_as_float(x), __uint_as_float(y), __uint_as_float(z), __uint_as_float(w)); };
                            quadStep(f4(a.s0, a.s1, a.s2, a.s3), f4(a.s4, a.s5, a.s6, a.s7), f4(a.s8, a.s9, a.sa, a.sb), f4(a.sc, a.sd, a.se, a.sf), f4(b.s0, b.s1, b.s2, b.s3),
                                     f4(b.s4, b.s5, b.s6, b.s7));
                            w0 = c.x, w1 = c.y, w2 = c.z, w3 = c.w;
                        }
                        else
                        {
                            const float4* n = wide.quad + 8 * static_cast<size_t>(node);
                            const float4  v0 = n[0], v1 = n[1], v2 = n[2], v3 = n[3], v4 = n[4], v5 = n[5], v6 = n[6];
                            w0 = __float_as_uint(v6.x), w1 = __float_as_uint(v6.y), w2 = __float_as_uint(v6.z), w3 = __float_as_uint(v6.w);
                            quadStep(v0, v1, v2, v3, v4, v5);
                        }
                    }
                    else if constexpr (COMPACT == 5)
                    {
                        // ---- local-grid quad records (rf_wide.hpp, WideBuild::quadLocal): 8-bit planes on the record's own power-of-two grid,
                        // 64 bytes -- four loads.  CONSERVATIVE tests, as with the half-precision records; the leaf phase applies the exact boxes.
                        float          f0, f1, f2, f3;
                        const uint32_t uNode = __builtin_amdgcn_readfirstlane(node);
                        uint4          v0, v1, v2, v3;
                        if (uniformFetch && __ballot(node != uNode) == 0ull)
                        {
                            typedef uint32_t u16v __attribute__((ext_vector_type(16)));
                            const uint4*     un = wide.quadLocal + 4 * static_cast<size_t>(uNode);
                            u16v             a;
                            asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a) : "s"(un) : "memory");
                            v0 = make_uint4(a.s0, a.s1, a.s2, a.s3), v1 = make_uint4(a.s4, a.s5, a.s6, a.s7), v2 = make_uint4(a.s8, a.s9, a.sa, a.sb), v3 = make_uint4(a.sc, a.sd, a.se, a.sf);
                        }
                        else
                        {
                            const uint4* n = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(wide.quadLocal) + (node << 6)); // (32-bit byte offset: see the half-precision records)
                            v0 = n[0], v1 = n[1], v2 = n[2], v3 = n[3];
                        }
                        // A = scale / d (exact: a power of two times 1/d), B = (anchor - o) / d - 1024 A (one FMA)
                        const float ax = __uint_as_float(v0.w) * pr.iXY.x, ay = __uint_as_float(v1.x) * pr.iXY.y, az = __uint_as_float(v1.y) * pr.iZ;
                        const float bx = __builtin_fmaf(-1024.0f, ax, (__uint_as_float(v0.x) - pr.oXY.x) * pr.iXY.x), by = __builtin_fmaf(-1024.0f, ay, (__uint_as_float(v0.y) - pr.oXY.y) * pr.iXY.y),
                                    bz = __builtin_fmaf(-1024.0f, az, (__uint_as_float(v0.z) - pr.oZ) * pr.iZ);
                        localEntryBounds<0>(v1.z, v2.x, v2.z, lselX, lselY, lselZ, ax, ay, az, bx, by, bz, tq0, f0);
                        localEntryBounds<1>(v1.z, v2.x, v2.z, lselX, lselY, lselZ, ax, ay, az, bx, by, bz, tq1, f1);
                        localEntryBounds<0>(v1.w, v2.y, v2.w, lselX, lselY, lselZ, ax, ay, az, bx, by, bz, tq2, f2);
                        localEntryBounds<1>(v1.w, v2.y, v2.w, lselX, lselY, lselZ, ax, ay, az, bx, by, bz, tq3, f3);
                        w0 = v3.x, w1 = v3.y, w2 = v3.z, w3 = v3.w;
                        okq0 = tq0 <= f0 && f0 > 0.0f;
                        okq1 = tq1 <= f1 && f1 > 0.0f;
                        okq2 = tq2 <= f2 && f2 > 0.0f;
                        okq3 = tq3 <= f3 && f3 > 0.0f;
                    }
                    else
                    {
                        // ---- half-precision quad records (rf_wide.hpp, WideBuild::quadHalf): the same four entries, planes as binary16,
                        // 64 bytes -- four loads.  CONSERVATIVE tests (a superset passes; the leaf phase applies the exact boxes).
                        const float    bx = hbx, by = hby, bz = hbz;
                        const uint32_t rx = hrot, ry = hrotY, rz = hrotZ;
                        float       f0, f1, f2, f3;
                        const uint32_t uNode = __builtin_amdgcn_readfirstlane(node);
                        if (uniformFetch && __ballot(node != uNode) == 0ull)
                        {
                            typedef uint32_t u16v __attribute__((ext_vector_type(16)));
                            const uint4*     un = wide.quadHalf + 4 * static_cast<size_t>(uNode);
                            u16v             a;
                            asm volatile("s_load_dwordx16 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a) : "s"(un) : "memory");
                            halfEntryBounds<true>(a.s0, a.s1, a.s2, rx, ry, rz, pr.iXY.x, pr.iXY.y, pr.iZ, bx, by, bz, tq0, f0);
                            halfEntryBounds<true>(a.s3, a.s4, a.s5, rx, ry, rz, pr.iXY.x, pr.iXY.y, pr.iZ, bx, by, bz, tq1, f1);
                            halfEntryBounds<true>(a.s6, a.s7, a.s8, rx, ry, rz, pr.iXY.x, pr.iXY.y, pr.iZ, bx, by, bz, tq2, f2);
                            halfEntryBounds<true>(a.s9, a.sa, a.sb, rx, ry, rz, pr.iXY.x, pr.iXY.y, pr.iZ, bx, by, bz, tq3, f3);
                            // (the four words reach the lanes HERE: left to the compiler, the SGPR -> VGPR copies sit in the join block and the per-lane
                            // path pays for them on every step too: closest-hit launches -1 %)
                            asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7" : "=v"(w0), "=v"(w1), "=v"(w2), "=v"(w3) : "s"(a.sc), "s"(a.sd), "s"(a.se), "s"(a.sf));
                        }
                        else
                        {
                            // (a 32-bit BYTE offset -- record indices stay below 2^26 (kWideIndexBits), 64 bytes each -- so that the four loads take the array's base from SGPRs and
                            // the offset from one VGPR: one shift instead of a 64-bit shift and add per step)
                            const uint4* n = reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(wide.quadHalf) + (node << 6));
                            const uint4  v0 = n[0], v1 = n[1], v2 = n[2], v3 = n[3];
                            halfEntryBounds<false>(v0.x, v0.y, v0.z, rx, ry, rz, pr.iXY.x, pr.iXY.y, pr.iZ, bx, by, bz, tq0, f0);
                            halfEntryBounds<false>(v0.w, v1.x, v1.y, rx, ry, rz, pr.iXY.x, pr.iXY.y, pr.iZ, bx, by, bz, tq1, f1);
                            halfEntryBounds<false>(v1.z, v1.w, v2.x, rx, ry, rz, pr.iXY.x, pr.iXY.y, pr.iZ, bx, by, bz, tq2, f2);
                            halfEntryBounds<false>(v2.y, v2.z, v2.w, rx, ry, rz, pr.iXY.x, pr.iXY.y, pr.iZ, bx, by, bz, tq3, f3);
                            w0 = v3.x, w1 = v3.y, w2 = v3.z, w3 = v3.w;
                        }
                        okq0 = tq0 <= f0 && f0 > 0.0f;
                        okq1 = tq1 <= f1 && f1 > 0.0f;
                        okq2 = tq2 <= f2 && f2 > 0.0f;
                        okq3 = tq3 <= f3 && f3 > 0.0f;
                    }
                    if (__builtin_expect(hasNaN, 0))
                    {
                        // class B ray: a 0 * inf product means the packed test is not the reference's here
                        needScalar = true;
                        okq0 = okq1 = okq2 = okq3 = false;
                        stackSize = spBase, negMask &= 0xFFu; // -> popNext() ends the ray; it is redone below
                    }
                    const uint32_t axN = (w0 >> kWideAxisShift) & 3u, axA = (w1 >> kWideAxisShift) & 3u, axB = (w3 >> kWideAxisShift) & 3u;
                    // (an any-hit ray on the conservative layouts leaves `tmin < rayTMax` to the exact leaf test: its rayTMax is the constant
                    // tMax of the launch, which no box of a real scene lies beyond, and a superset is all these steps have to accept)
                    constexpr bool kSkipTMax = ANY_HIT && (COMPACT == 4 || COMPACT == 5);
                    // (an empty slot of a half-precision record holds an inverted box that fails `near <= far` for every ray: rf_wide.hpp, kHalfEmptyPlanes; the other
                    // layouts' empty slots hold a degenerate box and are recognised by their word)
                    constexpr bool kEmptyByBox = COMPACT == 4;
                    const bool     h0 = okq0 && (kSkipTMax || tq0 < rayTMax), h1 = okq1 && (kSkipTMax || tq1 < rayTMax) && (kEmptyByBox || w1 != kQuadEmpty),
                                   h2 = okq2 && (kSkipTMax || tq2 < rayTMax), h3 = okq3 && (kSkipTMax || tq3 < rayTMax) && (kEmptyByBox || w3 != kQuadEmpty);
                    constexpr uint32_t kAxisMask = ~(3u << kWideAxisShift);
                    // an entry that cannot be hit any more carries kQuadEmpty from here on
                    const uint32_t e0 = h0 ? (w0 & kAxisMask) : kQuadEmpty, e1 = h1 ? (w1 & kAxisMask) : kQuadEmpty, e2 = h2 ? w2 : kQuadEmpty, e3 = h3 ? (w3 & kAxisMask) : kQuadEmpty;
                    // visit order.  Closest hit: the reference's -- inside each child by dirNeg[the child's split axis], the two children by
                    // dirNeg[the node's] (wgsl:409-417 applied at both levels).  Any hit: nearer slab entry first at both levels (the
                    // visibility bit does not depend on the order: see NEAREST_FIRST above).
                    bool swapA, swapB, swapN;
                    if (NEAREST_FIRST)
                    {
                        const float inf = __uint_as_float(0x7F800000u);
                        const float k0 = h0 ? tq0 : inf, k1 = h1 ? tq1 : inf, k2 = h2 ? tq2 : inf, k3 = h3 ? tq3 : inf;
                        swapA = k1 < k0, swapB = k3 < k2;
                        swapN = __builtin_fminf(k2, k3) < __builtin_fminf(k0, k1);
                    }
                    else
                    {
                        // (an any-hit ray that is not asked for nearest-first visits the entries in RECORD order: its answer does not depend on the
                        // order, and on the VALU-bound 64-byte layouts the step without the ordering network -- 17 instructions -- beats the
                        // shorter walks of any ordering: shadow launches -8 %)
                        if (ANY_HIT) swapA = swapB = swapN = false;
                        else swapA = ((negMask >> axA) & 1u) != 0u, swapB = ((negMask >> axB) & 1u) != 0u, swapN = ((negMask >> axN) & 1u) != 0u;
                    }
                    const uint32_t a0w = swapA ? e1 : e0, a1w = swapA ? e0 : e1, b0w = swapB ? e3 : e2, b1w = swapB ? e2 : e3;
                    const float    a0t = swapA ? tq1 : tq0, a1t = swapA ? tq0 : tq1, b0t = swapB ? tq3 : tq2, b1t = swapB ? tq2 : tq3;
                    const uint32_t s0w = swapN ? b0w : a0w, s1w = swapN ? b1w : a1w, s2w = swapN ? a0w : b0w, s3w = swapN ? a1w : b1w;
                    const float    s1t = swapN ? b1t : a1t, s2t = swapN ? a0t : b0t, s3t = swapN ? a1t : b1t;
                    const bool     x0 = s0w != kQuadEmpty, x1 = s1w != kQuadEmpty, x2 = s2w != kQuadEmpty, x3 = s3w != kQuadEmpty;
                    if (x0 || x1 || x2 || x3)
                    {
                        // enter the first entry that can be hit; the later ones wait on the stack with their tmin, last first
                        bool pushed = true;
                        if constexpr (kPtrStack || kSpill)
                        {
                            // one bound check per step: room for the three entries a step can leave behind (a stack this full that does not
                            // need all three evicts its oldest entries a little earlier than necessary: same entries, same order)
                            pushed = stackSize < kSpLimit - 2 * kSpStep;
                            if (__builtin_expect(!pushed, 0)) pushed = evict();
                            if (pushed)
                            {
                                if (x3 && (x0 || x1 || x2)) pushUnchecked(s3w, s3t);
                                if (x2 && (x0 || x1)) pushUnchecked(s2w, s2t);
                                if (x1 && x0) pushUnchecked(s1w, s1t);
                            }
                        }
                        else
                        {
                            if (x3 && (x0 || x1 || x2)) pushed = push(s3w, s3t);
                            if (x2 && (x0 || x1)) pushed = push(s2w, s2t) && pushed;
                            if (x1 && x0) pushed = push(s1w, s1t) && pushed;
                        }
                        node = x0 ? s0w : (x1 ? s1w : (x2 ? s2w : s3w));
                        if (!pushed)
                        {
                            needScalar = true;
                            node = kNodeDone;
                        }
                    }
                    else popNext();
                }
                else
                {
                uint2 words;
                float t0, t1;
                bool  ok0, ok1, hasNaN = false;
#if defined(RF_ABLATE)
                float4 q0, q1, q2;
#endif
                // both boxes of the record against the lane's ray; class B rays (0 * inf possible) also check that the packed
                // test is the reference's here (rf_wide.hpp)
                const auto slabStep = [&](float4 a0, float4 a1, float4 a2) {
                    float far0, far1;
                    slabPairBounds(pr, a0, a1, a2, t0, far0, t1, far1);
                    // (the four results are pinned here so that the min/max chains stay in the basic block of their products:
                    // behind the rare branch below, the compiler no longer knows the products to be canonical and spends twelve
                    // v_max x,x on quieting them)
                    asm volatile("" : "+v"(t0), "+v"(far0), "+v"(t1), "+v"(far1));
                    if (__builtin_expect((negMask & 8u) != 0u, 0)) hasNaN = slabPairHasNaN(pr, a0, a1, a2);
                    ok0 = t0 <= far0 && far0 > 0.0f;
                    ok1 = t1 <= far1 && far1 > 0.0f;
#if defined(RF_ABLATE)
                    q0 = a0, q1 = a1, q2 = a2;
#endif
                };
                float c0LoX = 0.0f, c0HiX = 0.0f, c1LoX = 0.0f, c1HiX = 0.0f; // COMPACT: the children's x-plane t-values
                BoxT  c0b{}, c1b{};                                             // COMPACT == 2: all six
                if constexpr (COMPACT == 2)
                {
                    // 32-byte records: two dwordx4 per step; the node's own box (second array) only for lanes that arrive from the stack
                    const auto hotStep = [&](float4 h0, float4 h1) {
                        words = make_uint2(__float_as_uint(h1.z), __float_as_uint(h1.w));
                        float far0, far1;
                        slabPairHotBounds(pr, h0, h1.x, h1.y, words.x, words.y, own, t0, far0, t1, far1, c0b, c1b);
                        asm volatile("" : "+v"(t0), "+v"(far0), "+v"(t1), "+v"(far1)); // (min/max chains stay with their products: see slabStep)
                        if (__builtin_expect((negMask & 8u) != 0u, 0)) hasNaN = boxPairHasNaN(c0b, c1b);
                        ok0 = t0 <= far0 && far0 > 0.0f;
                        ok1 = t1 <= far1 && far1 > 0.0f;
                        words.x &= ~(3u << 24);
                        words.y &= ~((3u << 24) | (3u << kWideAxisShift));
                    };
                    const uint32_t uNode = __builtin_amdgcn_readfirstlane(node);
                    if (uniformFetch && __ballot(node != uNode) == 0ull)
                    {
                        typedef uint32_t u8v __attribute__((ext_vector_type(8)));
                        const float4* un = wide.hot + 2 * static_cast<size_t>(uNode);
                        const float4* uo = wide.own + 2 * static_cast<size_t>(uNode);
                        u8v           a, b;
                        asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dwordx8 %1, %3, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(a), "=&s"(b) : "s"(un), "s"(uo) : "memory");
                        if (!haveOuter)
                            own = boxPlaneT(pr, make_float4(__uint_as_float(b.s0), __uint_as_float(b.s1), __uint_as_float(b.s2), __uint_as_float(b.s3)), __uint_as_float(b.s4),
                                            __uint_as_float(b.s5));
                        hotStep(make_float4(__uint_as_float(a.s0), __uint_as_float(a.s1), __uint_as_float(a.s2), __uint_as_float(a.s3)),
                                make_float4(__uint_as_float(a.s4), __uint_as_float(a.s5), __uint_as_float(a.s6), __uint_as_float(a.s7)));
                    }
                    else
                    {
                        const float4* n = wide.hot + 2 * static_cast<size_t>(node);
                        const float4  v0 = n[0], v1 = n[1];
                        if (!haveOuter)
                        {
                            const float4* o = wide.own + 2 * static_cast<size_t>(node);
                            const float4  o0 = o[0];
                            const uint2*  zPtr = reinterpret_cast<const uint2*>(o + 1);
                            asm volatile("" : "+v"(zPtr)); // (an 8-byte global load, not widened: see the words load of the plain layout below)
                            typedef const unsigned long long __attribute__((address_space(1)))* GlobalWordPtr;
                            const unsigned long long both = *(GlobalWordPtr)(zPtr);
                            own = boxPlaneT(pr, o0, __uint_as_float(static_cast<uint32_t>(both)), __uint_as_float(static_cast<uint32_t>(both >> 32)));
                        }
                        hotStep(v0, v1);
                    }
                }
                else if constexpr (COMPACT == 1)
                {
                    // Compact-capable records: three dwordx4 per step; the fourth piece (the node's own x planes) only for lanes that
                    // do not carry them -- 11 % of the steps (after a pop, at the root).
                    const auto compactStep = [&](float4 a0, float4 a1, float4 a2) {
                        words = make_uint2(__float_as_uint(a2.x), __float_as_uint(a2.z));
                        float far0, far1;
                        slabPairCompactBounds(pr, a0, a1, a2, tOuterLo, tOuterHi, words.y, t0, far0, t1, far1, c0LoX, c0HiX, c1LoX, c1HiX);
                        words.y &= ~(3u << kWideAxisShift);
                        asm volatile("" : "+v"(t0), "+v"(far0), "+v"(t1), "+v"(far1)); // (min/max chains stay with their products: see slabStep)
                        if (__builtin_expect((negMask & 8u) != 0u, 0)) hasNaN = slabPairCompactHasNaN(pr, a0, a1, a2, c0LoX, c0HiX, c1LoX, c1HiX);
                        ok0 = t0 <= far0 && far0 > 0.0f;
                        ok1 = t1 <= far1 && far1 > 0.0f;
                    };
                    const uint32_t uNode = __builtin_amdgcn_readfirstlane(node);
                    if (uniformFetch && __ballot(node != uNode) == 0ull)
                    {
                        typedef uint32_t u8v __attribute__((ext_vector_type(8)));
                        typedef uint32_t u4v __attribute__((ext_vector_type(4)));
                        typedef uint32_t u2v __attribute__((ext_vector_type(2)));
                        // a wave-uniform step costs no vector-L1 access whatever the layout: it reads the PLAIN record through the scalar
                        // cache and pays nothing for the selects -- the children's x-plane t-values are four of its twelve products
                        const float4* un = wide.nodes + 4 * static_cast<size_t>(uNode);
                        u8v           a;
                        u4v           b;
                        u2v           c;
                        asm volatile("s_load_dwordx8 %0, %3, 0x0\n\ts_load_dwordx4 %1, %3, 0x20\n\ts_load_dwordx2 %2, %3, 0x30\n\ts_waitcnt lgkmcnt(0)"
                                     : "=&s"(a), "=&s"(b), "=&s"(c)
                                     : "s"(un)
                                     : "memory");
                        const float4 a0 = make_float4(__uint_as_float(a.s0), __uint_as_float(a.s1), __uint_as_float(a.s2), __uint_as_float(a.s3)),
                                     a1 = make_float4(__uint_as_float(a.s4), __uint_as_float(a.s5), __uint_as_float(a.s6), __uint_as_float(a.s7)),
                                     a2 = make_float4(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w));
                        float        far0, far1;
                        slabPairBoundsX(pr, a0, a1, a2, t0, far0, t1, far1, c0LoX, c0HiX, c1LoX, c1HiX);
                        asm volatile("" : "+v"(t0), "+v"(far0), "+v"(t1), "+v"(far1)); // (min/max chains stay with their products: see slabStep)
                        if (__builtin_expect((negMask & 8u) != 0u, 0)) hasNaN = slabPairHasNaN(pr, a0, a1, a2);
                        ok0 = t0 <= far0 && far0 > 0.0f;
                        ok1 = t1 <= far1 && far1 > 0.0f;
                        words = make_uint2(c.x, c.y);
                    }
                    else
                    {
                        const float4* n = wide.compact + 4 * static_cast<size_t>(node);
                        const float4  v0 = n[0], v1 = n[1], v2 = n[2];
                        if (!haveOuter)
                        {
                            const uint2* outerPtr = reinterpret_cast<const uint2*>(n + 3);
                            asm volatile("" : "+v"(outerPtr)); // (see the words load of the plain layout below: an 8-byte global load, not widened)
                            typedef const unsigned long long __attribute__((address_space(1)))* GlobalWordPtr;
                            const unsigned long long both = *(GlobalWordPtr)(outerPtr);
                            tOuterLo = (__uint_as_float(static_cast<uint32_t>(both)) - pr.oXY.x) * pr.iXY.x;
                            tOuterHi = (__uint_as_float(static_cast<uint32_t>(both >> 32)) - pr.oXY.x) * pr.iXY.x;
                        }
                        compactStep(v0, v1, v2);
                    }
                }
                else
                {
                // With the pixel-major, direction-sorted slot order the 64 rays of a wave are one pixel's samples, and at
                // bounce 1 (and for the first steps of any freshly filled wave) every descending lane sits at the SAME
                // record.  Then the record comes through the scalar cache with three s_load instructions instead of
                // 4 x 64 per-lane vector loads of one line: no vector-L1 traffic at all for that step.  Same bytes, same
                // arithmetic -- only the path the record takes to the registers differs.  The slab arithmetic is issued
                // inside each branch, so that on this one its box operands stay in SGPRs (bounce 1 is VALU-issue bound:
                // copying the 14 dwords into VGPRs first cost 14 of the ~85 VALU instructions of a step).
                const uint32_t uNode = __builtin_amdgcn_readfirstlane(node);
                if (uniformFetch && __ballot(node != uNode) == 0ull)
                {
                    typedef uint32_t u8v __attribute__((ext_vector_type(8)));
                    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
                    typedef uint32_t u2v __attribute__((ext_vector_type(2)));
                    const float4* un = wide.nodes + 4 * static_cast<size_t>(uNode);
                    u8v           a;
                    u4v           b;
                    u2v           c;
                    asm volatile("s_load_dwordx8 %0, %3, 0x0\n\ts_load_dwordx4 %1, %3, 0x20\n\ts_load_dwordx2 %2, %3, 0x30\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&s"(a), "=&s"(b), "=&s"(c)
                                 : "s"(un)
                                 : "memory");
                    slabStep(make_float4(__uint_as_float(a.s0), __uint_as_float(a.s1), __uint_as_float(a.s2), __uint_as_float(a.s3)),
                             make_float4(__uint_as_float(a.s4), __uint_as_float(a.s5), __uint_as_float(a.s6), __uint_as_float(a.s7)),
                             make_float4(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w)));
                    words = make_uint2(c.x, c.y);
                }
                else
                {
                    const float4* n = wide.nodes + 4 * static_cast<size_t>(node);
                    // 56 of the record's 64 bytes: nothing is loaded that is not used
                    const float4 v0 = n[0], v1 = n[1], v2 = n[2];
                    // (the pointer goes through an empty asm so that the compiler forgets its 16-byte alignment and
                    // cannot widen the 8-byte load back to a dwordx4; it comes back as a GLOBAL pointer -- a generic one
                    // makes the load a flat_load, which also counts against lgkmcnt)
                    const uint2* wordPtr = reinterpret_cast<const uint2*>(n + 3);
                    asm volatile("" : "+v"(wordPtr));
                    typedef const unsigned long long __attribute__((address_space(1)))* GlobalWordPtr;
                    const unsigned long long both = *(GlobalWordPtr)(wordPtr);
                    words = make_uint2(static_cast<uint32_t>(both), static_cast<uint32_t>(both >> 32));
                    slabStep(v0, v1, v2);
                }
                }
                const uint32_t axis = (words.x >> kWideAxisShift) & 3u;
                const uint32_t word0 = words.x & ~(3u << kWideAxisShift), word1 = words.y;
                if (__builtin_expect(hasNaN, 0))
                {
                    // class B ray: a 0 * inf product means the packed test is not the reference's here
                    needScalar = true;
                    ok0 = ok1 = false;
                    stackSize = spBase, negMask &= 0xFFu; // -> popNext() ends the ray; it is redone below
                }
#if defined(RF_ABLATE) && RF_ABLATE == 1
                {   // ablation: the slab arithmetic twice more (result kept alive, never different)
                    float4 z0 = q0, z1 = q1, z2 = q2;
                    for (int rep = 0; rep < 2; ++rep)
                    {
                        asm volatile("" : "+v"(z0.x), "+v"(z0.y), "+v"(z0.z), "+v"(z0.w), "+v"(z1.x), "+v"(z1.y), "+v"(z1.z), "+v"(z1.w), "+v"(z2.x), "+v"(z2.y), "+v"(z2.z), "+v"(z2.w));
                        float a0, a1; bool b0, b1;
                        slabPair(pr, z0, z1, z2, b0, a0, b1, a1);
                        if (a0 != t0 || a1 != t1 || b0 != ok0 || b1 != ok1) t0 = __uint_as_float(0x7FC00000u);
                    }
                }
#elif defined(RF_ABLATE) && RF_ABLATE == 2
                {   // ablation: one more 64-byte record fetch per step, from an unrelated place
                    const uint32_t other = (node * 2654435761u) % wide.numRecords;
                    const float4*  m = wide.nodes + 4 * static_cast<size_t>(other);
                    const float4   y0 = m[0], y1 = m[1], y2 = m[2], y3 = m[3];
                    const float sum = ((y0.x + y0.y) + (y0.z + y0.w)) + ((y1.x + y1.y) + (y1.z + y1.w)) + ((y2.x + y2.y) + (y2.z + y2.w)) + ((y3.x + y3.y) + (y3.z + y3.w));
                    if (sum == 1.2345e-33f) t0 = __uint_as_float(0x7FC00000u);
                }
#endif
                // reference order: dirNeg[axis] ? second child first : first child first
                const bool neg = NEAREST_FIRST ? (t1 < t0) : (((negMask >> axis) & 1u) != 0u);
                if constexpr (kRefCount)
                {
                    const uint32_t nearWord = neg ? word1 : word0, farWord = neg ? word0 : word1;
                    const bool     okNear = neg ? ok1 : ok0, okFar = neg ? ok0 : ok1;
                    const float    tNear = neg ? t1 : t0, tFar = neg ? t0 : t1;
                    const bool     pushed = push(farWord, okFar ? tFar : __uint_as_float(0x7F800000u));
                    rayStackHigh = max(rayStackHigh, static_cast<uint32_t>(stackSize)); // (kRefCount => COUNT => plain depth)
                    ++rayNodes; // the near child
                    if (!pushed)
                    {
                        needScalar = true;
                        node = kNodeDone;
                    }
                    else if (okNear && tNear < rayTMax) node = nearWord;
                    else popNext();
                }
                else
                {
                    if (COUNT) rayNodes += 2; // this build counts box tests
                    // which child is entered first: the near one if both can still be hit, else the one that can
                    const bool     hit0 = ok0 && t0 < rayTMax, hit1 = ok1 && t1 < rayTMax;
                    const bool     both = hit0 && hit1;
                    const bool     second = both ? neg : hit1;
                    const uint32_t firstWord = second ? word1 : word0, otherWord = second ? word0 : word1;
                    const float    otherT = second ? t0 : t1;
                    if (hit0 || hit1)
                    {
                        node = firstWord;
                        if (COMPACT == 1)
                        {
                            // the child entered straight from this step: its own x-plane t-values travel with the lane
                            tOuterLo = second ? c1LoX : c0LoX;
                            tOuterHi = second ? c1HiX : c0HiX;
                            haveOuter = true;
                        }
                        if (COMPACT == 2)
                        {
                            // ... all six of them with the 32-byte records
                            own.loX = second ? c1b.loX : c0b.loX, own.loY = second ? c1b.loY : c0b.loY;
                            own.hiX = second ? c1b.hiX : c0b.hiX, own.hiY = second ? c1b.hiY : c0b.hiY;
                            own.loZ = second ? c1b.loZ : c0b.loZ, own.hiZ = second ? c1b.hiZ : c0b.hiZ;
                            haveOuter = true;
                        }
                        if (both && !push(otherWord, otherT))
                        {
                            needScalar = true;
                            node = kNodeDone;
                        }
                    }
                    else popNext();
                }
                }
            }
        } while (__popcll(__ballot(static_cast<int32_t>(node) >= 0)) >= leafVote);
        // (leaving the loop EARLIER -- as soon as 24 / 32 / 40 lanes are parked at a leaf, however many still descend -- measured +5.5 / +3 / +1.5 % on the closest-hit
        // launches: parking pays; profiles/r05_leafrep/ab_leafearly.log)

        // ---- leaves
        uint32_t occluderWord = 0u; // kOccluderCache: the leaf in which this lane has just found an occluder
        // kEagerLeaves: the leaf phase REPEATS while kLeafRepeat or more lanes stand at a leaf (see the declaration of kEagerLeaves)
        do
        {
        // ---- Leaf phase over dense (lane, triangle) pairs (round 5).  The loop further down tests triangle i of every parked lane's leaf in trip i: a phase lasts as
        // long as its LONGEST leaf, and on a scene whose leaves differ in length (the atrium with clutter: 1 ... 12 triangles, 7.2 tests per closest-hit ray) most trips
        // run for a handful of lanes.  When a parked lane's leaf holds kDenseMin triangles or more, the phase runs over PAIRS instead: the lanes' triangle counts are
        // prefix-summed, pair p = (owner lane, triangle p - offset[owner]) goes to lane p mod 64 of trip p / 64 (whole leaves per trip), which fetches the owner's ray
        // through ds_bpermute and tests that one triangle against the owner's rayTMax AT ENTRY; the owner then walks through the hits among its own pairs in triangle
        // order with the reference's `t < rayTMax` (wgsl:385-402).  Same result as the sequential walk: a triangle the walk accepts has t below the rayTMax of that
        // moment <= the entry value, so it is among the hits here; a hit here that the walk would reject (t >= the rayTMax an earlier triangle left) is rejected by the
        // owner's own walk over the hits, in the same order with the same comparison.  Any-hit: a leaf with a hit among its pairs stops the ray.
        // The block is self-contained (its own leaf decode and exact box test) so that the loop below keeps its registers to itself: what it needs of a leaf's
        // first triangle record is live only inside its own branch.  And it is a template parameter (DENSE_LEAVES): its mere presence costs the closest-hit launches
        // of a scene that never uses it 2.5 % (profiles/r05_leaf/ab_presence.log), so scenes without long leaves run the instantiations without it.
#if defined(RF_EXP_PHASE)
        if (__ballot(node - kWideLeafBit < kNodeDone - kWideLeafBit) != 0ull)
        {
            ++phaseLeafWave; // (every pass of a repeated leaf phase counts)
            if (node >= kNodeDone) ++phaseLeafIdle;
            else if (static_cast<int32_t>(node) >= 0) ++phaseLeafInterior;
        }
#endif
        bool denseDone = false; // this lane's leaf has been dealt with by this block
        if constexpr (!COUNT && DENSE_LEAVES)
        {
            const uint32_t     kDenseMin = (flags >> kFlagDenseLeafShift) & 15u;
            constexpr uint32_t kDenseMaxLeaf = 16u; // (longer leaves keep the loop below)
            const bool         atLeafD = node - kWideLeafBit < kNodeDone - kWideLeafBit;
            // (decided on the count field of the leaf word alone: 7 = a big leaf of 8 or more)
            if (kDenseMin != 0u && __ballot(atLeafD && ((node >> kWideIndexBits) & 7u) + 1u >= kDenseMin) != 0ull)
            {
                uint32_t firstD = 0u, cnt = 0u, hintD = 0u;
                bool     rejected = false;
                if (atLeafD)
                {
                    firstD = node & ((1u << kWideIndexBits) - 1u), cnt = ((node >> kWideIndexBits) & 7u) + 1u;
                    if (cnt == 8u)
                    {
                        const uint2 big = wide.bigLeaves[firstD];
                        firstD = big.x;
                        cnt = big.y;
                    }
                    if (cnt > kDenseMaxLeaf) cnt = 0u; // not taken here
                    else if (leafBoxAtLeaf)
                    {
                        // the leaf's exact box, as the loop below applies it (the reference's test at the leaf: same formula, the rayTMax of this moment)
                        const float* t0 = reinterpret_cast<const float*>(scene.triangles + kTriStride * static_cast<size_t>(firstD));
                        const float  loX = t0[3], loY = t0[7], loZ = t0[11];
                        const float4 hi = *reinterpret_cast<const float4*>(t0 + 12);
                        if constexpr (kOccluderCache) hintD = __float_as_uint(hi.w);
                        PackedRay exact = pr;
                        if (kConservative && __builtin_expect((negMask & 8u) != 0u, 0))
                        {
                            const float inf = __uint_as_float(0x7F800000u);
                            if (fabsf(exact.iXY.x) == 1e30f) exact.iXY.x = __builtin_copysignf(inf, exact.iXY.x);
                            if (fabsf(exact.iXY.y) == 1e30f) exact.iXY.y = __builtin_copysignf(inf, exact.iXY.y);
                            if (fabsf(exact.iZ) == 1e30f) exact.iZ = __builtin_copysignf(inf, exact.iZ);
                        }
                        float bn, bf;
                        bool  boxNaN;
                        slabSingleBounds(exact, loX, loY, loZ, hi.x, hi.y, hi.z, bn, bf, boxNaN);
                        if (__builtin_expect((negMask & 8u) != 0u && boxNaN, 0)) cnt = 0u; // (class B ray with a 0 * inf product: left to the loop below, which sends it to the scalar traversal)
                        else if (!(bn <= bf && bf > 0.0f && bn < rayTMax))
                        {
                            cnt = 0u; // the reference rejects this leaf: no triangle is tested
                            rejected = true;
                        }
                    }
                }
                bool dealt = rejected; // this lane's leaf is finished with (rejected by its box, or its pairs have been tested)
                bool stopped = false;                    // ANY_HIT: a pair of this lane's leaf was hit
                const uint32_t incl = waveScanInclusive<false>(cnt), off = incl - cnt;
                const uint32_t total = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 63));
                // (worth it when the pairs need fewer trips than the longest leaf has triangles: a pair trip costs about one and a half triangle trips)
                const uint32_t pairTrips = (total + 63u) / 64u;
                const bool     goDense = (pairTrips <= 1u) || (pairTrips <= 2u && __ballot(cnt >= 5u) != 0ull) || (pairTrips <= 4u && __ballot(cnt >= 9u) != 0ull);
                if (goDense)
                {
                    const float oX = pr.oXY.x, oY = pr.oXY.y, oZ = pr.oZ;
                    uint32_t    base = 0u;
                    while (base < total) // (wave-uniform)
                    {
                        // this trip: the leaves that start at or behind `base` and END within the next 64 pairs
                        const unsigned long long over = __ballot(cnt != 0u && off >= base && off + cnt > base + 64u);
                        const uint32_t           next = over != 0ull ? static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(off), __builtin_ctzll(over))) : total;
                        const bool               inTrip = cnt != 0u && off >= base && off < next;
                        const uint32_t           segLo = off - base; // (meaningful for inTrip lanes)
                        // owner of pair slot q: every leaf of the trip drops lane + 1 at the slot of its first pair (ds_permute_b32; the other lanes drop a 0 at a slot
                        // that starts no leaf -- the highest lane wins a slot, and only zeros compete there), then a running maximum fills the leaf's other slots
                        const uint32_t           pairs = next - base;
                        const unsigned long long longer = __ballot(inTrip && cnt >= 2u);
                        const uint32_t           dump = (pairs < 64u || longer == 0ull) ? (pairs & 63u) : static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(segLo), __builtin_ctzll(longer))) + 1u; // (64 one-triangle leaves: every lane sends)
                        const uint32_t           mark = static_cast<uint32_t>(__builtin_amdgcn_ds_permute(static_cast<int>((inTrip ? segLo : dump) << 2), static_cast<int>(inTrip ? lane + 1u : 0u)));
                        const uint32_t           owner = waveScanInclusive<true>(mark) - 1u;
                        const bool               pairLive = lane < pairs;
                        const uint32_t           src = pairLive ? owner : lane;
                        const uint32_t           triOfPair = laneGather(firstD - off, src) + base + lane;
                        const Vec3               po = vec3(laneGather(oX, src), laneGather(oY, src), laneGather(oZ, src));
                        const Vec3               pd = vec3(laneGather(rayDir.x, src), laneGather(rayDir.y, src), laneGather(rayDir.z, src));
                        const float              pTMax = ANY_HIT ? tMax : laneGather(rayTMax, src);
                        TriangleHit              th{};
                        bool                     pairHit = false;
                        if (pairLive)
                        {
                            const v3f a = *reinterpret_cast<const v3f*>(scene.triangles + kTriStride * static_cast<size_t>(triOfPair));
                            const v3f b = *reinterpret_cast<const v3f*>(scene.triangles + kTriStride * static_cast<size_t>(triOfPair) + 1);
                            const v3f c = *reinterpret_cast<const v3f*>(scene.triangles + kTriStride * static_cast<size_t>(triOfPair) + 2);
                            pairHit = intersectTriangle(po, pd, vec3(a.x, a.y, a.z), vec3(b.x, b.y, b.z), vec3(c.x, c.y, c.z), pTMax, th);
                        }
                        const unsigned long long hitMask = __ballot(pairHit);
                        uint32_t                 mine = inTrip ? static_cast<uint32_t>(hitMask >> segLo) & ((1u << cnt) - 1u) : 0u; // hits among this lane's own pairs, bit j = triangle first + j
                        if constexpr (ANY_HIT)
                        {
                            if (mine != 0u) stopped = true;
                        }
                        else
                        {
                            while (__ballot(mine != 0u) != 0ull) // (wave-uniform: the gathers below read other lanes' registers)
                            {
                                const uint32_t j = mine != 0u ? static_cast<uint32_t>(__builtin_ctz(mine)) : 0u;
                                const uint32_t from = mine != 0u ? segLo + j : lane;
                                const float    tj = laneGather(th.t, from), uj = laneGather(th.u, from), vj = laneGather(th.v, from);
                                if (mine != 0u && tj < rayTMax)
                                {
                                    rayTMax = tj;
                                    best.u = uj;
                                    best.v = vj;
                                    best.triangle = firstD + j;
                                }
                                mine &= mine - 1u;
                            }
                        }
                        if (inTrip) dealt = true;
                        base = next;
                    }
                }
                else dealt = false; // (not worth it: the loop below takes every leaf, the rejected ones included -- it repeats their box test)
                // what the loop below does with a leaf it has finished with
                if (dealt)
                {
                    if (ANY_HIT && stopped)
                    {
                        occluded = true;
                        if (kOccluderCache) occluderWord = hintD != 0u ? hintD : node;
                        node = kNodeDone;
                    }
                    else popNext();
                }
                denseDone = dealt;
            }
        }
        if (node - kWideLeafBit < kNodeDone - kWideLeafBit && !denseDone) // (a lane the dense phase has moved on may hold its NEXT leaf by now: that one waits for the next phase)
        {
            uint32_t first = node & ((1u << kWideIndexBits) - 1u), n = ((node >> kWideIndexBits) & 7u) + 1u;
            if (n == 8u)
            {
                const uint2 big = wide.bigLeaves[first];
                first = big.x;
                n = big.y;
            }
            bool finished = false;
            if (kPhase) ++wLeafPhase;
            float4 firstA{}, firstB{}, firstC{};
            uint32_t leafHint = 0u;
            if (leafBoxAtLeaf)
            {
                // The half-precision / local-grid quad records let a SUPERSET of the reference's nodes through; what the reference does at a leaf --
                // test its box, exactly, with its own formula, against the rayTMax of this moment -- happens here.  The leaf's box
                // rides in the spare floats of its first triangle record (leafBoxesIntoTriangles): the same 64-byte line.
                static_assert(kTriStride * sizeof(float4) == 64, "the leaf phase addresses triangle records by a 32-bit byte offset: 64 bytes each, indices below 2^26");
                const float4* t0 = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(scene.triangles) + (first << 6));
                float4 hi;
                {
                    // FOUR loads for the 64-byte record, each pinned in its own register tuple: left alone, the compiler re-cuts the record to suit the packed arithmetic
                    // below -- six loads (4 + 16, 8, 8, 16, 4, 16 bytes), six vector-L1 tag accesses per lane and leaf instead of four
                    typedef float v4f __attribute__((ext_vector_type(4)));
                    const v4f* t4 = reinterpret_cast<const v4f*>(t0);
                    // (ONE pin behind all four loads: the empty asm is a scheduling barrier that needs its operands, i.e. a wait for the loads in front of it)
                    v4f        ra = t4[0], rb = t4[1], rc = t4[2];
                    if constexpr (kOccluderCache)
                    {
                        v4f rd = t4[3]; // .w: what the occluder cache remembers for this leaf (leafBoxesIntoTriangles)
                        asm volatile("" : "+v"(ra), "+v"(rb), "+v"(rc), "+v"(rd));
                        hi = make_float4(rd.x, rd.y, rd.z, rd.w);
                    }
                    else
                    {
                        v3f h3 = *reinterpret_cast<const v3f*>(t0 + 3);
                        asm volatile("" : "+v"(ra), "+v"(rb), "+v"(rc), "+v"(h3));
                        hi = make_float4(h3.x, h3.y, h3.z, 0.0f);
                    }
                    firstA = make_float4(ra.x, ra.y, ra.z, ra.w), firstB = make_float4(rb.x, rb.y, rb.z, rb.w), firstC = make_float4(rc.x, rc.y, rc.z, rc.w);
                }
                if constexpr (kOccluderCache) leafHint = __float_as_uint(hi.w);
                float     bn, bf;
                bool      boxNaN;
                PackedRay exact = pr;
                if (kConservative && __builtin_expect((negMask & 8u) != 0u, 0))
                {
                    // class B: the infinite components of 1/d that the conservative tests replaced by +-1e30 (refill) are infinite again
                    const float inf = __uint_as_float(0x7F800000u);
                    if (fabsf(exact.iXY.x) == 1e30f) exact.iXY.x = __builtin_copysignf(inf, exact.iXY.x);
                    if (fabsf(exact.iXY.y) == 1e30f) exact.iXY.y = __builtin_copysignf(inf, exact.iXY.y);
                    if (fabsf(exact.iZ) == 1e30f) exact.iZ = __builtin_copysignf(inf, exact.iZ);
                }
                slabSingleBounds(exact, firstA.w, firstB.w, firstC.w, hi.x, hi.y, hi.z, bn, bf, boxNaN);
                if (__builtin_expect((negMask & 8u) != 0u && boxNaN, 0))
                {
                    // class B ray with a 0 * inf product at this box: the reference's NaN rules apply -- the whole ray is redone by
                    // the scalar traversal (as the exact-record kernels do for any step with such a product)
                    needScalar = true;
                    stackSize = spBase, negMask &= 0xFFu;
                    n = 0;
                }
                else if (!(bn <= bf && bf > 0.0f && bn < rayTMax)) n = 0; // the reference rejects this leaf: no triangle is tested
            }
            for (uint32_t i = 0; i < n; ++i)
            {
                if (kPhase) ++wLeaf;
                const uint32_t tri = first + i;
                Vec3           p0, p1, p2;
                // the same triangle in every lane of this leaf phase (one pixel's samples reaching the same leaf): scalar cache
                const uint32_t uTri = __builtin_amdgcn_readfirstlane(tri);
                if (leafBoxAtLeaf && i == 0u)
                {
                    p0 = vec3(firstA.x, firstA.y, firstA.z), p1 = vec3(firstB.x, firstB.y, firstB.z), p2 = vec3(firstC.x, firstC.y, firstC.z);
                }
                else if (uniformTri && __ballot(tri != uTri) == 0ull)
                {
                    typedef uint32_t u8v __attribute__((ext_vector_type(8)));
                    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
                    const float4* ut = scene.triangles + kTriStride * static_cast<size_t>(uTri);
                    u8v           ab;
                    u4v           cc;
                    asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dwordx4 %1, %2, 0x20\n\ts_waitcnt lgkmcnt(0)" : "=&s"(ab), "=&s"(cc) : "s"(ut) : "memory");
                    p0 = vec3(__uint_as_float(ab.s0), __uint_as_float(ab.s1), __uint_as_float(ab.s2));
                    p1 = vec3(__uint_as_float(ab.s4), __uint_as_float(ab.s5), __uint_as_float(ab.s6));
                    p2 = vec3(__uint_as_float(cc.x), __uint_as_float(cc.y), __uint_as_float(cc.z));
                }
                else
                {
                    const v3f a = *reinterpret_cast<const v3f*>(scene.triangles + kTriStride * tri);
                    const v3f b = *reinterpret_cast<const v3f*>(scene.triangles + kTriStride * tri + 1);
                    const v3f c = *reinterpret_cast<const v3f*>(scene.triangles + kTriStride * tri + 2);
                    p0 = vec3(a.x, a.y, a.z), p1 = vec3(b.x, b.y, b.z), p2 = vec3(c.x, c.y, c.z);
                }
                if (COUNT) ++rayTris;
#if defined(RF_EXP_PHASE)
                ++phaseTris;
#endif
                TriangleHit th;
                if (intersectTriangle(vec3(pr.oXY.x, pr.oXY.y, pr.oZ), rayDir, p0, p1, p2, rayTMax, th))
                {
                    if (ANY_HIT)
                    {
                        occluded = true;
                        finished = true;
                        break;
                    }
                    // the offset hit point (wgsl:511-519) is rebuilt from (triangle, u, v) by kShade
                    rayTMax = th.t;
                    best.u = th.u;
                    best.v = th.v;
                    best.triangle = tri;
                }
            }
            if (finished)
            {
                if (kOccluderCache) occluderWord = leafHint != 0u ? leafHint : node;
#if defined(RF_EXP_PHASE)
                if (ANY_HIT) { ++phaseOccluded; if (phaseFromCache && stackSize == spBase + kSpStep) ++phaseOccHit; }
#endif
                node = kNodeDone;
            }
            else popNext();
        }
        } while (kEagerLeaves && __popcll(__ballot(node - kWideLeafBit < kNodeDone - kWideLeafBit)) >= kLeafRepeat);

        // ---- write back finished rays
        if (kSpill && node == kNodeDone && !needScalar && !occluded)
        {
            while (node == kNodeDone && (negMask >> 8) != 0u) // evicted entries pending: not finished after all (rare: see evict())
            {
                unspill();
                popNext();
            }
        }
        if (node == kNodeDone)
        {
            if (needScalar)
            {
                // axis-parallel / denormal / non-finite rays (0 * inf slabs) and rays whose stack outgrew
                // LDS: the reference's own scalar traversal, whole ray at once
                TraversalCounters c2;
                atomicAdd(&counters->scalarRedo[ANY_HIT ? 1 : 0], 1ull);
                best.triangle = kMiss;
                occluded = traverse<ANY_HIT, COUNT, 0>(scene, vec3(pr.oXY.x, pr.oXY.y, pr.oZ), rayDir, tMax, nullptr, best, c2);
                if (c2.abandoned) atomicAdd(&counters->abandonedRays, 1ull);
                rayTMax = best.triangle != kMiss ? best.t : tMax;
                rayNodes = c2.nodesVisited;
                rayTris = c2.triangleTests;
                rayStackHigh = c2.stackHigh;
            }
            if (COUNT)
            {
                tc.nodesVisited += rayNodes;
                tc.triangleTests += rayTris;
                tc.stackHigh = max(tc.stackHigh, rayStackHigh);
            }
            if constexpr (kOccluderCache)
            {
                // kOccSlots entries per cell, most recent first: a new occluder goes to the front (the others move back, the last one drops out); a ray
                // that tried the cell's entries and reached the sun drops the first one
                if (occluderCache && (occluderWord != 0u || (!occluded && (negMask & kNegTriedHint) != 0u)))
                {
                    uint32_t* const cell = wide.occGrid + kOccSlots * static_cast<size_t>(occluderCell(pr.oXY.x, pr.oXY.y, pr.oZ));
                    uint32_t        old[kOccSlots], now[kOccSlots];
                    loadOccluderCell(cell, old);
                    if (occluderWord == 0u)
                    {
#pragma unroll
                        for (int k = 0; k < kOccSlots; ++k) now[k] = k + 1 < kOccSlots ? old[k + 1 < kOccSlots ? k + 1 : k] : 0u;
                        storeOccluderCell(cell, now);
                    }
                    else if (occluderWord != old[0])
                    {
                        int at = kOccSlots - 1; // where the word sits already (else: the last place is given up)
#pragma unroll
                        for (int k = kOccSlots - 2; k >= 1; --k)
                            if (old[k] == occluderWord) at = k;
                        now[0] = occluderWord;
#pragma unroll
                        for (int k = 1; k < kOccSlots; ++k) now[k] = k <= at ? old[k - 1] : old[k];
                        storeOccluderCell(cell, now);
                    }
                }
            }
            if (ANY_HIT)
            {
                const float visibility = occluded ? 0.0f : 1.0f;
                const Vec3  add = (pendingTerm * visibility) * __uint_as_float(kSolarInvPdfBits);
                // An occluded ray adds pending * 0 = +-0 to a sum that is never -0 (it starts at +0, and x + y = -0 only for two
                // negative zeros): the sum keeps its bits, so its slot -- a random 16-byte read-modify-write by now -- is left
                // alone.  Not at bounce 1 (the sum is not in memory yet), and not when the product is NaN (an infinite or NaN
                // NEE term times 0: the reference's sum turns NaN, and so does this one).
                const bool unchanged = !firstBounce && add.x == 0.0f && add.y == 0.0f && add.z == 0.0f;
                if (!unchanged)
                {
                    const Vec3 radiance = (firstBounce ? vec3(0.0f, 0.0f, 0.0f) : load3s(ps.rad + slot)) + add; // bounce 1: still 0 (wgsl:183)
                    store4s(ps.rad + slot, radiance.x, radiance.y, radiance.z, 0.0f);
                }
            }
            else
            {
                // .w = t of the hit (rayTMax == best.t then); read by the query path only
                store4s(ps.hit + resultIndex, __uint_as_float(best.triangle), best.u, best.v, rayTMax);
            }
            node = kNodeIdle;
        }
    }

    if (COUNT)
    {
        const unsigned long long nv = waveSum(tc.nodesVisited), tt = waveSum(tc.triangleTests);
        const uint32_t           sh = waveMax(tc.stackHigh);
        if (lane == 0)
        {
            atomicAdd(ANY_HIT ? &counters->shadowNodeVisits : &counters->closestNodeVisits, nv);
            atomicAdd(ANY_HIT ? &counters->shadowTriangleTests : &counters->closestTriangleTests, tt);
            if (!ANY_HIT) atomicMax(&counters->stackHigh, sh);
        }
        const unsigned long long rf = waveSum(recordFetches);
        if (lane == 0) atomicAdd(ANY_HIT ? &counters->shadowRecordFetches : &counters->closestRecordFetches, rf);
        // wave-level trips: a loop body executed by the wave counts once whatever the number of active lanes
        // (lanes that were active carry the count; take the max over the wave), except pops (lane work)
        const int                k = ANY_HIT ? 1 : 0;
        const unsigned long long pops = waveSum(wPop);
        const uint32_t           d = waveMax(wDescend), l = waveMax(wLeaf), lp = waveMax(wLeafPhase), r = waveMax(wRefill), o = waveMax(wOuter);
        if (lane == 0)
        {
            atomicAdd(&counters->descendTrips[k], static_cast<unsigned long long>(d));
            atomicAdd(&counters->leafTrips[k], static_cast<unsigned long long>(l));
            atomicAdd(&counters->leafPhases[k], static_cast<unsigned long long>(lp));
            atomicAdd(&counters->refillTrips[k], static_cast<unsigned long long>(r));
            atomicAdd(&counters->popLaneTrips[k], pops);
            atomicAdd(&counters->outerTrips[k], static_cast<unsigned long long>(o));
        }
    }
#if defined(RF_EXP_PHASE)
    if (!COUNT)
    {
        const int                k = ANY_HIT ? 1 : 0;
        const unsigned long long laneSteps = waveSum(recordFetches), laneLeaves = waveSum(wLeafPhase), laneTris = waveSum(phaseTris);
        const uint32_t           d = waveMax(wDescend), lp = waveMax(phaseLeafWave), r = waveMax(wRefill), o = waveMax(wOuter);
        if (lane == 0)
        {
            atomicAdd(ANY_HIT ? &counters->shadowRecordFetches : &counters->closestRecordFetches, laneSteps);
            atomicAdd(&counters->descendTrips[k], static_cast<unsigned long long>(d));
            atomicAdd(&counters->leafPhases[k], static_cast<unsigned long long>(lp));
            atomicAdd(&counters->leafTrips[k], laneLeaves);
            atomicAdd(&counters->popLaneTrips[k], laneTris);
            atomicAdd(&counters->refillTrips[k], static_cast<unsigned long long>(r));
            atomicAdd(&counters->outerTrips[k], static_cast<unsigned long long>(o));
        }
        const unsigned long long pk = waveSum(phaseParked), pi = waveSum(phaseIdle), li = waveSum(phaseLeafInterior), ld = waveSum(phaseLeafIdle);
        if (lane == 0) atomicAdd(&counters->descendParked[k], pk), atomicAdd(&counters->descendIdle[k], pi), atomicAdd(&counters->leafInterior[k], li), atomicAdd(&counters->leafIdle[k], ld);
        const unsigned long long ot = waveSum(phaseOccTried), oh = waveSum(phaseOccHit), oc = waveSum(phaseOccluded);
        if (lane == 0 && ANY_HIT) atomicAdd(&counters->occluderTried, ot), atomicAdd(&counters->occluderHit, oh), atomicAdd(&counters->occludedRays, oc);
    }
#endif
    if (blockIdx.x == 0 && threadIdx.x == 0 && !(flags & kFlagNoRayCount)) atomicAdd(ANY_HIT ? &counters->shadowRays : &counters->closestRays, static_cast<unsigned long long>(count));
}


// ------------------------------------------------------------------------------------------------
// kShadowFirstLook: the occluder cache (kTraceWide, kFlagOccluderCache) without the traversal kernel around it.  Once the grid is warm nine
// shadow rays in ten are stopped by one of the (up to) kOccSlots leaves their cell names -- 1.4 leaf visits and no interior step at all --
// and a persistent, stack-carrying, lane-refilling kernel is a poor place for work that short.  This kernel walks the bounce's shadow queue
// densely, one ray per lane and nothing to carry: cell of the origin -> its leaves in turn -> each leaf's exact box with the reference's
// formula (leafBoxesIntoTriangles) -> the leaf's triangles.  A ray stopped there is finished (its NEE term times 0, exactly as the
// traversal's write-back adds it; a leaf other than the cell's first moves to the front); every other ray's queue position goes onto a list
// that the traversal launch works through -- without a first look of its own (kFlagOccluderNoTry), recording what it finds in the grid.
//
// Same visibility as the reference's shadowRay (wgsl:321-368), by the argument at kOccluderCache: a triangle is tested there iff the walk
// reaches its leaf, i.e. iff the boxes of the leaf and of all its ancestors pass; an ancestor's box contains the leaf's and the slab
// arithmetic is monotone in the planes, so a ray that passes the leaf's own test passes every ancestor's: the reference either reaches this
// leaf and finds the same triangle, or has found another one before -- occluded either way.  Rays that are not class A (rf_wide.hpp: an
// infinite 1/direction component, a non-finite origin), big leaves and cells without an entry are simply passed on.
// ------------------------------------------------------------------------------------------------
// `inList` (or nullptr = every position of the queue): the queue positions to look at -- what kShade's own-triangle test (kShadeSelfShadow) has not settled; `inCount`: how many.
__global__ __launch_bounds__(kBlock) void kShadowFirstLook(DeviceScene scene, WideScene wide, SkyStateGpu sky, SunBasis sunBasis, PathStreams ps, const uint32_t* queue,
                                                            const uint32_t* inList, const uint32_t* inCount, uint32_t* list, uint32_t* listCount, DeviceCounters* counters, float tMax,
                                                            uint32_t flags)
{
    __shared__ uint32_t sScratch[8];
    const uint32_t      count = *inCount;
    const uint32_t      firstBounce = flags & kLookFirstBounce;
    const uint32_t      tiles = (count + kItems * kBlock - 1) / (kItems * kBlock);
    if (blockIdx.x == 0 && threadIdx.x == 0 && !(flags & kLookNoRayCount)) atomicAdd(&counters->shadowRays, static_cast<unsigned long long>(count));
    // (one entry after the other: staging the kItems entries of a thread -- four cells, then four triangle records in flight per lane -- takes 163
    // registers, three waves per SIMD instead of eight, and measured 17 % slower: profiles/r04_occluder/firstlook2.log)
    for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x)
    {
        bool     keep[kItems];
        uint32_t entry[kItems];
#pragma unroll
        for (int k = 0; k < kItems; ++k)
        {
            const uint32_t j = (tile * kItems + k) * kBlock + threadIdx.x;
            keep[k] = j < count;
            entry[k] = j;
            if (j >= count) continue;
            const uint32_t i = inList != nullptr ? inList[j] : j;
            entry[k] = i;
            // (one dwordx3, pinned: the compiler cuts the 12 bytes into two overlapping dwordx2 for the packed arithmetic further down otherwise)
            v3fu           o3 = *reinterpret_cast<const v3fu*>(ps.rayO + i);
            asm volatile("" : "+v"(o3));
            const Vec3     o = vec3(o3.x, o3.y, o3.z);
            uint32_t* const cell = wide.occGrid + kOccSlots * static_cast<size_t>(occluderCellIndex(wide, o.x, o.y, o.z));
            uint32_t        e[kOccSlots];
            loadOccluderCell(cell, e);
            if (e[0] == 0u) continue;
            const Vec3    nz = load3(ps.noiseOut + i);
            const Vec3    dir = sunSample(sky, sunBasis, nz.x, nz.y, nz.z);
            const RayPrep ray = prepareRay(o, dir);
            if (classifyRay(ray) != kRayPlain) continue;
            entry[k] = i | 0x80000000u; // has tried its cell's leaves
            const PackedRay pr = packRay(ray);
            int             at = -1; // which of the cell's leaves stopped the ray
#pragma unroll
            for (int j = 0; j < kOccSlots; ++j)
            {
                const uint32_t w = e[j];
                // (a leaf word with its triangle count in the word, not in the big-leaf table)
                if (at >= 0 || (w & kWideLeafBit) == 0u || ((w >> kWideIndexBits) & 7u) == 7u) continue;
                const uint32_t first = w & ((1u << kWideIndexBits) - 1u), n = ((w >> kWideIndexBits) & 7u) + 1u;
                const float4*  t0 = scene.triangles + kTriStride * static_cast<size_t>(first);
                // (four loads, each pinned in its own register tuple: the compiler re-cuts the 64-byte record into six otherwise -- see the leaf phase of kTraceWide)
                typedef float v4f __attribute__((ext_vector_type(4)));
                const v4f*     t4 = reinterpret_cast<const v4f*>(t0);
                v4f            a = t4[0], b = t4[1], c = t4[2];
                v3f            hi = *reinterpret_cast<const v3f*>(t0 + 3);
                asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(hi));
                float          bn, bf;
                bool           boxNaN;
                slabSingleBounds(pr, a.w, b.w, c.w, hi.x, hi.y, hi.z, bn, bf, boxNaN);
                if (!(bn <= bf && bf > 0.0f && bn < tMax)) continue; // the reference rejects this leaf
                TriangleHit th;
                bool        stopped = intersectTriangle(o, dir, vec3(a.x, a.y, a.z), vec3(b.x, b.y, b.z), vec3(c.x, c.y, c.z), tMax, th);
                for (uint32_t t = 1; t < n && !stopped; ++t)
                {
                    const v3f q0 = *reinterpret_cast<const v3f*>(t0 + kTriStride * t), q1 = *reinterpret_cast<const v3f*>(t0 + kTriStride * t + 1),
                              q2 = *reinterpret_cast<const v3f*>(t0 + kTriStride * t + 2);
                    stopped = intersectTriangle(o, dir, vec3(q0.x, q0.y, q0.z), vec3(q1.x, q1.y, q1.z), vec3(q2.x, q2.y, q2.z), tMax, th);
                }
                if (stopped) at = j;
            }
            if (at < 0) continue;
            keep[k] = false;
            if (at > 0)
            {
                uint32_t now[kOccSlots];
                now[0] = e[at];
#pragma unroll
                for (int j = 1; j < kOccSlots; ++j) now[j] = j <= at ? e[j - 1] : e[j];
                storeOccluderCell(cell, now);
            }
            // the traversal's write-back for an occluded ray (kTraceWide): radiance += (pending * 0) * invPdf -- a sum that keeps its bits unless the
            // product is NaN, or the sum is not in memory yet (bounce 1)
            const Vec3 add = (load3(ps.pending + i) * 0.0f) * __uint_as_float(kSolarInvPdfBits);
            const bool unchanged = firstBounce == 0u && add.x == 0.0f && add.y == 0.0f && add.z == 0.0f;
            if (!unchanged)
            {
                const uint32_t slot = queue[i];
                const Vec3     radiance = (firstBounce != 0u ? vec3(0.0f, 0.0f, 0.0f) : load3(ps.rad + slot)) + add;
                ps.rad[slot] = make_float4(radiance.x, radiance.y, radiance.z, 0.0f);
            }
        }
        blockAppend<kItems>(keep, entry, list, listCount, sScratch);
    }
}


// ------------------------------------------------------------------------------------------------
// kTracePacket: 64 consecutive queue entries = ONE packet that walks the tree in lockstep.
//
// With the pixel-major, direction-sorted slot order the 64 rays of a wave at bounce 1 are 64 samples of one pixel:
// (almost) one origin, one direction.  Such a wave does not need 64 private traversals.  The packet runs the
// reference's depth-first order ONCE -- wave-uniform node, wave-uniform stack, records and triangles through the
// scalar cache (s_load: no vector-L1 traffic for the tree at all), scalar branches -- and every lane carries only its
// own ray, its own rayTMax and an `active` bit:
//
//   at a record:   hitN/hitF per lane as in kTraceWide (P(child) && tmin < rayTMax, for lanes active at this node);
//                  any lane enters near -> the packet enters near with active = hitN, and far is pushed (if any lane
//                  hits it) with EVERY lane's own tmin (+inf for lanes that do not hit it); no lane near but some far
//                  -> the packet enters far directly; none -> pop
//   at a pop:      active = (the lane's stored tmin < the lane's rayTMax NOW) -- the reference's test at pop time;
//                  an entry no lane wants is skipped
//   at a leaf:     the active lanes test the leaf's triangles in order.
//
// A lane is active at a node iff its own traversal would visit that node, and the nodes at which it is active come in
// its own depth-first order PROVIDED the near/far order is the lane's: the order is dirNeg[splitAxis] (wgsl:409-417),
// so a closest-hit packet is formed of lanes with equal direction signs (a wave with mixed signs -- pixels on the
// screen's axes -- runs one pass per sign pattern).  Its rayTMax therefore evolves exactly as in the reference and
// hit{triangle,u,v,t} are bit-identical.  Any-hit packets take all lanes at once and choose the order by vote (the
// visibility bit does not depend on the order: see NEAREST_FIRST above); an occluded lane drops out with rayTMax = -inf.
// Rays that are not class A (rf_wide.hpp), and the members of a packet whose shared stack outgrows kPacketDepth, are
// redone by the scalar reference-ordered traversal, as in kTraceWide.
// ------------------------------------------------------------------------------------------------
#if defined(RF_EXP_LEGACY_LAYOUTS) // (round 5: the packet kernel lost to kTraceWide in round 2 and has been off since; `make EXP=RF_EXP_LEGACY_LAYOUTS` builds it, the compact-capable and the 32-byte records)
constexpr int kPacketDepth = 24; // shared stack entries per wave (<= 64): per-lane tmin [depth][lane] in LDS + one child word per entry

template<bool ANY_HIT>
__global__ __launch_bounds__(kBlock, 6) void kTracePacket(DeviceScene scene, WideScene wide, SkyStateGpu sky, SunBasis sunBasis, PathStreams ps,
                                                          const uint32_t* queue, const uint32_t* queueCount, DeviceCounters* counters, float tMax, uint32_t flags)
{
    __shared__ float    sTMin[kPacketDepth * kBlock];
    const uint32_t      count = *queueCount;
    const uint32_t      lane = __lane_id(), wave = threadIdx.x >> 6;
    const bool          shadowDirFromStream = flags & kFlagShadowDirFromStream;
    const bool          firstBounce = flags & kFlagFirstBounce;
    const float         kInf = __uint_as_float(0x7F800000u);
    float* const        myTMin = sTMin + threadIdx.x;
    const uint32_t      numChunks = (count + 63u) / 64u;
    const uint32_t      totalWaves = gridDim.x * (kBlock / 64);

    for (uint32_t chunkIdx = blockIdx.x * (kBlock / 64) + wave; chunkIdx < numChunks; chunkIdx += totalWaves)
    {
        const uint32_t idx = chunkIdx * 64u + lane;
        const bool     valid = idx < count;
        uint32_t       slot = 0;
        Vec3           o = vec3(0.0f, 0.0f, 0.0f), dir = vec3(0.0f, 0.0f, 1.0f);
        if (valid)
        {
            if (ANY_HIT) slot = loadQ(queue + idx);
            o = load3s(ps.rayO + idx);
            if (ANY_HIT && !shadowDirFromStream)
            {
                const Vec3 nz = load3s(ps.noiseOut + idx);
                dir = sunSample(sky, sunBasis, nz.x, nz.y, nz.z);
            }
            else dir = load3s(ps.rayD + idx);
        }
        const RayPrep   ray = prepareRay(o, dir);
        const PackedRay pr = packRay(ray);
        const uint32_t  rayClass = classifyRay(ray);
        const uint32_t  negMask = ray.negX | (ray.negY << 1) | (ray.negZ << 2);
        bool            needScalar = valid && rayClass != kRayPlain;
        const bool      regular = valid && rayClass == kRayPlain;
        float           rootTMin;
        const bool      rootOk = slabBounds(ray, wide.rootLo, wide.rootHi, rootTMin);
        ClosestHit      best{};
        best.triangle = kMiss;
        float resultT = tMax;  // closest: t of the hit (tMax: none); any-hit: -inf once occluded
        bool  occluded = false;

        unsigned long long todo = __ballot(regular);
        while (todo != 0ull)
        {
            // members of this pass: closest-hit -- the lanes that share the first waiting lane's direction signs
            bool member = regular;
            if (!ANY_HIT)
            {
                const uint32_t leader = static_cast<uint32_t>(__ffsll(static_cast<long long>(todo))) - 1u;
                const uint32_t uNeg = __builtin_amdgcn_readlane(negMask, leader);
                member = regular && ((todo >> lane) & 1ull) != 0ull && negMask == uNeg;
            }
            const unsigned long long memberMask = __ballot(member);
            todo &= ~memberMask;
            const uint32_t passNeg = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_readlane(negMask, static_cast<uint32_t>(__ffsll(static_cast<long long>(memberMask))) - 1u));

            float limit = member ? tMax : -kInf; // the lane's rayTMax; -inf: every comparison `t < limit` fails
            bool  active = rootOk && rootTMin < limit;
            if (__ballot(active) == 0ull) continue;
            uint32_t node = wide.rootLeaf != kWideNone ? wide.rootLeaf : 0u;
            int      depth = 0;
            bool     overflow = false;
            uint32_t wordStack = 0; // the shared stack's child words: entry d lives in lane d of this register (v_writelane / v_readlane)
            for (;;)
            {
                node = __builtin_amdgcn_readfirstlane(node);
                bool popNow = false;
                if (static_cast<int32_t>(node) >= 0)
                {
                    typedef uint32_t u8v __attribute__((ext_vector_type(8)));
                    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
                    typedef uint32_t u2v __attribute__((ext_vector_type(2)));
                    const float4* un = wide.nodes + 4 * static_cast<size_t>(node);
                    u8v           a;
                    u4v           b;
                    u2v           c;
                    asm volatile("s_load_dwordx8 %0, %3, 0x0\n\ts_load_dwordx4 %1, %3, 0x20\n\ts_load_dwordx2 %2, %3, 0x30\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&s"(a), "=&s"(b), "=&s"(c)
                                 : "s"(un)
                                 : "memory");
                    const float4   q0 = make_float4(__uint_as_float(a.s0), __uint_as_float(a.s1), __uint_as_float(a.s2), __uint_as_float(a.s3));
                    const float4   q1 = make_float4(__uint_as_float(a.s4), __uint_as_float(a.s5), __uint_as_float(a.s6), __uint_as_float(a.s7));
                    const float4   q2 = make_float4(__uint_as_float(b.x), __uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w));
                    const uint32_t axis = (c.x >> kWideAxisShift) & 3u;
                    const uint32_t word0 = c.x & ~(3u << kWideAxisShift), word1 = c.y;
                    float          t0, t1;
                    bool           ok0, ok1;
                    slabPair(pr, q0, q1, q2, ok0, t0, ok1, t1);
                    const bool hit0 = active && ok0 && t0 < limit, hit1 = active && ok1 && t1 < limit;
                    const unsigned long long m0 = __ballot(hit0), m1 = __ballot(hit1);
                    // which child is "near": the reference's split-axis order (closest-hit), a vote (any-hit)
                    bool secondFirst;
                    if (ANY_HIT) secondFirst = 2 * __popcll(__ballot(hit0 && hit1 && t1 < t0)) > __popcll(m0 & m1);
                    else secondFirst = ((passNeg >> axis) & 1u) != 0u;
                    const unsigned long long mN = secondFirst ? m1 : m0, mF = secondFirst ? m0 : m1;
                    const uint32_t           nearWord = secondFirst ? word1 : word0, farWord = secondFirst ? word0 : word1;
                    const bool               hitN = secondFirst ? hit1 : hit0, hitF = secondFirst ? hit0 : hit1;
                    const float              tF = secondFirst ? t0 : t1;
                    if (mN != 0ull)
                    {
                        if (mF != 0ull)
                        {
                            if (depth >= kPacketDepth)
                            {
                                overflow = true;
                                break;
                            }
                            myTMin[depth * kBlock] = hitF ? tF : kInf;
                            {
                                // v_writelane takes its lane select from m0 when the value is an SGPR too (constant-bus limit); m0 is put back
                                uint32_t keepM0;
                                asm volatile("s_mov_b32 %1, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tv_writelane_b32 %0, %2, m0\n\ts_mov_b32 m0, %1"
                                             : "+v"(wordStack), "=&s"(keepM0)
                                             : "s"(farWord), "s"(depth));
                            }
                            ++depth;
                        }
                        node = nearWord;
                        active = hitN;
                    }
                    else if (mF != 0ull)
                    {
                        node = farWord;
                        active = hitF;
                    }
                    else popNow = true;
                }
                else
                {
                    // ---- leaf: the active lanes test its triangles in order
                    uint32_t first = node & ((1u << kWideIndexBits) - 1u), n = ((node >> kWideIndexBits) & 7u) + 1u;
                    if (n == 8u)
                    {
                        const uint2 big = wide.bigLeaves[first];
                        first = __builtin_amdgcn_readfirstlane(big.x);
                        n = __builtin_amdgcn_readfirstlane(big.y);
                    }
                    for (uint32_t i = 0; i < n; ++i)
                    {
                        typedef uint32_t u8v __attribute__((ext_vector_type(8)));
                        typedef uint32_t u4v __attribute__((ext_vector_type(4)));
                        const uint32_t tri = first + i;
                        const float4*  ut = scene.triangles + kTriStride * static_cast<size_t>(tri);
                        u8v            ab;
                        u4v            cc;
                        asm volatile("s_load_dwordx8 %0, %2, 0x0\n\ts_load_dwordx4 %1, %2, 0x20\n\ts_waitcnt lgkmcnt(0)" : "=&s"(ab), "=&s"(cc) : "s"(ut) : "memory");
                        const Vec3 p0 = vec3(__uint_as_float(ab.s0), __uint_as_float(ab.s1), __uint_as_float(ab.s2));
                        const Vec3 p1 = vec3(__uint_as_float(ab.s4), __uint_as_float(ab.s5), __uint_as_float(ab.s6));
                        const Vec3 p2 = vec3(__uint_as_float(cc.x), __uint_as_float(cc.y), __uint_as_float(cc.z));
                        TriangleHit th;
                        if (active && intersectTriangle(o, dir, p0, p1, p2, limit, th))
                        {
                            if (ANY_HIT)
                            {
                                occluded = true;
                                limit = -kInf;
                                active = false;
                            }
                            else
                            {
                                limit = th.t;
                                best.u = th.u;
                                best.v = th.v;
                                best.triangle = tri;
                            }
                        }
                    }
                    if (ANY_HIT && __ballot(member && !occluded) == 0ull) break; // every member has its answer
                    popNow = true;
                }
                if (popNow)
                {
                    bool found = false;
                    while (depth > 0)
                    {
                        --depth;
                        const float tm = myTMin[depth * kBlock];
                        active = tm < limit;
                        if (__ballot(active) != 0ull)
                        {
                            node = __builtin_amdgcn_readlane(wordStack, static_cast<uint32_t>(depth));
                            found = true;
                            break;
                        }
                    }
                    if (!found) break;
                }
            }
            if (overflow)
            {
                // deeper than the shared stack: the members of this pass are redone one by one
                if (member)
                {
                    needScalar = true;
                    best.triangle = kMiss;
                    occluded = false;
                }
            }
            else if (member) resultT = limit;
        }

        if (needScalar)
        {
            TraversalCounters c2;
            atomicAdd(&counters->scalarRedo[ANY_HIT ? 1 : 0], 1ull);
            best.triangle = kMiss;
            occluded = traverse<ANY_HIT, false, 0>(scene, o, dir, tMax, nullptr, best, c2);
            if (c2.abandoned) atomicAdd(&counters->abandonedRays, 1ull);
            resultT = best.triangle != kMiss ? best.t : tMax;
        }
        if (valid)
        {
            if (ANY_HIT)
            {
                const float visibility = occluded ? 0.0f : 1.0f;
                const Vec3  add = (load3s(ps.pending + idx) * visibility) * __uint_as_float(kSolarInvPdfBits);
                const Vec3  radiance = (firstBounce ? vec3(0.0f, 0.0f, 0.0f) : load3s(ps.rad + slot)) + add;
                store4s(ps.rad + slot, radiance.x, radiance.y, radiance.z, 0.0f);
            }
            else store4s(ps.hit + idx, __uint_as_float(best.triangle), best.u, best.v, resultT);
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(ANY_HIT ? &counters->shadowRays : &counters->closestRays, static_cast<unsigned long long>(count));
}
#endif // RF_EXP_LEGACY_LAYOUTS

// Query path: offset hit points of a hit stream (the render path does this in kShade).
__global__ void kHitPoints(DeviceScene scene, const float4* hit, P3* rayO, uint32_t n)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4   h = hit[i];
    const uint32_t tri = __float_as_uint(h.x);
    if (tri == kMiss) return;
    const Vec3 hp = hitPoint(scene, tri, h.y, h.z);
    store3(rayO + i, hp);
}

// bvh-visualizer pass (src/bvh-visualizer/main.cpp:60-78): pinhole camera.cpp:44-52 rays.
__global__ __launch_bounds__(kBlock) void kPrimaryStats(DeviceScene scene, Camera cam, uint32_t width, uint32_t height,
                                                         uint32_t* nodesVisited, uint8_t* hitOut, float* tOut, uint32_t* triTests, DeviceCounters* counters)
{
    __shared__ uint32_t sStack[kLdsStack * kBlock];
    // 8x8 pixel blocks per wave for coherence; output is row-major
    const uint32_t blocksX = (width + 7u) / 8u;
    const uint32_t wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t j = (wave % blocksX) * 8u + (lane & 7u);
    const uint32_t i = (wave / blocksX) * 8u + (lane >> 3);
    if (j >= width || i >= height) return;
    const float u = static_cast<float>(j) / static_cast<float>(width);
    const float v = 1.0f - static_cast<float>(i + 1) / static_cast<float>(height);
    const Vec3  dir = normalize(cam.lowerLeftCorner + cam.horizontal * u + cam.vertical * v - cam.origin);
    ClosestHit        h;
    TraversalCounters tc;
    const bool        found = traverse<false, true>(scene, cam.origin, dir, FLT_MAX, &sStack[threadIdx.x], h, tc);
    if (tc.abandoned) atomicAdd(&counters->abandonedRays, 1ull);
    const size_t      k = static_cast<size_t>(i) * width + j;
    nodesVisited[k] = tc.nodesVisited;
    if (hitOut) hitOut[k] = found ? 1 : 0;
    if (tOut) tOut[k] = found ? h.t : 0.0f;
    if (triTests) triTests[k] = tc.triangleTests;
}

__global__ __launch_bounds__(kBlock) void kIntersectRays(DeviceScene scene, const float* rays, uint64_t n, float tMax, uint32_t* triOut,
                                                          float* tOut, float* uvOut, float* pOut, uint32_t* nvOut, uint32_t* ttOut, DeviceCounters* counters)
{
    __shared__ uint32_t sStack[kLdsStack * kBlock];
    const uint64_t      i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n) return;
    const float*      r = rays + 6 * i;
    ClosestHit        h;
    TraversalCounters tc;
    const bool        found = traverse<false, true>(scene, vec3(r[0], r[1], r[2]), vec3(r[3], r[4], r[5]), tMax, &sStack[threadIdx.x], h, tc);
    if (tc.abandoned) atomicAdd(&counters->abandonedRays, 1ull);
    triOut[i] = h.triangle;
    if (tOut) tOut[i] = found ? h.t : 0.0f;
    if (uvOut)
    {
        uvOut[2 * i] = found ? h.u : 0.0f;
        uvOut[2 * i + 1] = found ? h.v : 0.0f;
    }
    if (pOut)
    {
        pOut[3 * i] = found ? h.p.x : 0.0f;
        pOut[3 * i + 1] = found ? h.p.y : 0.0f;
        pOut[3 * i + 2] = found ? h.p.z : 0.0f;
    }
    if (nvOut) nvOut[i] = tc.nodesVisited;
    if (ttOut) ttOut[i] = tc.triangleTests;
}

__global__ __launch_bounds__(kBlock) void kOccludedRays(DeviceScene scene, const float* rays, uint64_t n, float tMax, float* visOut, DeviceCounters* counters)
{
    __shared__ uint32_t sStack[kLdsStack * kBlock];
    const uint64_t      i = static_cast<uint64_t>(blockIdx.x) * kBlock + threadIdx.x;
    if (i >= n) return;
    const float*      r = rays + 6 * i;
    ClosestHit        h;
    TraversalCounters tc;
    const bool        occluded = traverse<true, false>(scene, vec3(r[0], r[1], r[2]), vec3(r[3], r[4], r[5]), tMax, &sStack[threadIdx.x], h, tc);
    if (tc.abandoned) atomicAdd(&counters->abandonedRays, 1ull);
    visOut[i] = occluded ? 0.0f : 1.0f;
}


} // namespace

namespace kern
{
TraceWideKernel traceWideKernel(bool anyHit, bool count, bool nearestFirst, int compact, bool denseLeaves)
{
    // the instantiations this build ships: the layouts the renderer picks by itself (+ the oct records, an option), each closest-hit one and the any-hit
    // variants it pairs them with; with the dense leaf phase only where a default launch can reach it (rf_renderer.hip: launchClosestWide / launchShadowWide)
#define RF_TW(A, C, N, L, D) \
    if (anyHit == A && count == C && nearestFirst == N && compact == L && denseLeaves == D) return kTraceWide<A, C, N, L, D>;
    RF_TW(false, true, false, 0, false) RF_TW(true, true, true, 0, false) RF_TW(true, true, false, 0, false)
    RF_TW(false, false, false, 0, false) RF_TW(false, false, false, 3, false) RF_TW(false, false, false, 4, false) RF_TW(false, false, false, 5, false) RF_TW(false, false, false, 6, false)
    RF_TW(false, false, false, 3, true) RF_TW(false, false, false, 4, true) RF_TW(false, false, false, 5, true)
    RF_TW(true, false, true, 0, false) RF_TW(true, false, false, 0, false)
    RF_TW(true, false, true, 3, false) RF_TW(true, false, false, 3, false) RF_TW(true, false, true, 4, false) RF_TW(true, false, false, 4, false) RF_TW(true, false, true, 5, false) RF_TW(true, false, false, 5, false)
    RF_TW(true, false, true, 3, true) RF_TW(true, false, false, 4, true) RF_TW(true, false, false, 5, true)
#if defined(RF_EXP_LEGACY_LAYOUTS)
    RF_TW(false, false, false, 1, false) RF_TW(false, false, false, 2, false) RF_TW(true, false, true, 1, false) RF_TW(true, false, true, 2, false)
#endif
#undef RF_TW
    return nullptr;
}
TraceClosestKernel    traceClosestKernel(bool count) { return count ? kTraceClosest<true> : kTraceClosest<false>; }
TraceShadowKernel     traceShadowKernel(bool count) { return count ? kTraceShadow<true> : kTraceShadow<false>; }
ShadowFirstLookKernel shadowFirstLookKernel() { return kShadowFirstLook; }
TracePacketKernel     tracePacketKernel(bool anyHit)
{
#if defined(RF_EXP_LEGACY_LAYOUTS)
    return anyHit ? kTracePacket<true> : kTracePacket<false>;
#else
    (void)anyHit;
    return nullptr;
#endif
}
HitPointsKernel     hitPointsKernel() { return kHitPoints; }
PrimaryStatsKernel  primaryStatsKernel() { return kPrimaryStats; }
IntersectRaysKernel intersectRaysKernel() { return kIntersectRays; }
OccludedRaysKernel  occludedRaysKernel() { return kOccludedRays; }
} // namespace kern
} // namespace rf
