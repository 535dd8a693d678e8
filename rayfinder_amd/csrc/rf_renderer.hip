// rf_renderer.hip -- wavefront path tracer for MI355X (gfx950): the host driver (the kernels live in rf_trace.hip and rf_shade.hip, what the three
// units share in rf_kernels.hpp).
//
// The reference traces one full path per fragment-shader invocation
// (src/pt/reference_path_tracer.wgsl:34-64,180-234).  Here the same per-path arithmetic is cut
// into a wavefront pipeline so that every stage runs with full, coherent waves:
//
//   raygen            wgsl:42-54,236-245,594-616   S samples x all pixels of this rank's tiles
//   for bounce = 1..numBounces
//     traceClosest    wgsl:370-521                 queue of live paths -> hit record, origin <- p
//     shade           wgsl:189-228,247-319,546-592 miss: += throughput*sky, path ends
//                                                  hit : albedo, pending NEE term, next direction,
//                                                        throughput *= albedo; ballot-compacted queue
//     traceShadow     wgsl:321-368                 radiance += pending * visibility * invPdf
//   accumulate        wgsl:47-57                   image += radiance, samples in index order (f32)
//
// Path state lives in HBM as six float4 streams indexed by path slot (slot = sample*pixels +
// pixel, fixed for the life of the path); queues hold slot ids.  Because the reference reuses ONE
// blue-noise pair for lens, sun cone and every bounce (wgsl:52-55,194,209), cos/sin of 2*pi*u.y
// are evaluated once per path in raygen and carried in the .w lanes.
//
// Kernel grids are sized for the worst case (all paths alive) and read the live count from
// device memory, so a whole batch is enqueued without any host round trip.
#include "rf_kernels.hpp"

#include <map>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <stdexcept>
#include <string>
#include <type_traits>

namespace rf
{
bool operator==(const RenderParameters& a, const RenderParameters& b)
{
    return a.width == b.width && a.height == b.height && std::memcmp(&a.camera, &b.camera, sizeof(Camera)) == 0 &&
           a.samplingParams == b.samplingParams && a.sky == b.sky && a.exposure == b.exposure;
}

namespace
{
template<typename T>
struct DeviceBuffer
{
    T*     ptr = nullptr;
    size_t count = 0;
    void   alloc(size_t n)
    {
        release();
        count = n;
        if (n) RF_HIP(hipMalloc(reinterpret_cast<void**>(&ptr), n * sizeof(T)));
    }
    void upload(const T* src, size_t n)
    {
        alloc(n);
        if (n) RF_HIP(hipMemcpy(ptr, src, n * sizeof(T), hipMemcpyHostToDevice));
    }
    void release()
    {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        count = 0;
    }
    ~DeviceBuffer() { release(); }
};

// Morton (Z-order) key of a tile position
uint32_t tileMortonKey(uint32_t tx, uint32_t ty)
{
    const auto spread = [](uint32_t v) {
        v &= 0xFFFFu;
        v = (v | (v << 8)) & 0x00FF00FFu;
        v = (v | (v << 4)) & 0x0F0F0F0Fu;
        v = (v | (v << 2)) & 0x33333333u;
        v = (v | (v << 1)) & 0x55555555u;
        return v;
    };
    return spread(tx) | (spread(ty) << 1);
}
} // namespace

std::vector<uint32_t> tilesForRank(uint32_t width, uint32_t height, uint32_t rank, uint32_t worldSize)
{
    const uint32_t tilesX = (width + kTileSize - 1) / kTileSize, tilesY = (height + kTileSize - 1) / kTileSize;
    const uint32_t n = tilesX * tilesY;
    std::vector<uint32_t> order(n);
    for (uint32_t i = 0; i < n; ++i) order[i] = i;
    if (worldSize > 1)
    {
        // Tiles walked along a Z-order curve; each run of `world` consecutive tiles of the curve -- a compact block of the image
        // (4x2 tiles for 8 ranks, 2x2 for 4) -- is split over all ranks, and the deal is rotated by one rank from block to block so
        // that no rank always gets the same corner of a block (tile cost has a vertical gradient: a fixed position is a
        // systematic bias).  Every rank gets n/world +-1 tiles.  Neighbouring tiles cost about the same (sky vs interior), so the
        // ranks' loads are stratified samples of the frame: on the atrium at 1080p the busiest of 8 ranks traces 1.0 % more rays
        // than the mean, against 3.0 % for the hashed deal used before (3.6 % unrotated); the strong-scaling time is the
        // slowest rank's.
        std::stable_sort(order.begin(), order.end(), [tilesX](uint32_t a, uint32_t b) { return tileMortonKey(a % tilesX, a / tilesX) < tileMortonKey(b % tilesX, b / tilesX); });
    }
    std::vector<uint32_t> mine;
    for (uint32_t i = 0; i < n; ++i)
        if ((i % worldSize + i / worldSize) % worldSize == rank) mine.push_back(order[i]);
    std::sort(mine.begin(), mine.end());
    return mine;
}

void untileHost(const float* compact, const uint32_t* tileIds, uint32_t numTiles, uint32_t width, uint32_t height, float* image)
{
    const uint32_t tilesX = (width + kTileSize - 1) / kTileSize;
    for (uint32_t t = 0; t < numTiles; ++t)
    {
        const uint32_t tx = tileIds[t] % tilesX, ty = tileIds[t] / tilesX;
        for (uint32_t w = 0; w < kTileSize * kTileSize; ++w)
        {
            const uint32_t block = w >> 6, lane = w & 63u;
            const uint32_t x = tx * kTileSize + (block & 3u) * 8u + (lane & 7u);
            const uint32_t y = ty * kTileSize + (block >> 2) * 8u + (lane >> 3);
            if (x >= width || y >= height) continue;
            std::memcpy(image + 4 * (static_cast<size_t>(y) * width + x), compact + 4 * (static_cast<size_t>(t) * 1024 + w), 16);
        }
    }
}

// ------------------------------------------------------------------------------------------------
struct Renderer::Impl
{
    int         device = 0;
    hipStream_t stream = nullptr;

    DeviceBuffer<float4>            nodes, triangles, wideNodes, wideCompact, wideHot, wideOwn, wideQuad;
    DeviceBuffer<uint4>             wideQuadHalf, wideQuadLocal, wideOct;
    DeviceBuffer<uint2>             bigLeaves;
    WideScene                       wide{};
    DeviceBuffer<float4>            attributes; // 4 per triangle (packed, see the constructor)
    DeviceBuffer<float4>            shadeRecords; // 8 per triangle: kShade's 128-byte record
    DeviceBuffer<TextureDescriptor> textureDescriptors;
    DeviceBuffer<uint32_t>          texels;
    DeviceBuffer<uint8_t>           blueNoise;
    DeviceBuffer<float>             albedoLut;
    DeviceScene                     scene{};

    RenderParameters params;
    SkyStateGpu      sky{};
    SunBasis         sunBasis{};

    void updateSunBasis() { pixarOnb(vec3(sky.sunDirection[0], sky.sunDirection[1], sky.sunDirection[2]), sunBasis.u, sunBasis.v); }
    uint32_t         maxWidth = 0, maxHeight = 0;
    uint32_t         frameCount = 0, accumulated = 0;
    uint32_t         rank = 0, worldSize = 1;

    std::vector<uint32_t>   tiles;
    DeviceBuffer<uint32_t>  tileIds;
    DeviceBuffer<uint32_t>  tileValidBefore; // prefix sum of the valid pixels of the shard's tiles (numTiles + 1): kRaygen's queue positions without an atomic (FrameParams)
    bool                    optDenseRaygen = true;
    DeviceBuffer<float4>    ownedImage;
    float4*                 image = nullptr; // compact tile-major accumulation buffer
    uint64_t                imageBytes = 0;
    bool                    imageDirty = true; // needs zeroing before the next sample

    uint64_t                validPixels = 0;     // pixels of this rank's tiles that lie inside the frame
    unsigned long long      primaryRaysHost = 0; // samples traced x validPixels since the last resetStats()
    uint64_t                maxPaths = 0;
    DeviceBuffer<P3>        sRayO, sRayD, sRayD2, sThr, sThr2, sPending, sNoise, sNoise2; // queue-position arrays, packed xyz; rayD / thr / noise: double-buffered (PathStreams)
    DeviceBuffer<float4>    sRad, sHit;
    DeviceBuffer<uint32_t>  queueA, queueB, missQueue, missSlots, shadowList, queueCounts; // shadowList: kShade's list of the shadow rays its own-triangle test has not settled (kShadeSelfShadow)
    DeviceBuffer<DeviceCounters> counters;
    DeviceBuffer<unsigned long long> bounceTotals; // 4 x kMaxBounceStats: closest-hit rays, shadow rays, shadow rays answered by kShadowFirstLook, shadow rays settled by kShade's own-triangle test

    // deferred-lighting variant: its own frame counter and buffers (array<array<f32, 3>>)
    uint32_t               deferredFrameCount = 0;
    DeviceBuffer<float>    deferredSample, deferredAccum;
    DeviceBuffer<uint32_t> deferredBgra;

    bool counting = false, timing = false;
    int      traversalVariant = 2; // 0 = one ray per thread over 32-B nodes (A/B baseline), 2 = persistent waves over 64-B wide nodes
    uint32_t wideBlocks = 0;
    bool     wideBlocksForced = false; // option persistent_blocks: every persistent launch takes that grid
    uint32_t multiProcessors = 0;
    std::map<std::pair<const void*, uint32_t>, uint32_t> residentCache; // kernel, extra LDS -> workgroups the device holds at once
    uint32_t skyBlocks = 2048; // grid of the per-bounce kSky launches (grid-stride; set from the CU count)
    bool     wideUsable = true;
    int      queryVariant = 0; // 2: rf_renderer_intersect_rays / _occluded_rays run through kTraceWide (test hook; no per-ray counters)
    bool     shadowNearestFirst = true; // shadow rays: nearest child first (visibility is order independent)
    // the cached any-hit launches of bounce >= this run behind kShadowFirstLook (0: never).  From bounce 2: the coherent launch of bounce 1 costs less per ray than
    // a pass over the queue does (atrium -4 %, Duck +14 % with it: profiles/r04_occluder/firstlook_scenes2.log)
    uint32_t optShadowFirstLookFromBounce = 2;
    bool     occluderGridWarm = false;       // a batch has filled the occluder grid since it was allocated
    // kShadowFirstLook pays where most shadow rays are stopped by their cell's leaves; in a scene lit from everywhere it only adds a pass over the queue.  Every
    // batch reports {answered, rays} of its first looks (asynchronously: read when the copy has landed); below a quarter the next 16 batches go without.
    DeviceBuffer<unsigned long long> lookBatch;
    unsigned long long*              lookBatchHost = nullptr;
    hipEvent_t                       lookEvent = nullptr;
    bool                             lookPending = false;
    uint32_t                         firstLookHoldOff = 0;
    uint32_t occluderHintLevels = 0;  // see leafBoxesIntoTriangles (0: the cache remembers leaves)
    bool     leafBoxesValid = false;   // every leaf's exact box sits in its first triangle record (leafBoxesIntoTriangles)
    uint32_t optOccluderGridLog2Cells = 22; // table size: 2^n cells of kOccSlots words
    uint32_t optOccluderGridCells = 1024; // occluder grid: cells along the longest axis of the root box (0: no occluder cache)
    DeviceBuffer<uint32_t> occluderGrid;
    uint32_t optOccluderCacheBounces = 64; // the any-hit launches of bounces 1..n first visit the leaves their ray's cell of the occluder grid names (kFlagOccluderCache)
    bool     optShadowSignOrder = true; // the half-precision / local-grid shadow launches (VALU bound) visit entries in record order: a cheaper step beats the shorter walks of nearest-first there
    uint32_t optRefillMin = kRefillMin, optLeafVote = kLeafVote, optChunk = kChunk;
    // closest-hit launches of scenes without long leaves whose tree stays cache resident leave the descend loop later (round 6: 14 instead of 20 lanes still descending; plain atrium
    // closest-hit -1.5 %, Duck -1.4 %; the dense-leaf instantiations +4.7 % and the out-of-cache scene +1 % with it: they keep 20 -- profiles/r06_lanes/ab_leaf_vote*.log)
    uint32_t optLeafVoteClosest = 14;
    // leaf phases in which a parked lane holds this many triangles or more run over dense (lane, triangle) pairs (kTraceWide; 0: never), from this bounce on
    uint32_t optDenseLeafMin = 5, optDenseLeafFromBounce = 1; // (5: the plain atrium's leaves of up to 4 triangles keep the loop -- 3 measured +1 % there; the clutter scene gains the same with 2 .. 5)
    uint32_t maxLeafTriangles = 0; // of the scene (upload): scenes without a leaf as long as the threshold run the instantiations WITHOUT the dense block
    uint32_t optShadeSortFromBounce = 2, sortScale = 0;         // kShade of bounce >= this appends its tile's hits in triangle order (0: never)
    uint32_t optChunkEarly = 256, optChunkEarlyBounces = 2;      // queue entries per cursor claim at bounces 1-2
    uint32_t optRefillMinDeep = 12, optRefillMinDeepDense = 22, optRefillMinDeepQuad = 40, optRefillDeepFromBounce = 2; // closest-hit launches of bounce >= 2 (3 until the eager-leaves schedule: bounce 2 -2.6 % with it) refill at another count: 12 idle lanes (22 until round 6: with the
                                                                                             // refill trip's claim logic on the scalar unit and its one-test classification, 16 ... 4 measure the same and 1.3 % under 22: profiles/r06_lanes) on the 64-byte and the
                                                                                             // half-precision quad records (VALU bound: idle lanes cost most), 40 on the exact quad records (L1 bound: a refill is a wave-wide stall)
    // kShade grid cap (0: one workgroup per tile of 1024 entries, the default: workgroups then append to the hit queue in
    // roughly queue order, which keeps neighbouring pixels' rays together -- a capped, grid-striding kShade saved its empty
    // workgroups but cost the traversal kernels 2-5 %)
    uint32_t optShadeBlocks = 0;
    bool     optTranscendentalsF32 = false; // option `transcendentals`: kRaygen / kSky call the f32 math library instead of the specified f64 evaluation (opt-in)
    bool     optShadowSelfTest = true; // kShade tests every shadow ray against the triangle it starts on first (kShadeSelfShadow); needs selfShadowOk
    bool     selfShadowOk = false;     // the tree's boxes are nested and regular, and the shading records carry their triangles' leaf boxes
    bool     optConstPrimaryOrigin = true; // a pinhole camera's primary launch takes its one origin as a kernel argument (kFlagConstOrigin): kRaygen writes no origins
    bool                   optSampleSort = true, optAccumulateRuns = true;
    uint32_t               optCompactFromBounce = 3;       // closest-hit launches of bounce >= this use the compact-capable records (0: never)
    uint32_t               optCompactShadowFromBounce = 2; // ... and the shadow launches of bounce >= this
    uint32_t               optQuadFromBounce = 1, optQuadShadowFromBounce = 1; // the 128-byte quad records (two levels per fetch) from this bounce on (0: never); takes precedence over the others
    uint32_t               optQuadExceptMask = 0, optQuadShadowExceptMask = 0; // ... except at the bounces whose bit (bounce - 1) is set here
    uint32_t               optQuadHalfFromBounce = 0, optQuadHalfShadowFromBounce = 0; // quad launches of bounce >= this read the 64-byte half-precision quad records (0: never; set to 1 at upload when the scene suits them)
    float                  quadHalfAreaRatio = 0.0f;
    uint32_t               optQuadLocalFromBounce = 0, optQuadLocalShadowFromBounce = 0; // ... the 64-byte local-grid quad records (0: never; set to 1 at upload when the half-precision ones do not suit the scene)
    // closest-hit launches of bounce >= this read the 128-byte oct records (three levels per fetch; 0: never -- the default: measured slower, see the upload)
    uint32_t               optOctFromBounce = 0;
    uint64_t               treeBytes = 0; // quad records + triangle records: what the traversal launches touch
    uint32_t               optHotFromBounce = 0, optHotShadowFromBounce = 0; // the 32-byte records (all six planes carried) from this bounce on (0: never); takes precedence
    int                    optQueryCompact = 0;            // the ray-query entry points use the compact-capable (1) / 32-byte (2) records too (tests)
    uint32_t               optExtraLds = 0;      // experiment: dynamic LDS bytes added to the kTraceWide launches (lowers the occupancy)
    uint32_t               optPacketBounces = 0; // bounces 1..n traced by kTracePacket (one wave = one lockstep packet) instead of kTraceWide
    int                    optUniformFetch = 2; // scalar-cache fetch for wave-uniform steps: 0 = never (5 878 Mrays/s), 1 = records (6 039), 2 = records + leaf triangles (6 059), -1 = records at bounces 1-2 only
    DeviceBuffer<uint32_t> samplePerm;
    uint32_t optSlotGroupShift = 0; // see FrameParams::slotGroupShift (r02 A/B on the atrium, Mrays/s: sample-major 5282; unsorted g = 6: 5416, 2: 5507, 0: 5450; with sorted samples g = 2: 5519, 1: 5589, 0: 5650)
    RenderStats hostStats;

    struct TimedLaunch
    {
        hipEvent_t start, stop;
        int        kind;
        uint32_t   bounce;
    };
    std::vector<TimedLaunch> timed;
    std::vector<hipEvent_t>  eventPool;

    struct BatchTiming
    {
        hipEvent_t start, stop;
        uint32_t   samples;
    };
    std::vector<BatchTiming> pendingBatches;
    std::deque<float>        passDurationsMs;

    hipEvent_t getEvent()
    {
        if (!eventPool.empty())
        {
            hipEvent_t e = eventPool.back();
            eventPool.pop_back();
            return e;
        }
        hipEvent_t e;
        RF_HIP(hipEventCreate(&e));
        return e;
    }

    uint64_t allocatedPaths = 0;
    uint64_t effectivePaths = 0;       // batch depth the last render() call ended up with (<= maxPaths: less when less memory was free THEN)

    // bytes of path state + queues per path slot (eight packed xyz streams, two float4 streams, five u32 queues / lists): 148
    static constexpr uint64_t kBytesPerPath = 8 * sizeof(P3) + 2 * sizeof(float4) + 5 * sizeof(uint32_t);

    void releasePathState()
    {
        sRayO.release(), sRayD.release(), sRayD2.release(), sThr.release(), sThr2.release(), sRad.release(), sHit.release();
        sPending.release(), sNoise.release(), sNoise2.release(), queueA.release(), queueB.release(), missQueue.release(), missSlots.release(), shadowList.release();
        allocatedPaths = 0;
    }

    // Path streams and queues are allocated on demand for the largest batch actually traced
    // (at most maxPaths): small renders stay small, big ones use the HBM that is there.
    // Returns false when the device is out of memory: nothing stays allocated then (allocatedPaths = 0), and the
    // caller retries with a smaller batch.  allocatedPaths is only raised once every buffer exists.
    bool ensurePathState(uint64_t paths)
    {
        if (paths <= allocatedPaths) return true;
        RF_HIP(hipStreamSynchronize(stream));
        releasePathState();
        const auto tryAlloc = [](auto& buf, uint64_t n) -> bool {
            void* p = nullptr;
            const hipError_t e = hipMalloc(&p, n * sizeof(*buf.ptr));
            if (e == hipErrorOutOfMemory || e == hipErrorMemoryAllocation)
            {
                (void)hipGetLastError(); // clear the sticky error
                return false;
            }
            RF_HIP(e);
            buf.ptr = static_cast<decltype(buf.ptr)>(p);
            buf.count = n;
            return true;
        };
        const bool ok = tryAlloc(sRayO, paths) && tryAlloc(sRayD, paths) && tryAlloc(sRayD2, paths) && tryAlloc(sThr, paths) && tryAlloc(sThr2, paths) &&
                        tryAlloc(sRad, paths) && tryAlloc(sHit, paths) && tryAlloc(sPending, paths) && tryAlloc(sNoise, paths) && tryAlloc(sNoise2, paths) &&
                        tryAlloc(queueA, paths) && tryAlloc(queueB, paths) && tryAlloc(missQueue, paths) && tryAlloc(missSlots, paths) && tryAlloc(shadowList, paths);
        if (!ok)
        {
            releasePathState();
            return false;
        }
        allocatedPaths = paths;
        return true;
    }

    // Largest batch (in paths) the device can hold right now: what is free plus what the path state already holds, with a
    // tenth kept back for the allocator's granularity and for whoever else uses the device.
    uint64_t pathsThatFit()
    {
        size_t freeBytes = 0, totalBytes = 0;
        RF_HIP(hipMemGetInfo(&freeBytes, &totalBytes));
        const uint64_t budget = static_cast<uint64_t>(freeBytes) + allocatedPaths * kBytesPerPath;
        return budget / 10 * 9 / kBytesPerPath;
    }

    void configureShard()
    {
        tiles = tilesForRank(params.width, params.height, rank, worldSize);
        tileIds.upload(tiles.data(), tiles.size());
        const uint64_t pixelsPadded = static_cast<uint64_t>(tiles.size()) * 1024;
        {
            const uint32_t tilesX = (params.width + kTileSize - 1) / kTileSize;
            validPixels = 0;
            std::vector<uint32_t> before;
            before.reserve(tiles.size() + 1);
            for (const uint32_t t : tiles)
            {
                const uint32_t x0 = (t % tilesX) * kTileSize, y0 = (t / tilesX) * kTileSize;
                before.push_back(static_cast<uint32_t>(validPixels));
                validPixels += static_cast<uint64_t>(std::min(kTileSize, params.width - x0)) * std::min(kTileSize, params.height - y0);
            }
            before.push_back(static_cast<uint32_t>(validPixels));
            tileValidBefore.upload(before.data(), before.size());
        }
        if (image == nullptr || image == ownedImage.ptr)
        {
            ownedImage.alloc(std::max<uint64_t>(pixelsPadded, 1));
            image = ownedImage.ptr;
        }
        else if (imageBytes < pixelsPadded * sizeof(float4))
        {
            throw std::runtime_error("bound accumulation buffer is too small for this shard");
        }
        if (image == ownedImage.ptr) imageBytes = pixelsPadded * sizeof(float4);
        accumulated = 0;
        imageDirty = true;
    }

    template<typename F>
    void launchTimed(int kind, F&& launch, uint32_t bounce = 0)
    {
        if (timing)
        {
            TimedLaunch t{getEvent(), getEvent(), kind, std::min(bounce, RenderStats::kMaxBounceStats - 1)};
            RF_HIP(hipEventRecord(t.start, stream));
            launch();
            RF_HIP(hipEventRecord(t.stop, stream));
            timed.push_back(t);
        }
        else
        {
            launch();
        }
    }

    void collectTimings()
    {
        if (timed.empty()) return;
        RF_HIP(hipStreamSynchronize(stream));
        for (const TimedLaunch& t : timed)
        {
            float ms = 0.0f;
            RF_HIP(hipEventElapsedTime(&ms, t.start, t.stop));
            switch (t.kind)
            {
            case 0: hostStats.msRaygen += ms; hostStats.launchesRaygen++; break;
            case 1: hostStats.msClosest += ms; hostStats.launchesClosest++; hostStats.msClosestByBounce[t.bounce] += ms; break;
            case 2: hostStats.msShade += ms; hostStats.launchesShade++; break;
            case 3: hostStats.msShadow += ms; hostStats.launchesShadow++; hostStats.msShadowByBounce[t.bounce] += ms; break;
            default: hostStats.msAccumulate += ms; hostStats.launchesAccumulate++; break;
            }
            eventPool.push_back(t.start);
            eventPool.push_back(t.stop);
        }
        timed.clear();
    }

    void collectBatchTimings()
    {
        for (const BatchTiming& b : pendingBatches)
        {
            RF_HIP(hipEventSynchronize(b.stop));
            float ms = 0.0f;
            RF_HIP(hipEventElapsedTime(&ms, b.start, b.stop));
            for (uint32_t i = 0; i < b.samples; ++i)
            {
                passDurationsMs.push_back(ms / static_cast<float>(b.samples));
                if (passDurationsMs.size() > 30) passDurationsMs.pop_front();
            }
            eventPool.push_back(b.start);
            eventPool.push_back(b.stop);
        }
        pendingBatches.clear();
    }

    // ---- which record layout a launch reads (kTraceWide's COMPACT parameter), and the launch itself
    struct WideArgs
    {
        PathStreams     ps;
        const uint32_t* queue;
        const uint32_t* count;
        uint32_t*       cursor;
        uint32_t        refillMin, chunk;
        uint32_t        leafVote; // 0: optLeafVote
        float           tMax;
        dim3            grid;
        uint32_t        extraLds;
        uint32_t        blocksWanted = 0; // != 0: the launch's worst-case workgroup count; the grid becomes min(this, what the device holds of the kernel launched)
    };
    // One kTraceWide launch: the instantiation named by (any-hit, counting build, nearest-first, record layout, dense leaf phase) from rf_trace.hip's table; a
    // combination the table does not hold falls back to the same one without the dense leaf phase (options can ask for it on a layout it is not built for)
    // Workgroups of THIS instantiation the device holds at once.  The persistent grid used to be sized once, from the closest-hit kernel (6 per CU: 80 VGPRs, 24 KB of LDS); the
    // any-hit instantiations need 63 ... 72 VGPRs and -- round 6: their stack holds child words only, 4 bytes per entry -- 12 KB: most of them fit 7 per CU.
    uint32_t residentBlocks(TraceWideKernel k, uint32_t extraLds)
    {
        const auto key = std::make_pair(reinterpret_cast<const void*>(k), extraLds);
        const auto it = residentCache.find(key);
        if (it != residentCache.end()) return it->second;
        int perCu = 0;
        RF_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, reinterpret_cast<const void*>(k), kBlock, extraLds));
        const uint32_t n = static_cast<uint32_t>(std::max(perCu, 1)) * std::max(multiProcessors, 1u);
        residentCache[key] = n;
        return n;
    }
    void launchWide(bool anyHit, bool count, bool nearest, int compact, bool dense, const WideScene& w, const WideArgs& a, uint32_t flags)
    {
        TraceWideKernel k = traceWideKernel(anyHit, count, nearest, compact, dense);
        if (k == nullptr && dense) k = traceWideKernel(anyHit, count, nearest, compact, false);
        if (k == nullptr) throw std::logic_error("kTraceWide: record layout " + std::to_string(compact) + " is not compiled into this build");
        dim3 grid = a.grid;
        if (a.blocksWanted != 0u && !wideBlocksForced && !count) grid = dim3(std::min(a.blocksWanted, residentBlocks(k, a.extraLds)));
        hipLaunchKernelGGL(k, grid, dim3(kBlock), a.extraLds, stream, scene, w, sky, sunBasis, a.ps, a.queue, a.count, a.cursor, counters.ptr, a.refillMin, a.leafVote ? a.leafVote : optLeafVote, a.chunk, a.tMax, flags);
    }
    // the layout a test asks for, if this scene has it (else the binary records)
    int layoutIfPresent(int want) const
    {
        switch (want)
        {
        case kLayoutOct: return wide.oct != nullptr ? want : kLayoutBinary;
        case kLayoutQuadLocal: return wide.quadLocal != nullptr ? want : kLayoutBinary;
        case kLayoutQuadHalf: return wide.quadHalf != nullptr ? want : kLayoutBinary;
        case kLayoutQuad: return wide.quad != nullptr ? want : kLayoutBinary;
        case kLayoutHot: return wide.hot != nullptr ? want : kLayoutBinary;
        case kLayoutCompact: return wide.compact != nullptr ? want : kLayoutBinary;
        default: return kLayoutBinary;
        }
    }
    // does a launch with these flags want the dense leaf phase?  (the scene has a leaf as long as the threshold the flags carry: the instantiations that
    // contain the block -- the layouts the renderer picks by itself -- are used only then: its mere presence costs a launch 2.5 %, profiles/r05_leaf)
    bool denseWanted(uint32_t flags) const
    {
        const uint32_t threshold = (flags >> kFlagDenseLeafShift) & 15u;
        return threshold != 0u && maxLeafTriangles >= threshold;
    }
    void launchClosestWide(int layout, bool count, const WideScene& w, const WideArgs& a, uint32_t flags)
    {
        if (count) return launchWide(false, true, false, 0, false, w, a, flags);
        launchWide(false, false, false, layout <= kLayoutOct ? layout : kLayoutBinary, denseWanted(flags), w, a, flags);
    }
    // nearest: entries ordered by slab distance (NEAREST_FIRST); else record order (the conservative layouts' default: optShadowSignOrder)
    void launchShadowWide(int layout, bool nearest, bool count, const WideScene& w, const WideArgs& a, uint32_t flags)
    {
        if (count) return launchWide(true, true, nearest, 0, false, w, a, flags);
        const int compact = (layout <= kLayoutQuadLocal) ? layout : kLayoutBinary;
        launchWide(true, false, (compact == kLayoutCompact || compact == kLayoutHot) ? true : nearest, compact, denseWanted(flags), w, a, flags);
    }
    // The layout the renderer picks BY ITSELF for the closest-hit / any-hit launch of a bounce (the per-scene defaults set at upload + the options; reported by
    // rf_renderer_layout_info and used by traceBatch): kLayoutScalar = the one-ray-per-thread kernels (trees the packed tests cannot serve), kLayoutPacket = kTracePacket
    int closestLayoutFor(uint32_t bounce) const
    {
        if (traversalVariant == 0) return kLayoutScalar;
        if (counting) return kLayoutBinary;
#if defined(RF_EXP_LEGACY_LAYOUTS)
        if (bounce <= optPacketBounces) return kLayoutPacket;
#endif
        const bool  quadNow = wide.quad != nullptr && optQuadFromBounce != 0u && bounce >= optQuadFromBounce && !((optQuadExceptMask >> std::min(bounce - 1u, 31u)) & 1u);
        // A camera that stands outside the conservative records' origin bound (4 R + 1 for a root box within +-R: a turntable shot from far away) would
        // send EVERY primary ray to the scalar traversal (2 x the launch, tools/gpu_far_camera.py): that launch reads the exact quad records, which have
        // no such bound.  Later bounces start on surfaces, inside the bound.
        const float camReach = std::max({std::fabs(params.camera.origin.x), std::fabs(params.camera.origin.y), std::fabs(params.camera.origin.z)}) + (std::fabs(params.camera.lensRadius) * 2.0f);
        const bool  primaryOutside = bounce == 1u && !(camReach <= wide.originBound);
        if (quadNow && !primaryOutside)
        {
            if (wide.oct != nullptr && optOctFromBounce != 0u && bounce >= optOctFromBounce) return kLayoutOct;
            if (wide.quadHalf != nullptr && optQuadHalfFromBounce != 0u && bounce >= optQuadHalfFromBounce) return kLayoutQuadHalf;
            if (wide.quadLocal != nullptr && optQuadLocalFromBounce != 0u && bounce >= optQuadLocalFromBounce) return kLayoutQuadLocal;
        }
        if (quadNow) return kLayoutQuad;
        if (wide.hot != nullptr && optHotFromBounce != 0u && bounce >= optHotFromBounce) return kLayoutHot;
        if (wide.compact != nullptr && optCompactFromBounce != 0u && bounce >= optCompactFromBounce) return kLayoutCompact;
        return kLayoutBinary;
    }
    int shadowLayoutFor(uint32_t bounce) const
    {
        if (traversalVariant == 0) return kLayoutScalar;
#if defined(RF_EXP_LEGACY_LAYOUTS)
        if (!counting && bounce <= optPacketBounces) return kLayoutPacket;
#endif
        if (counting || !shadowNearestFirst) return kLayoutBinary;
        const bool quadShadowNow = optQuadShadowFromBounce != 0u && bounce >= optQuadShadowFromBounce && !((optQuadShadowExceptMask >> std::min(bounce - 1u, 31u)) & 1u);
        if (quadShadowNow)
        {
            if (wide.quadLocal != nullptr && optQuadLocalShadowFromBounce != 0u && bounce >= optQuadLocalShadowFromBounce) return kLayoutQuadLocal;
            if (wide.quadHalf != nullptr && optQuadHalfShadowFromBounce != 0u && bounce >= optQuadHalfShadowFromBounce) return kLayoutQuadHalf;
            if (wide.quad != nullptr) return kLayoutQuad;
        }
        if (wide.hot != nullptr && optHotShadowFromBounce != 0u && bounce >= optHotShadowFromBounce) return kLayoutHot;
        if (wide.compact != nullptr && optCompactShadowFromBounce != 0u && bounce >= optCompactShadowFromBounce) return kLayoutCompact;
        return kLayoutBinary;
    }
    // does the any-hit launch of this bounce start at the occluder cache's entries? (kTraceWide, kFlagOccluderCache)
    bool cachedShadowFor(uint32_t bounce) const
    {
        const int layout = shadowLayoutFor(bounce);
        const bool conservative = layout == kLayoutQuadLocal || layout == kLayoutQuadHalf;
        const bool exactQuad = layout == kLayoutQuad && leafBoxesValid; // (its leaf visits then apply the box in the leaf's triangle record)
        return !counting && shadowNearestFirst && (conservative || exactQuad) && bounce <= optOccluderCacheBounces && optOccluderGridCells != 0u;
    }

    // Test hook (option query_variant = 2): arbitrary rays through the render path's persistent
    // traversal kernel.  closest: out0 = hit stream {tri, u, v, t}, out1 = rayO stream (offset hit
    // point); shadow: out0 = rad stream, .x != 0 iff the ray is unoccluded.
    void queryWide(const float* rays6, uint64_t n, float tMax, bool shadow, std::vector<float4>& out0, std::vector<float4>& out1)
    {
        if (n > 0xFFFFFFFFull) throw std::runtime_error("too many rays");
        const uint32_t denseFlag = std::min(optDenseLeafMin, 15u) << kFlagDenseLeafShift; // (the dense leaf phase, as in the render path)
        if (!ensurePathState(n)) throw std::runtime_error("out of device memory for the ray batch");
        std::vector<P3>       o(n), d(n);
        std::vector<uint32_t> ids(n);
        for (uint64_t i = 0; i < n; ++i)
        {
            o[i] = P3{rays6[6 * i], rays6[6 * i + 1], rays6[6 * i + 2]};
            d[i] = P3{rays6[6 * i + 3], rays6[6 * i + 4], rays6[6 * i + 5]};
            ids[i] = static_cast<uint32_t>(i);
        }
        // every copy goes through the handle's stream: ordered behind a batch that may still be in flight there
        RF_HIP(hipStreamSynchronize(stream));
        RF_HIP(hipMemcpyAsync(sRayO.ptr, o.data(), n * sizeof(P3), hipMemcpyHostToDevice, stream));
        RF_HIP(hipMemcpyAsync(sRayD.ptr, d.data(), n * sizeof(P3), hipMemcpyHostToDevice, stream));
        RF_HIP(hipMemcpyAsync(queueA.ptr, ids.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
        const uint32_t words = kLineWords * (1 + kShards);
        if (queueCounts.count < words) queueCounts.alloc(words);
        RF_HIP(hipMemsetAsync(queueCounts.ptr, 0, queueCounts.count * sizeof(uint32_t), stream));
        const uint32_t count = static_cast<uint32_t>(n);
        RF_HIP(hipMemcpyAsync(queueCounts.ptr, &count, sizeof count, hipMemcpyHostToDevice, stream));
        PathStreams ps{sRayO.ptr, sRayD.ptr, sThr.ptr, sRad.ptr, sHit.ptr, sPending.ptr, sNoise.ptr, sRayD2.ptr, sThr2.ptr, sNoise2.ptr};
        const dim3  grid(std::min<uint32_t>(static_cast<uint32_t>((n + kBlock - 1) / kBlock), wideBlocks));
        const WideArgs wa{ps, queueA.ptr, queueCounts.ptr, queueCounts.ptr + kLineWords, optRefillMin, optChunk, 0u, tMax, grid, 0u};
        if (shadow)
        {
            // rad = 0, pending = 1: rad.x becomes visibility * SOLAR_INV_PDF
            std::vector<P3> ones(n, P3{1.0f, 1.0f, 1.0f});
            RF_HIP(hipMemcpyAsync(sPending.ptr, ones.data(), n * sizeof(P3), hipMemcpyHostToDevice, stream));
            RF_HIP(hipMemsetAsync(sRad.ptr, 0, n * sizeof(float4), stream));
            RF_HIP(hipStreamSynchronize(stream)); // `ones` leaves scope before the launches are waited for
            // (the ray-query entry points take the layout the test asks for -- query_compact -- where the scene has it; the oct records serve closest-hit rays only)
            const int layout = layoutIfPresent((optQueryCompact == kLayoutOct || (!shadowNearestFirst && optQueryCompact < kLayoutQuad)) ? kLayoutBinary : optQueryCompact);
            launchShadowWide(layout, shadowNearestFirst, false, wide, wa, kFlagShadowDirFromStream | denseFlag);
        }
        else
        {
            // RF_DEBUG_QUERY_LIST=<file of n u32>: (read and uploaded BEFORE the timed span) the launch visits the rays in the order of that list (queue POSITIONS, as the any-hit launches behind kShadowFirstLook do) while
            // the rays stay where they are -- what a global order of a bounce's rays costs when only an index list is sorted (tools/gpu_sort_potential.py --indirect)
            WideScene wq = wide;
            wq.rayList = nullptr;
            if (const char* listPath = std::getenv("RF_DEBUG_QUERY_LIST"))
            {
                std::vector<uint32_t> list(n);
                FILE* f = std::fopen(listPath, "rb");
                if (f == nullptr || std::fread(list.data(), sizeof(uint32_t), n, f) != n) throw std::runtime_error("RF_DEBUG_QUERY_LIST: cannot read the ray list");
                std::fclose(f);
                RF_HIP(hipMemcpyAsync(queueB.ptr, list.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
                RF_HIP(hipStreamSynchronize(stream));
                wq.rayList = queueB.ptr;
            }
            // RF_DEBUG_QUERY_MS: the traversal launch alone between two events (tools/gpu_sort_potential.py: what would an order of the rays be worth?)
            const bool timed = std::getenv("RF_DEBUG_QUERY_MS") != nullptr;
            hipEvent_t e0 = nullptr, e1 = nullptr;
            if (timed)
            {
                RF_HIP(hipEventCreate(&e0));
                RF_HIP(hipEventCreate(&e1));
                RF_HIP(hipEventRecord(e0, stream));
            }
            launchClosestWide(layoutIfPresent(optQueryCompact), false, wq, wa, denseFlag);
            if (timed) RF_HIP(hipEventRecord(e1, stream));
            hipLaunchKernelGGL(hitPointsKernel(), dim3((count + 255) / 256), dim3(256), 0, stream, scene, sHit.ptr, sRayO.ptr, count);
            if (timed)
            {
                RF_HIP(hipStreamSynchronize(stream));
                float ms = 0.0f;
                RF_HIP(hipEventElapsedTime(&ms, e0, e1));
                std::fprintf(stderr, "[rf-query] closest-hit launch: %u rays, %.4f ms\n", count, ms);
                RF_HIP(hipEventDestroy(e0));
                RF_HIP(hipEventDestroy(e1));
            }
        }
        RF_HIP(hipGetLastError());
        RF_HIP(hipStreamSynchronize(stream));
        out0.resize(n);
        RF_HIP(hipMemcpy(out0.data(), shadow ? sRad.ptr : sHit.ptr, n * sizeof(float4), hipMemcpyDeviceToHost));
        if (!shadow)
        {
            std::vector<P3> packed(n);
            RF_HIP(hipMemcpy(packed.data(), sRayO.ptr, n * sizeof(P3), hipMemcpyDeviceToHost));
            out1.resize(n);
            for (uint64_t i = 0; i < n; ++i) out1[i] = make_float4(packed[i].x, packed[i].y, packed[i].z, 0.0f);
        }
    }

    // Trace `numSamples` consecutive samples (sample indices start at frame `firstFrame`).
    void traceBatch(uint32_t firstFrame, uint32_t numSamples)
    {
        FrameParams fp{};
        fp.width = params.width;
        fp.height = params.height;
        fp.camera = params.camera;
        fp.samplesPerPixel = params.samplingParams.numSamplesPerPixel;
        fp.numBounces = params.samplingParams.numBounces;
        fp.firstFrame = firstFrame;
        fp.numSamples = numSamples;
        fp.numTiles = static_cast<uint32_t>(tiles.size());
        fp.pixelsPadded = fp.numTiles * 1024u;
        fp.slotGroupShift = optSlotGroupShift;
        fp.samplePerm = fp.sampleInvPerm = nullptr;
        // (kSamplePermutation ranks by counting, O(S^2): fine for the few hundred samples a batch of a real frame holds,
        // skipped for the huge sample counts a tiny frame can put into one batch)
        if (optSampleSort && numSamples > 1 && numSamples <= 8192 && optSlotGroupShift != kSlotSampleMajor)
        {
            if (samplePerm.count < 2ull * numSamples) samplePerm.alloc(2ull * numSamples); // (the stream is idle the first time; later batches are no larger)
            fp.samplePerm = samplePerm.ptr;
            fp.sampleInvPerm = samplePerm.ptr + numSamples;
        }
        fp.tilesX = (params.width + kTileSize - 1) / kTileSize;
        if (fp.numTiles == 0) return;
        fp.divNumSamples = FastDiv::make(fp.numSamples), fp.divPixelsPadded = FastDiv::make(fp.pixelsPadded), fp.divTilesX = FastDiv::make(fp.tilesX);
        fp.divSamplesPerPixel = FastDiv::make(fp.samplesPerPixel);
        // Pinhole camera (lensRadius == 0): kRaygen's origin = camera.origin + (0 * right + 0 * up) is camera.origin itself for every path, bit for bit, as long as no
        // component of camera.origin is a zero (whose sign the +-0 addend could flip) and right / up are finite (0 * inf = NaN).  The primary launch then takes it as a kernel
        // argument and kRaygen writes 28 instead of 40 bytes per path.  Only the wide traversal kernels know the flag (the scalar kernels read ps.rayO).
        const auto finite3 = [](Vec3 v) { return std::isfinite(v.x) && std::isfinite(v.y) && std::isfinite(v.z); };
        const Camera& camNow = params.camera;
        const int     primaryLayout = closestLayoutFor(1);
        const bool    constOrigin = optConstPrimaryOrigin && camNow.lensRadius == 0.0f && camNow.origin.x != 0.0f && camNow.origin.y != 0.0f && camNow.origin.z != 0.0f &&
                                 finite3(camNow.origin) && finite3(camNow.right) && finite3(camNow.up) && primaryLayout != kLayoutScalar && primaryLayout != kLayoutPacket;
        fp.skipOrigins = constOrigin ? 1u : 0u;
        // (kRaygen computes its queue positions instead of appending with one atomic per 1 024 slots, where the slot order allows it: FrameParams::tileValidBefore)
        const bool denseRaygen = optDenseRaygen && (fp.slotGroupShift == 0u || fp.slotGroupShift == kSlotSampleMajor) && static_cast<uint64_t>(validPixels) * numSamples <= 0xFFFFFFFFull;
        fp.tileValidBefore = denseRaygen ? tileValidBefore.ptr : nullptr;
        fp.validPixels = static_cast<uint32_t>(validPixels);
        wide.constOriginX = camNow.origin.x, wide.constOriginY = camNow.origin.y, wide.constOriginZ = camNow.origin.z;

        const uint64_t paths = static_cast<uint64_t>(numSamples) * fp.pixelsPadded;
        const uint32_t blocks = static_cast<uint32_t>((paths + kBlock - 1) / kBlock);
        primaryRaysHost += static_cast<unsigned long long>(numSamples) * validPixels;
        // bounce b reads direction / throughput from buffer (b - 1) & 1 and kShade writes the next bounce's into the other one
        PathStreams    ps{sRayO.ptr, sRayD.ptr, sThr.ptr, sRad.ptr, sHit.ptr, sPending.ptr, sNoise.ptr, sRayD2.ptr, sThr2.ptr, sNoise2.ptr};
        const uint32_t numBounces = fp.numBounces;

        wide.occGrid = nullptr;
        wide.rayList = nullptr;
        if (optOccluderCacheBounces != 0u && optOccluderGridCells != 0u)
        {
            const size_t kOccluderGridEntries = (size_t{1} << optOccluderGridLog2Cells) * kOccSlots;
            if (occluderGrid.count != kOccluderGridEntries)
            {
                occluderGrid.alloc(kOccluderGridEntries);
                occluderGridWarm = false;
                RF_HIP(hipMemsetAsync(occluderGrid.ptr, 0, kOccluderGridEntries * sizeof(uint32_t), stream));
            }
            const float extent = std::max({wide.rootHi.x - wide.rootLo.x, wide.rootHi.y - wide.rootLo.y, wide.rootHi.z - wide.rootLo.z, 1e-20f});
            wide.occGrid = occluderGrid.ptr;
            wide.occScale = static_cast<float>(optOccluderGridCells) / extent;
            wide.occMask = static_cast<uint32_t>(kOccluderGridEntries / kOccSlots - 1);
        }
        if (lookBatch.count == 0)
        {
            lookBatch.alloc(2);
            RF_HIP(hipHostMalloc(reinterpret_cast<void**>(&lookBatchHost), 2 * sizeof(unsigned long long)));
            RF_HIP(hipEventCreateWithFlags(&lookEvent, hipEventDisableTiming));
        }
        if (lookPending && hipEventQuery(lookEvent) == hipSuccess)
        {
            lookPending = false;
            if (lookBatchHost[1] != 0ull && lookBatchHost[0] * 4ull < lookBatchHost[1]) firstLookHoldOff = 16u;
        }
        else if (firstLookHoldOff != 0u) --firstLookHoldOff;
        RF_HIP(hipMemsetAsync(lookBatch.ptr, 0, 2 * sizeof(unsigned long long), stream));
        BatchTiming bt{getEvent(), getEvent(), numSamples};
        RF_HIP(hipEventRecord(bt.start, stream));

        // device words, one per 64-byte line (they are all hot atomics): [0, B]: queue lengths per
        // bounce; [B+1, 2B]: miss-list length per bounce; then two work cursors per bounce for the traversal launches
        constexpr uint32_t kLine = kLineWords;
        const uint32_t words = kLine * (2 * numBounces + 1) + kLine * kShards * 2 * numBounces + kLine * numBounces + kLine * numBounces;
        if (queueCounts.count < words) queueCounts.alloc(words);
        uint32_t* const missCounts = queueCounts.ptr + kLine * (numBounces + 1);
        uint32_t* const cursors = queueCounts.ptr + kLine * (2 * numBounces + 1);
        uint32_t* const listCounts = cursors + kLine * kShards * 2 * numBounces; // lengths of the lists kShadowFirstLook leaves to the any-hit launches, per bounce
        uint32_t* const shadowListCounts = listCounts + kLine * numBounces; // lengths of kShade's lists of shadow rays still to trace (kShadeSelfShadow), per bounce
        unsigned long long lookMask = 0ull, selfMask = 0ull;
        const uint32_t  itemBlocks = static_cast<uint32_t>((paths + kBlock * kItems - 1) / (kBlock * kItems));
        RF_HIP(hipMemsetAsync(queueCounts.ptr, 0, queueCounts.count * sizeof(uint32_t), stream));

        uint32_t* qIn = queueA.ptr;
        uint32_t* qOut = queueB.ptr;
        if (fp.samplePerm)
            hipLaunchKernelGGL(samplePermutationKernel(), dim3((numSamples + 255) / 256), dim3(256), 0, stream, firstFrame, fp.samplesPerPixel, numSamples,
                               const_cast<uint32_t*>(fp.samplePerm), const_cast<uint32_t*>(fp.sampleInvPerm));
        launchTimed(0, [&] {
            hipLaunchKernelGGL(raygenKernel(optTranscendentalsF32), dim3(itemBlocks), dim3(kBlock), 0, stream, fp, scene, tileIds.ptr, ps, qIn, queueCounts.ptr, counters.ptr);
        });
        const dim3 persistentGrid(std::min(blocks, wideBlocks));
        for (uint32_t bounce = 1; bounce <= numBounces; ++bounce)
        {
            uint32_t* countIn = queueCounts.ptr + kLine * (bounce - 1);
            uint32_t* countOut = queueCounts.ptr + kLine * bounce;
            uint32_t* cursorClosest = cursors + kLine * kShards * 2 * (bounce - 1);
            uint32_t* cursorShadow = cursorClosest + kLine * kShards;
            const uint32_t uniformFlag = ((optUniformFetch < 0 ? bounce <= 2 : optUniformFetch > 0) ? kFlagUniformFetch : 0u) | (optUniformFetch >= 2 ? kFlagUniformTri : 0u) |
                                         (bounce >= optDenseLeafFromBounce ? (std::min(optDenseLeafMin, 15u) << kFlagDenseLeafShift) : 0u);
            // incoherent closest-hit launches refill earlier: their rays differ most in length, so lanes go idle sooner (per-bounce
            // sweep, profiles/r02_final/bounce_sweep*.log: bounces 3-8 -4 % at 20-24 idle lanes, bounces 1-2 and the shadow launches +6 .. +10 %)
            // ... and the coherent launches of the first bounces, whose rays are short and alike, claim larger chunks (one cursor atomic = one
            // wave-wide stall: bounce 1 -4 % at 256 entries, the deep bounces +0.5 %)
            const uint32_t chunkNow = bounce <= optChunkEarlyBounces ? optChunkEarly : optChunk;
            const int      layoutClosest = closestLayoutFor(bounce);
            const bool     conservativeClosest = layoutClosest == kLayoutOct || layoutClosest == kLayoutQuadHalf || layoutClosest == kLayoutQuadLocal;
            // (scenes with long leaves -- the DENSE_LEAVES instantiations -- keep 22: a phase over dense (lane, triangle) pairs wants many parked lanes; clutter atrium +2.4 % at 12: profiles/r06_lanes)
            const uint32_t refillClosest = bounce >= optRefillDeepFromBounce ? (layoutClosest == kLayoutQuad ? optRefillMinDeepQuad : (denseWanted(uniformFlag) ? optRefillMinDeepDense : optRefillMinDeep)) : optRefillMin;
            (void)conservativeClosest;
            // (kInfinityCacheBytes: the scene's records + triangles against the 256-MB Infinity Cache, as the layout selector's own test)
            const uint32_t leafVoteClosest = (!counting && !denseWanted(uniformFlag) && treeBytes <= (192ull << 20)) ? optLeafVoteClosest : 0u;
            launchTimed(1, [&] {
                if (layoutClosest == kLayoutScalar)
                {
                    if (counting)
                        hipLaunchKernelGGL(traceClosestKernel(true), dim3(blocks), dim3(kBlock), 0, stream, scene, ps, qIn, countIn, counters.ptr);
                    else
                        hipLaunchKernelGGL(traceClosestKernel(false), dim3(blocks), dim3(kBlock), 0, stream, scene, ps, qIn, countIn, counters.ptr);
                }
#if defined(RF_EXP_LEGACY_LAYOUTS)
                else if (layoutClosest == kLayoutPacket)
                    hipLaunchKernelGGL(tracePacketKernel(false), persistentGrid, dim3(kBlock), 0, stream, scene, wide, sky, sunBasis, ps, qIn, countIn, counters.ptr, kTMax, 0u);
#endif
                else
                    launchClosestWide(layoutClosest, counting, wide, WideArgs{ps, qIn, countIn, cursorClosest, counting ? optRefillMin : refillClosest, chunkNow, leafVoteClosest, kTMax, persistentGrid, counting ? 0u : optExtraLds, blocks},
                                      uniformFlag | (bounce == 1 && constOrigin ? kFlagConstOrigin : 0u));
            }, bounce - 1);
            uint32_t* const missCount = missCounts + kLine * (bounce - 1);
            // kShade's own-triangle test of the shadow rays (kShadeSelfShadow): wherever this bounce's any-hit launch is one that can work through a list of queue positions
            // (the kTraceWide instantiations with the occluder-cache code: quad / half / local-grid records) and nothing has to be counted node by node
            const int      layoutShadow = shadowLayoutFor(bounce);
            // (bounce <= 64: kBounceTotals keeps one mask bit per bounce -- as for firstLook below; deeper bounces trace every shadow ray and count them in the launch: ADVICE r5)
            const bool     selfShadow = optShadowSelfTest && selfShadowOk && !counting && bounce <= 64u && (layoutShadow == kLayoutQuad || layoutShadow == kLayoutQuadHalf || layoutShadow == kLayoutQuadLocal);
            uint32_t* const shadowListCount = shadowListCounts + kLine * (bounce - 1);
            if (selfShadow) selfMask |= 1ull << (bounce - 1);
            launchTimed(2, [&] {
                const uint32_t shadeFlags = (bounce == numBounces ? kShadeLastBounce : 0u) | (bounce == 1 ? kShadeFirstBounce : 0u) | (selfShadow ? kShadeSelfShadow : 0u);
                const dim3     shadeGrid(optShadeBlocks ? std::min(itemBlocks, optShadeBlocks) : itemBlocks);
                if (optShadeSortFromBounce != 0u && bounce >= optShadeSortFromBounce)
                    hipLaunchKernelGGL(shadeKernel(true), shadeGrid, dim3(kBlock), 0, stream, scene, sky, sunBasis, ps, qIn, countIn, qOut, countOut, missQueue.ptr, missSlots.ptr, missCount, shadowList.ptr, shadowListCount, shadeFlags, sortScale);
                else
                    hipLaunchKernelGGL(shadeKernel(false), shadeGrid, dim3(kBlock), 0, stream, scene, sky, sunBasis, ps, qIn, countIn, qOut, countOut, missQueue.ptr, missSlots.ptr, missCount, shadowList.ptr, shadowListCount, shadeFlags, 0u);
                // the paths that left the scene at this bounce, while its direction / throughput arrays are intact
                // (its slot comes with the miss list: kSky reads nothing of the bounce's queue -- round 5: reading the slot through the queue position was a third
                // dependent gather, and kShade + kSky went from 19.7 to 15.8 ms per 64 spp without it)
                hipLaunchKernelGGL(skyKernel(optTranscendentalsF32), dim3(std::min(blocks, skyBlocks)), dim3(kBlock), 0, stream, sky, ps, missSlots.ptr, missQueue.ptr, missCount, bounce == 1 ? 1u : 0u);
            });
            // occluder cache (kTraceWide, kFlagOccluderCache): the conservative-record any-hit launches of bounces 1..optOccluderCacheBounces; their rays are
            // short (a third of the steps), so the deep launches refill earlier
            const bool     cachedShadow = layoutShadow != kLayoutScalar && layoutShadow != kLayoutPacket && cachedShadowFor(bounce);
            // ... behind kShadowFirstLook (see there) from the second batch on: the first batch of a renderer fills the grid (the traversal kernel's own first look serves)
            const bool      firstLook = cachedShadow && occluderHintLevels == 0u && optShadowFirstLookFromBounce != 0u && bounce >= optShadowFirstLookFromBounce && bounce <= 64u && occluderGridWarm && firstLookHoldOff == 0u;
            uint32_t* const listCount = listCounts + kLine * (bounce - 1);
            uint32_t* const countShadow = firstLook ? listCount : (selfShadow ? shadowListCount : countOut);
            if (firstLook) lookMask |= 1ull << (bounce - 1);
            const uint32_t shadowFlags = (bounce == 1 ? kFlagFirstBounce : 0u) | uniformFlag | (cachedShadow ? kFlagOccluderCache : 0u) | (firstLook ? (kFlagOccluderNoTry | kFlagNoRayCount) : 0u) | (selfShadow ? kFlagNoRayCount : 0u);
            launchTimed(3, [&] {
                WideScene wide = this->wide; // (the launches below name `wide`)
                if (firstLook)
                {
                    // (the bounce's input queue is free by now -- kShade and kSky have consumed it -- and holds the list)
                    const dim3 lookGrid(optShadeBlocks ? std::min(itemBlocks, optShadeBlocks) : itemBlocks);
                    hipLaunchKernelGGL(shadowFirstLookKernel(), lookGrid, dim3(kBlock), 0, stream, scene, wide, sky, sunBasis, ps, qOut, selfShadow ? shadowList.ptr : nullptr,
                                       selfShadow ? shadowListCount : countOut, qIn, listCount, counters.ptr, kTMax, (bounce == 1 ? kLookFirstBounce : 0u) | (selfShadow ? kLookNoRayCount : 0u));
                    wide.rayList = qIn;
                }
                else if (selfShadow) wide.rayList = shadowList.ptr; // (positions only: bit 31 clear, the traversal kernel takes its own first look)
                if (layoutShadow == kLayoutScalar)
                {
                    if (counting)
                        hipLaunchKernelGGL(traceShadowKernel(true), dim3(blocks), dim3(kBlock), 0, stream, scene, sky, sunBasis, ps, qOut, countOut, counters.ptr, bounce == 1 ? 1u : 0u);
                    else
                        hipLaunchKernelGGL(traceShadowKernel(false), dim3(blocks), dim3(kBlock), 0, stream, scene, sky, sunBasis, ps, qOut, countOut, counters.ptr, bounce == 1 ? 1u : 0u);
                }
#if defined(RF_EXP_LEGACY_LAYOUTS)
                else if (layoutShadow == kLayoutPacket)
                    hipLaunchKernelGGL(tracePacketKernel(true), persistentGrid, dim3(kBlock), 0, stream, scene, wide, sky, sunBasis, ps, qOut, countOut, counters.ptr, kTMax,
                                       shadowFlags & kFlagFirstBounce);
#endif
                else
                {
                    // the conservative layouts (VALU bound) visit a record's entries in record order unless asked otherwise; the exact and binary records nearest-first
                    const bool conservative = layoutShadow == kLayoutQuadLocal || layoutShadow == kLayoutQuadHalf;
                    const bool nearest = shadowNearestFirst && !(conservative && optShadowSignOrder);
                    launchShadowWide(layoutShadow, nearest, counting, wide, WideArgs{ps, qOut, countShadow, cursorShadow, optRefillMin, chunkNow, 0u, kTMax, persistentGrid, counting ? 0u : optExtraLds, blocks}, shadowFlags);
                }
            }, bounce - 1);
            std::swap(qIn, qOut);
            std::swap(ps.rayD, ps.rayDOut);
            std::swap(ps.thr, ps.thrOut);
            std::swap(ps.noise, ps.noiseOut);
        }
        hipLaunchKernelGGL(bounceTotalsKernel(), dim3(1), dim3(64), 0, stream, queueCounts.ptr, std::min(numBounces, 64u), bounceTotals.ptr, listCounts, lookMask, lookBatch.ptr, shadowListCounts, selfMask, counters.ptr);
        if (lookMask != 0ull && lookBatchHost != nullptr)
        {
            RF_HIP(hipMemcpyAsync(lookBatchHost, lookBatch.ptr, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
            RF_HIP(hipEventRecord(lookEvent, stream));
            lookPending = true;
        }
        if (wide.occGrid != nullptr) occluderGridWarm = true;
        launchTimed(4, [&] {
            if (fp.slotGroupShift == 0u && numSamples > 4u && numSamples <= kAccMaxSamples && optAccumulateRuns)
            {
                // (pixels per workgroup by the LDS their runs take: <= ~8 KB per workgroup keeps twenty of them resident per CU)
                const uint32_t accPixels = numSamples > 640u ? 1u : (numSamples > 160u ? 2u : kAccPixels);
                hipLaunchKernelGGL(accumulateRunsKernel(accPixels), dim3((fp.pixelsPadded + accPixels - 1) / accPixels), dim3(64), accPixels * 3u * (numSamples + 1u) * sizeof(float), stream, fp,
                                   tileIds.ptr, ps, image);
            }
            else
                hipLaunchKernelGGL(accumulateKernel(), dim3((fp.pixelsPadded + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, fp, tileIds.ptr, ps, image);
        });
        RF_HIP(hipGetLastError());
        RF_HIP(hipEventRecord(bt.stop, stream));
        pendingBatches.push_back(bt);
        if (timing) collectTimings();
        if (pendingBatches.size() > 64) collectBatchTimings();
    }
};

Renderer::Renderer(const RendererDescriptor& desc, const SceneView& sceneView) : mImpl(std::make_unique<Impl>())
{
    Impl& m = *mImpl;
    int   deviceCount = 0;
    if (hipGetDeviceCount(&deviceCount) != hipSuccess || deviceCount == 0)
        throw std::runtime_error("rayfinder_amd: no HIP device available (this library has no CPU fallback)");
    m.device = desc.deviceOrdinal;
    RF_HIP(hipSetDevice(m.device));
    RF_HIP(hipStreamCreateWithFlags(&m.stream, hipStreamNonBlocking));

    if (sceneView.bvhNodes.empty()) throw std::runtime_error("scene has no BVH nodes");
    if (sceneView.positionAttributes.size() != sceneView.vertexAttributes.size())
        throw std::runtime_error("position and vertex attribute counts differ");
    // child links, leaf ranges and texture indices are followed blindly on the device: check them once here
    validateScene(sceneView.bvhNodes, sceneView.positionAttributes.size(), sceneView.vertexAttributes, sceneView.baseColorTextures.size());
    m.sortScale = static_cast<uint32_t>(std::min<uint64_t>((static_cast<uint64_t>(kSortBins) << 32) / std::max<uint64_t>(sceneView.positionAttributes.size(), 1), 0xFFFFFFFFull));

    // 48-B reference nodes -> 32-B device nodes
    bool                  treeNestedRegular = false; // every child's box inside its parent's, all finite and ordered (buildWide)
    std::vector<uint32_t> quadIndexOfNode; // buildWide's numbering of the quad records, for the occluder-cache entries of the leaves (leafBoxesIntoTriangles)
    {
        std::vector<float4> packed(2 * sceneView.bvhNodes.size());
        for (size_t i = 0; i < sceneView.bvhNodes.size(); ++i)
        {
            const BvhNode& n = sceneView.bvhNodes[i];
            const bool     leaf = n.triangleCount > 0;
            if (n.triangleCount >= (1u << 30)) throw std::runtime_error("BVH leaf too large");
            const uint32_t link = leaf ? n.trianglesOffset : n.secondChildOffset;
            const uint32_t meta = leaf ? ((n.triangleCount << 2) | kLeafAxis) : (n.splitAxis & 3u);
            if (!leaf && n.splitAxis > 2) throw std::runtime_error("interior BVH node with invalid split axis");
            packed[2 * i] = make_float4(n.aabb.min.x, n.aabb.min.y, n.aabb.min.z, bitsFloat(link));
            packed[2 * i + 1] = make_float4(n.aabb.max.x, n.aabb.max.y, n.aabb.max.z, bitsFloat(meta));
        }
        m.nodes.upload(packed.data(), packed.size());
        const WideBuild wb = buildWide(sceneView.bvhNodes.data(), sceneView.bvhNodes.size());
        quadIndexOfNode = wb.quadIndexOfNode;
        treeNestedRegular = wb.boxesNested && wb.boxesRegular;
        m.wideNodes.upload(wb.nodes.data(), wb.nodes.size());
        m.bigLeaves.upload(wb.bigLeaves.data(), wb.bigLeaves.size());
        m.wide.nodes = m.wideNodes.ptr;
        m.wide.compact = nullptr;
#if defined(RF_EXP_LEGACY_LAYOUTS)
        if (wb.compactUsable && !wb.compact.empty())
        {
            m.wideCompact.upload(wb.compact.data(), wb.compact.size());
            m.wide.compact = m.wideCompact.ptr;
        }
#endif
        m.wide.hot = m.wide.own = nullptr;
#if defined(RF_EXP_LEGACY_LAYOUTS)
        if (wb.hotUsable && !wb.hot.empty())
        {
            m.wideHot.upload(wb.hot.data(), wb.hot.size());
            m.wideOwn.upload(wb.own.data(), wb.own.size());
            m.wide.hot = m.wideHot.ptr;
            m.wide.own = m.wideOwn.ptr;
        }
#endif
        m.wide.quad = nullptr;
        if (wb.quadUsable && !wb.quad.empty())
        {
            m.wideQuad.upload(wb.quad.data(), wb.quad.size());
            m.wide.quad = m.wideQuad.ptr;
        }
        m.wide.quadHalf = nullptr;
        m.wide.originBound = wb.originBound;
        if (!wb.quadHalf.empty())
        {
            m.wideQuadHalf.upload(wb.quadHalf.data(), wb.quadHalf.size());
            m.wide.quadHalf = m.wideQuadHalf.ptr;
            // default (rf_wide.hpp, kQuadHalfMaxAreaRatio): the closest-hit launches read the half-precision records unless the binary16
            // grid is too coarse for this scene; the shadow launches prefer the local-grid records (below)
            m.optQuadHalfFromBounce = m.optQuadHalfShadowFromBounce = wb.quadHalfAreaRatio <= kQuadHalfMaxAreaRatio ? 1u : 0u;
            m.quadHalfAreaRatio = wb.quadHalfAreaRatio;
        }
        m.wide.quadLocal = nullptr;
        if (!wb.quadLocal.empty())
        {
            m.wideQuadLocal.upload(wb.quadLocal.data(), wb.quadLocal.size());
            m.wide.quadLocal = m.wideQuadLocal.ptr;
            // closest-hit: the per-record 8-bit grid where binary16 of absolute coordinates is too coarse (small triangles far from the
            // origin); shadow: the local grid unless the scene is so finely tessellated that the exact records win (kQuadLocalShadowMaxAreaRatio)
            const float ratio = wb.quadHalf.empty() ? 2.0f : wb.quadHalfAreaRatio;
            // (from bounce 2: the coherent launches of bounce 1 gain nothing from it and lose 5 - 25 % in the larger scenes; the shadow launches
            // of bounce 1 stay on the half-precision records where those suit the scene, on the exact ones elsewhere)
            m.optQuadLocalFromBounce = m.optQuadHalfFromBounce == 0u ? 2u : 0u;
            m.optQuadLocalShadowFromBounce = ratio <= kQuadLocalShadowMaxAreaRatio ? 2u : 0u;
        }
        m.wide.oct = nullptr;
        m.treeBytes = static_cast<uint64_t>(wb.quadLocal.size()) * sizeof(uint4) + static_cast<uint64_t>(sceneView.positionAttributes.size()) * kTriStride * sizeof(float4);
        if (!wb.oct.empty())
        {
            m.wideOct.upload(wb.oct.data(), wb.oct.size());
            m.wide.oct = m.wideOct.ptr;
            // NOT selected by default: measured 17 % SLOWER than the local-grid quad records on the out-of-cache atrium (profiles/r05_hbm: the launch is bound by
            // L1 -> L2 requests, and a 112-byte record is two of them per step for 0.65 x the steps); option oct_from_bounce turns it on
            m.optOctFromBounce = 0u;
        }
        m.wide.bigLeaves = m.bigLeaves.ptr;
        m.wide.rootLo = wb.rootLo;
        m.wide.rootHi = wb.rootHi;
        m.wide.rootLeaf = wb.rootLeaf;
        m.wide.numRecords = static_cast<uint32_t>(wb.nodes.size() / 4);
        m.wideUsable = wb.boxesRegular; // NaN / inverted boxes: only the reference-ordered scalar kernels are exact
    }
    {
        // 48-B PositionAttribute -> 64-B aligned device triangles (one L2 sector per triangle test)
        const size_t        n = sceneView.positionAttributes.size();
        std::vector<float4> padded(kTriStride * n, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
        for (size_t i = 0; i < n; ++i)
        {
            const PositionAttribute& t = sceneView.positionAttributes[i];
            padded[kTriStride * i] = make_float4(t.p0.x, t.p0.y, t.p0.z, 0.0f);
            padded[kTriStride * i + 1] = make_float4(t.p1.x, t.p1.y, t.p1.z, 0.0f);
            padded[kTriStride * i + 2] = make_float4(t.p2.x, t.p2.y, t.p2.z, 0.0f);
        }
        // ... and the exact box of every leaf in the spare floats of its first triangle (read by the half-precision quad kernels)
        {
            // What the occluder cache remembers per stopped ray: the leaf -- or, where the leaves are small against the sun disc's footprint at the occluder (the
            // next ray from the same place is stopped by a NEIGHBOUR of that triangle), a record a few levels above it.  Footprint: the disc's 0.51 degrees over a
            // third of the scene = 0.003 extents; every quad level doubles the patch.  Atrium (18-cm leaves in 30 m): 0 -- the leaf; its x8 tessellation (2.3 cm): 2
            // (shadow launches -22 % against -2 % with leaves: profiles/r04_occluder/x8_hint_levels.log).
            std::vector<float> diag;
            diag.reserve(sceneView.bvhNodes.size() / 2 + 1);
            for (const BvhNode& nd : sceneView.bvhNodes)
                if (nd.triangleCount != 0)
                {
                    m.maxLeafTriangles = std::max(m.maxLeafTriangles, nd.triangleCount);
                    const float dx = nd.aabb.max.x - nd.aabb.min.x, dy = nd.aabb.max.y - nd.aabb.min.y, dz = nd.aabb.max.z - nd.aabb.min.z;
                    diag.push_back(std::sqrt(dx * dx + dy * dy + dz * dz));
                }
            const Aabb& root = sceneView.bvhNodes[0].aabb;
            const float extent = std::max({root.max.x - root.min.x, root.max.y - root.min.y, root.max.z - root.min.z});
            m.occluderHintLevels = 0;
            if (!diag.empty() && extent > 0.0f)
            {
                std::nth_element(diag.begin(), diag.begin() + diag.size() / 2, diag.end());
                const float median = diag[diag.size() / 2];
                if (median > 0.0f && std::isfinite(median) && std::isfinite(extent))
                    m.occluderHintLevels = static_cast<uint32_t>(std::clamp(static_cast<int>(std::floor(std::log2(0.003f * extent / median))) + 1, 0, 3));
            }
            if (const char* v = std::getenv("RF_OCCLUDER_HINT_LEVELS")) m.occluderHintLevels = static_cast<uint32_t>(std::clamp(std::atoi(v), 0, 8)); // experiments
        }
        m.leafBoxesValid = leafBoxesIntoTriangles(sceneView.bvhNodes.data(), sceneView.bvhNodes.size(), padded.data(), n, m.occluderHintLevels, &quadIndexOfNode);
        if (!m.leafBoxesValid)
        {
            // leaves that share a first triangle (hand-made tree): one slot cannot hold two exact boxes, so the layouts that cull a leaf by
            // the box in that slot stay off and the exact records (which carry every box themselves) are used
            m.wide.quadHalf = m.wide.quadLocal = m.wide.oct = nullptr;
            m.wideQuadHalf.release(), m.wideQuadLocal.release(), m.wideOct.release();
            m.optQuadHalfFromBounce = m.optQuadHalfShadowFromBounce = m.optQuadLocalFromBounce = m.optQuadLocalShadowFromBounce = m.optOctFromBounce = 0u;
        }
        m.triangles.upload(padded.data(), padded.size());
    }
    {
        // 80-B VertexAttributes (three padded normals, three uvs, texture index, pad) -> 64 B = one L2 sector per shaded hit
        const size_t        n = sceneView.vertexAttributes.size();
        std::vector<float4> packed(4 * n);
        for (size_t i = 0; i < n; ++i)
        {
            const VertexAttributes& v = sceneView.vertexAttributes[i];
            packed[4 * i] = make_float4(v.n0.x, v.n0.y, v.n0.z, v.n1.x);
            packed[4 * i + 1] = make_float4(v.n1.y, v.n1.z, v.n2.x, v.n2.y);
            packed[4 * i + 2] = make_float4(v.n2.z, v.uv0.x, v.uv0.y, v.uv1.x);
            packed[4 * i + 3] = make_float4(v.uv1.y, v.uv2.x, v.uv2.y, bitsFloat(v.textureIdx));
        }
        m.attributes.upload(packed.data(), packed.size());
        // kShade's record: positions + these four float4, 128 B per triangle
        std::vector<float4> rec(8 * n, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
        for (size_t i = 0; i < n; ++i)
        {
            const PositionAttribute& t = sceneView.positionAttributes[i];
            rec[8 * i] = make_float4(t.p0.x, t.p0.y, t.p0.z, 0.0f);
            rec[8 * i + 1] = make_float4(t.p1.x, t.p1.y, t.p1.z, 0.0f);
            rec[8 * i + 2] = make_float4(t.p2.x, t.p2.y, t.p2.z, 0.0f);
            for (int k = 0; k < 4; ++k) rec[8 * i + 3 + k] = packed[4 * i + k];
        }
        // kShadeSelfShadow: the exact box of the leaf a triangle sits in, in the spare floats of its record ({p0 lo.x} {p1 lo.y} {p2 lo.z} ... {hi.xyz, 1}).  Only where the
        // argument holds (kShade): boxes nested and regular all the way up (buildWide checked every parent / child pair), and the triangle in exactly ONE leaf.
        m.selfShadowOk = treeNestedRegular;
        if (m.selfShadowOk)
        {
            std::vector<uint8_t> leaves(n, 0);
            for (const BvhNode& nd : sceneView.bvhNodes)
                for (uint32_t t = 0; t < nd.triangleCount; ++t)
                {
                    const size_t i = static_cast<size_t>(nd.trianglesOffset) + t;
                    if (i >= n) continue; // (validateScene has rejected this)
                    const bool regular = std::fabs(nd.aabb.min.x) < 1e30f && std::fabs(nd.aabb.min.y) < 1e30f && std::fabs(nd.aabb.min.z) < 1e30f && std::fabs(nd.aabb.max.x) < 1e30f &&
                                         std::fabs(nd.aabb.max.y) < 1e30f && std::fabs(nd.aabb.max.z) < 1e30f && nd.aabb.min.x <= nd.aabb.max.x && nd.aabb.min.y <= nd.aabb.max.y &&
                                         nd.aabb.min.z <= nd.aabb.max.z;
                    leaves[i] = static_cast<uint8_t>(std::min(leaves[i] + (regular ? 1 : 2), 2));
                    rec[8 * i].w = nd.aabb.min.x, rec[8 * i + 1].w = nd.aabb.min.y, rec[8 * i + 2].w = nd.aabb.min.z;
                    rec[8 * i + 7] = make_float4(nd.aabb.max.x, nd.aabb.max.y, nd.aabb.max.z, 0.0f);
                }
            for (size_t i = 0; i < n; ++i) rec[8 * i + 7].w = leaves[i] == 1 ? 1.0f : 0.0f;
        }
        m.shadeRecords.upload(rec.data(), rec.size());
    }

    // texture blob + descriptors in the order of the model's textures (reference_path_tracer.cpp:210-270)
    {
        std::vector<TextureDescriptor> descs;
        std::vector<uint32_t>          blob;
        for (const TextureView& t : sceneView.baseColorTextures)
        {
            const uint32_t offset = static_cast<uint32_t>(blob.size());
            const size_t   n = static_cast<size_t>(t.width) * t.height;
            blob.insert(blob.end(), t.pixels, t.pixels + n);
            descs.push_back({t.width, t.height, offset});
        }
        // the reference refuses texture blobs above its 1 GiB binding limit (reference_path_tracer.cpp:254-263,
        // gpu_limits.hpp); HBM has room, but u32 texel offsets cap the blob at 2^32 texels
        if (blob.size() > 0xFFFFFFFFull) throw std::runtime_error("Texture buffer size exceeds the 32-bit texel offset range.");
        if (blob.empty()) blob.push_back(0xFFFFFFFFu);
        if (descs.empty()) descs.push_back({1, 1, 0});
        m.textureDescriptors.upload(descs.data(), descs.size());
        m.texels.upload(blob.data(), blob.size());
    }
    {
        m.blueNoise.upload(blueNoiseTable(), 128 * 128 * 2);
        float lut[256];
        for (int i = 0; i < 256; ++i)
            lut[i] = static_cast<float>(std::pow(static_cast<double>(static_cast<float>(i) / 255.0f), static_cast<double>(2.2f)));
        m.albedoLut.upload(lut, 256);
    }
    m.scene.nodes = m.nodes.ptr;
    m.scene.triangles = m.triangles.ptr;
    m.scene.attributes = m.attributes.ptr;
    m.scene.shadeRecords = m.shadeRecords.ptr;
    m.scene.textureDescriptors = m.textureDescriptors.ptr;
    m.scene.texels = m.texels.ptr;
    m.scene.numTexels = m.texels.count;
    m.scene.blueNoise = m.blueNoise.ptr;
    m.scene.albedoLut = m.albedoLut.ptr;

    DeviceCounters zero{};
    m.counters.upload(&zero, 1);
    {
        const std::vector<unsigned long long> z(4 * RenderStats::kMaxBounceStats, 0ull);
        m.bounceTotals.upload(z.data(), z.size());
    }
    {
        hipDeviceProp_t prop{};
        RF_HIP(hipGetDeviceProperties(&prop, m.device));
        int perCu = 0;
        RF_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, reinterpret_cast<const void*>(traceWideKernel(false, false, false, 0, false)), kBlock, 0));
        m.wideBlocks = static_cast<uint32_t>(std::max(perCu, 1)) * static_cast<uint32_t>(prop.multiProcessorCount);
        m.multiProcessors = static_cast<uint32_t>(prop.multiProcessorCount);
        m.skyBlocks = 8u * static_cast<uint32_t>(prop.multiProcessorCount);
        if (const char* v = std::getenv("RF_TRAVERSAL_VARIANT")) m.traversalVariant = std::atoi(v);
        if (const char* v = std::getenv("RF_PACKET_BOUNCES")) m.optPacketBounces = static_cast<uint32_t>(std::max(std::atoi(v), 0)); // experiments: whole test suite through kTracePacket
        if (!m.wideUsable) m.traversalVariant = 0;
    }

    m.maxWidth = desc.maxWidth ? desc.maxWidth : desc.renderParams.width;
    m.maxHeight = desc.maxHeight ? desc.maxHeight : desc.renderParams.height;
    const uint64_t maxTiles = static_cast<uint64_t>((m.maxWidth + kTileSize - 1) / kTileSize) * ((m.maxHeight + kTileSize - 1) / kTileSize);
    // 1 Gi paths per batch by default (133 GB of path state + queues out of 288 GB; allocated on demand, so a render only ever
    // takes samples x pixels x 124 B): later bounces of a batch keep ~15 % of
    // the paths, a traversal launch needs millions of rays to fill 6144 persistent waves and to amortise its tail, and the
    // more samples of a pixel a batch holds, the closer the directions of the 64 direction-sorted samples that share a
    // wave (FrameParams::samplePerm).  Measured on the atrium, 1080p, Mrays/s: 64 Mi 5125, 128 Mi 5171, 256 Mi 5224 (before
    // the sample sort); 256 Mi 5692, 512 Mi 5826 / 5796, 1 Gi 5983 (with it).
    const uint64_t want = desc.maxPathsInFlight ? desc.maxPathsInFlight : (1024ull << 20);
    // path slots and queue indices are 32-bit: at most 2^31 paths per batch, and one sample of the whole
    // (padded) frame must fit in a batch
    constexpr uint64_t kMaxPathsPerBatch = 1ull << 31;
    if (maxTiles * 1024 > kMaxPathsPerBatch) throw std::runtime_error("framebuffer too large: more than 2^31 pixels per rank");
    m.maxPaths = std::min(std::max<uint64_t>(want, maxTiles * 1024), kMaxPathsPerBatch);

    m.params = desc.renderParams;
    if (alignedSkyState(m.params.sky, m.sky) != SkyResult::Success) throw std::runtime_error("sky parameters out of range");
    m.updateSunBasis();
    m.configureShard();
}

Renderer::~Renderer()
{
    if (!mImpl) return;
    (void)hipSetDevice(mImpl->device);
    (void)hipStreamSynchronize(mImpl->stream);
    for (auto& t : mImpl->timed)
    {
        (void)hipEventDestroy(t.start);
        (void)hipEventDestroy(t.stop);
    }
    for (auto& b : mImpl->pendingBatches)
    {
        (void)hipEventDestroy(b.start);
        (void)hipEventDestroy(b.stop);
    }
    for (auto e : mImpl->eventPool) (void)hipEventDestroy(e);
    if (mImpl->lookEvent) (void)hipEventDestroy(mImpl->lookEvent);
    if (mImpl->lookBatchHost) (void)hipHostFree(mImpl->lookBatchHost);
    (void)hipStreamDestroy(mImpl->stream);
}

void Renderer::setRenderParameters(const RenderParameters& p)
{
    Impl& m = *mImpl;
    if (m.params == p) return; // reference_path_tracer.cpp:556-563
    if (p.width > m.maxWidth || p.height > m.maxHeight) throw std::runtime_error("framebuffer size exceeds maxFramebufferSize");
    SkyStateGpu sky;
    if (alignedSkyState(p.sky, sky) != SkyResult::Success) throw std::runtime_error("sky parameters out of range");
    RF_HIP(hipSetDevice(m.device));
    RF_HIP(hipStreamSynchronize(m.stream));
    const bool resized = p.width != m.params.width || p.height != m.params.height;
    m.params = p;
    m.sky = sky;
    m.updateSunBasis();
    m.accumulated = 0;
    m.imageDirty = true;
    if (resized) m.configureShard();
}

void Renderer::setTileShard(uint32_t rank, uint32_t worldSize)
{
    Impl& m = *mImpl;
    if (worldSize == 0 || rank >= worldSize) throw std::runtime_error("invalid tile shard");
    RF_HIP(hipSetDevice(m.device));
    RF_HIP(hipStreamSynchronize(m.stream));
    m.rank = rank;
    m.worldSize = worldSize;
    m.configureShard();
}

std::span<const uint32_t> Renderer::shardTiles() const { return mImpl->tiles; }

void Renderer::render(uint32_t numFrames)
{
    Impl& m = *mImpl;
    RF_HIP(hipSetDevice(m.device));
    const uint32_t spp = m.params.samplingParams.numSamplesPerPixel;
    const uint64_t pixelsPadded = static_cast<uint64_t>(m.tiles.size()) * 1024;
    // Each reference render() call: frame = frameCount++, then one sample if accumulated < spp
    // (reference_path_tracer.cpp:577-591, wgsl:47-57).  Calls past spp only advance frameCount.
    uint32_t remaining = numFrames;
    while (remaining > 0)
    {
        if (m.accumulated >= spp || pixelsPadded == 0)
        {
            m.frameCount += remaining;
            break;
        }
        if (m.imageDirty)
        {
            RF_HIP(hipMemsetAsync(m.image, 0, pixelsPadded * sizeof(float4), m.stream)); // wgsl:47-49
            m.imageDirty = false;
        }
        // equal batches (320 samples with room for 256 per batch -> 160 + 160, not 256 + 64): a small trailing batch has
        // short launches and, with few samples per pixel, less coherent waves
        // m.maxPaths is the CONFIGURED depth (the default or the caller's) and is never changed here: what a call has to give up
        // because memory is short at that moment (another handle alive, a shared GPU) is given up for that call only.
        const uint32_t todo = std::min(remaining, spp - m.accumulated);
        uint32_t       n = 0;
        uint64_t       depth = m.maxPaths;
        for (;;)
        {
            const uint32_t perBatch = static_cast<uint32_t>(std::max<uint64_t>(1, depth / pixelsPadded));
            const uint32_t numBatches = (todo + perBatch - 1) / perBatch;
            n = (todo + numBatches - 1) / numBatches;
            const uint64_t need = static_cast<uint64_t>(n) * pixelsPadded;
            if (need <= m.allocatedPaths) break;
            // The batch depth is a speed knob, never a requirement: a device with less free memory than the
            // batch wants (a smaller or shared GPU, a second handle on this one) traces the same samples in more, smaller
            // batches -- same image.  First by what hipMemGetInfo reports, then by halving if hipMalloc still refuses.
            const uint64_t fit = m.pathsThatFit();
            if (need > fit && n > 1)
            {
                depth = std::max<uint64_t>(pixelsPadded, std::min(depth / 2, fit));
                continue;
            }
            if (m.ensurePathState(need)) break;
            if (n == 1) throw std::runtime_error("out of device memory: one sample of the frame (" + std::to_string(need * Impl::kBytesPerPath >> 20) + " MiB of path state) does not fit");
            depth = std::max<uint64_t>(pixelsPadded, depth / 2);
        }
        if (depth != m.effectivePaths)
        {
            // said once per change, not per batch: shallower batches are a silent loss of speed otherwise
            if (depth < m.maxPaths)
                std::fprintf(stderr, "[rf] device memory is short: batches of %llu paths instead of the configured %llu (%u samples per batch); same image, shorter launches\n",
                             static_cast<unsigned long long>(depth), static_cast<unsigned long long>(m.maxPaths), n);
            else if (m.effectivePaths != 0 && m.effectivePaths < m.maxPaths)
                std::fprintf(stderr, "[rf] device memory is back: batches of the configured %llu paths again\n", static_cast<unsigned long long>(m.maxPaths));
            m.effectivePaths = depth;
        }
        m.traceBatch(m.frameCount, n);
        m.hostStats.batchSamplesUsed = n, m.hostStats.batchPathsUsed = static_cast<uint64_t>(n) * pixelsPadded, ++m.hostStats.batchesTraced;
        m.frameCount += n;
        m.accumulated += n;
        remaining -= n;
    }
}

float Renderer::averageRenderpassDurationMs() const
{
    Impl& m = *mImpl;
    (void)hipSetDevice(m.device);
    m.collectBatchTimings();
    if (m.passDurationsMs.empty()) return 0.0f;
    float sum = 0.0f;
    for (float v : m.passDurationsMs) sum += v;
    return sum / static_cast<float>(m.passDurationsMs.size());
}

float Renderer::renderProgressPercentage() const
{
    return 100.0f * static_cast<float>(mImpl->accumulated) / static_cast<float>(mImpl->params.samplingParams.numSamplesPerPixel);
}

uint32_t Renderer::accumulatedSampleCount() const { return mImpl->accumulated; }
uint32_t Renderer::width() const { return mImpl->params.width; }
uint32_t Renderer::height() const { return mImpl->params.height; }
uint32_t Renderer::shardRank() const { return mImpl->rank; }
uint32_t Renderer::shardWorldSize() const { return mImpl->worldSize; }
int      Renderer::deviceOrdinal() const { return mImpl->device; }
void*    Renderer::streamHandle() const { return mImpl->stream; }
uint32_t Renderer::numBounces() const { return mImpl->params.samplingParams.numBounces; }

void Renderer::synchronize()
{
    RF_HIP(hipSetDevice(mImpl->device));
    RF_HIP(hipStreamSynchronize(mImpl->stream));
}

void Renderer::readAccumulation(float* dst)
{
    Impl& m = *mImpl;
    synchronize();
    const size_t       pixelsPadded = m.tiles.size() * 1024;
    std::vector<float> compact(pixelsPadded * 4);
    if (m.imageDirty) std::fill(compact.begin(), compact.end(), 0.0f);
    else if (pixelsPadded) RF_HIP(hipMemcpy(compact.data(), m.image, pixelsPadded * sizeof(float4), hipMemcpyDeviceToHost));
    std::memset(dst, 0, static_cast<size_t>(m.params.width) * m.params.height * 4 * sizeof(float));
    untileHost(compact.data(), m.tiles.data(), static_cast<uint32_t>(m.tiles.size()), m.params.width, m.params.height, dst);
}

void*    Renderer::accumulationDevicePointer() const { return mImpl->image; }

void Renderer::clearAccumulationIfStale()
{
    Impl& m = *mImpl;
    if (!m.imageDirty || m.image == nullptr) return;
    RF_HIP(hipSetDevice(m.device));
    // (this shard's part of the buffer; a bound buffer is never written past the size its owner gave in bindAccumulationBuffer --
    // configureShard refuses a shard that needs more)
    RF_HIP(hipMemsetAsync(m.image, 0, std::min<uint64_t>(accumulationBytes(), m.imageBytes), m.stream)); // wgsl:47-49
    m.imageDirty = false;
}

void Renderer::layoutInfo(uint32_t (&layouts)[48], uint32_t (&misc)[4], float& quadHalfAreaRatio, uint64_t& treeBytes) const
{
    const Impl& m = *mImpl;
    for (uint32_t b = 1; b <= 16; ++b)
    {
        const int ls = m.shadowLayoutFor(b);
        layouts[b - 1] = static_cast<uint32_t>(m.closestLayoutFor(b));
        layouts[16 + b - 1] = static_cast<uint32_t>(ls);
        layouts[32 + b - 1] = (ls != kLayoutScalar && ls != kLayoutPacket && m.cachedShadowFor(b)) ? 1u : 0u;
    }
    misc[0] = m.occluderHintLevels, misc[1] = m.optShadowFirstLookFromBounce, misc[2] = m.optDenseLeafMin;
#if defined(RF_EXP_LEGACY_LAYOUTS)
    misc[3] = 1u;
#else
    misc[3] = 0u;
#endif
    quadHalfAreaRatio = m.quadHalfAreaRatio;
    treeBytes = m.treeBytes;
}

void Renderer::memoryInfo(uint64_t& pathStateBytes, uint64_t& pathsAllocated, uint64_t& maxPathsPerBatch, uint64_t& sceneBytes) const
{
    const Impl& m = *mImpl;
    pathsAllocated = m.allocatedPaths;
    pathStateBytes = m.allocatedPaths * Impl::kBytesPerPath;
    maxPathsPerBatch = m.effectivePaths ? std::min(m.effectivePaths, m.maxPaths) : m.maxPaths;
    sceneBytes = (m.nodes.count + m.triangles.count + m.wideNodes.count + m.wideCompact.count + m.wideHot.count + m.wideOwn.count + m.wideQuad.count + m.wideQuadHalf.count + m.wideQuadLocal.count + m.wideOct.count + m.attributes.count + m.shadeRecords.count) * sizeof(float4) +
                 m.texels.count * sizeof(uint32_t) + m.bigLeaves.count * sizeof(uint2) + m.occluderGrid.count * sizeof(uint32_t); // (the occluder grid: allocated by the first batch)
}
uint64_t Renderer::accumulationBytes() const { return static_cast<uint64_t>(mImpl->tiles.size()) * 1024 * sizeof(float4); }

void Renderer::bindAccumulationBuffer(void* devicePtr, uint64_t bytes)
{
    Impl& m = *mImpl;
    synchronize();
    // The caller's buffer was produced on streams this library does not know (e.g. torch's current stream filling
    // it with zeros), and the handle's stream is non-blocking: wait for the whole device once, here, so that the
    // first memset / kAccumulate into the buffer cannot overtake the caller's own writes.  After this call the buffer
    // belongs to the handle's stream until rf_renderer_synchronize() returns ("Streams").
    RF_HIP(hipDeviceSynchronize());
    if (devicePtr == nullptr)
    {
        m.image = nullptr;
        m.configureShard();
        return;
    }
    if (bytes < accumulationBytes()) throw std::runtime_error("accumulation buffer too small");
    m.image = static_cast<float4*>(devicePtr);
    m.imageBytes = bytes;
    m.accumulated = 0;
    m.imageDirty = true;
}

void Renderer::readTonemapped(uint32_t* dst)
{
    Impl& m = *mImpl;
    synchronize();
    const uint32_t         n = static_cast<uint32_t>(m.tiles.size() * 1024);
    DeviceBuffer<uint32_t> out;
    out.alloc(std::max<uint32_t>(n, 1));
    if (m.imageDirty)
    {
        RF_HIP(hipMemsetAsync(m.image, 0, static_cast<size_t>(n) * sizeof(float4), m.stream));
        m.imageDirty = false;
    }
    if (n) hipLaunchKernelGGL(tonemapKernel(), dim3((n + 255) / 256), dim3(256), 0, m.stream, m.image, n, m.accumulated, m.params.exposure, out.ptr);
    RF_HIP(hipStreamSynchronize(m.stream));
    std::vector<uint32_t> compact(n);
    if (n) RF_HIP(hipMemcpy(compact.data(), out.ptr, static_cast<size_t>(n) * 4, hipMemcpyDeviceToHost));
    std::memset(dst, 0, static_cast<size_t>(m.params.width) * m.params.height * 4);
    const uint32_t tilesX = (m.params.width + kTileSize - 1) / kTileSize;
    for (size_t t = 0; t < m.tiles.size(); ++t)
        for (uint32_t w = 0; w < 1024; ++w)
        {
            const uint32_t block = w >> 6, lane = w & 63u;
            const uint32_t x = (m.tiles[t] % tilesX) * kTileSize + (block & 3u) * 8u + (lane & 7u);
            const uint32_t y = (m.tiles[t] / tilesX) * kTileSize + (block >> 2) * 8u + (lane >> 3);
            if (x < m.params.width && y < m.params.height) dst[static_cast<size_t>(y) * m.params.width + x] = compact[t * 1024 + w];
        }
}

void Renderer::tonemapDeviceImage(const void* imageDevice, uint64_t numPixels, uint32_t samples, uint32_t* dst)
{
    Impl& m = *mImpl;
    if (numPixels == 0) return;
    if (numPixels > 0xFFFFFFFFull) throw std::runtime_error("image too large");
    RF_HIP(hipSetDevice(m.device));
    const uint32_t         n = static_cast<uint32_t>(numPixels);
    DeviceBuffer<uint32_t> out;
    out.alloc(n);
    hipLaunchKernelGGL(tonemapKernel(), dim3((n + 255) / 256), dim3(256), 0, m.stream, static_cast<const float4*>(imageDevice), n, samples, m.params.exposure, out.ptr);
    RF_HIP(hipGetLastError());
    RF_HIP(hipMemcpyAsync(dst, out.ptr, static_cast<size_t>(n) * 4, hipMemcpyDeviceToHost, m.stream));
    RF_HIP(hipStreamSynchronize(m.stream));
}

void Renderer::renderDeferred(uint32_t numFrames)
{
    Impl& m = *mImpl;
    RF_HIP(hipSetDevice(m.device));
    const uint32_t W = m.params.width, H = m.params.height;
    const size_t   n = static_cast<size_t>(W) * H;
    if (m.deferredSample.count != 3 * n)
    {
        RF_HIP(hipStreamSynchronize(m.stream));
        m.deferredSample.alloc(3 * n);
        m.deferredAccum.alloc(3 * n);
        m.deferredBgra.alloc(n);
        m.deferredFrameCount = 0;
    }
    const uint32_t waves = ((W + 7) / 8) * ((H + 7) / 8);
    for (uint32_t f = 0; f < numFrames; ++f)
    {
        // r2Sequence(frameCount, 1 << 20), src/common/r_sequence.hpp:9-21 (the host-side variant: + 0.5, 1/G constants)
        constexpr float G = 1.32471795f;
        constexpr float A1 = 1.0f / G, A2 = 1.0f / (G * G);
        const float     i = static_cast<float>(m.deferredFrameCount % (1u << 20));
        const float     a = 0.5f + A1 * i, b = 0.5f + A2 * i;
        const float     jx = a - std::floor(a), jy = b - std::floor(b);
        hipLaunchKernelGGL(deferredLightingKernel(), dim3((waves * 64 + kBlock - 1) / kBlock), dim3(kBlock), 0, m.stream, m.scene, m.sky, m.sunBasis, m.params.camera, W, H,
                           m.deferredFrameCount, jx, jy, m.params.exposure, m.deferredSample.ptr, m.deferredAccum.ptr, m.deferredBgra.ptr, m.counters.ptr);
        ++m.deferredFrameCount;
    }
    RF_HIP(hipGetLastError());
}

void Renderer::resetDeferred() { mImpl->deferredFrameCount = 0; }
uint32_t Renderer::deferredFrameCount() const { return mImpl->deferredFrameCount; }

void Renderer::readDeferred(float* sampleRgb, float* accumulationRgb, uint32_t* bgra8)
{
    Impl& m = *mImpl;
    synchronize();
    const size_t n = static_cast<size_t>(m.params.width) * m.params.height;
    if (m.deferredSample.count != 3 * n) throw std::runtime_error("no deferred frame has been rendered at this framebuffer size");
    if (sampleRgb) RF_HIP(hipMemcpy(sampleRgb, m.deferredSample.ptr, 3 * n * sizeof(float), hipMemcpyDeviceToHost));
    if (accumulationRgb) RF_HIP(hipMemcpy(accumulationRgb, m.deferredAccum.ptr, 3 * n * sizeof(float), hipMemcpyDeviceToHost));
    if (bgra8) RF_HIP(hipMemcpy(bgra8, m.deferredBgra.ptr, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
}

void Renderer::setCounting(bool enabled) { mImpl->counting = enabled; }

void Renderer::setOption(const std::string& name, int64_t value)
{
    if (name == "traversal_variant") mImpl->traversalVariant = mImpl->wideUsable ? static_cast<int>(value) : 0;
    else if (name == "refill_min") mImpl->optRefillMin = mImpl->optRefillMinDeep = mImpl->optRefillMinDeepDense = mImpl->optRefillMinDeepQuad = static_cast<uint32_t>(value); // (all: a sweep of one value covers every launch)
    else if (name == "refill_min_deep") mImpl->optRefillMinDeep = mImpl->optRefillMinDeepDense = mImpl->optRefillMinDeepQuad = static_cast<uint32_t>(value);
    else if (name == "refill_deep_from_bounce") mImpl->optRefillDeepFromBounce = static_cast<uint32_t>(std::max<int64_t>(value, 1));
    else if (name == "leaf_vote") mImpl->optLeafVote = mImpl->optLeafVoteClosest = static_cast<uint32_t>(value);
    else if (name == "dense_leaf_min") mImpl->optDenseLeafMin = static_cast<uint32_t>(std::clamp<int64_t>(value, 0, 15));
    else if (name == "dense_leaf_from_bounce") mImpl->optDenseLeafFromBounce = static_cast<uint32_t>(std::max<int64_t>(value, 1));
    else if (name == "chunk") mImpl->optChunk = mImpl->optChunkEarly = static_cast<uint32_t>(std::clamp<int64_t>(value, 1, 1 << 20)); // (both, as refill_min)
    else if (name == "chunk_early") mImpl->optChunkEarly = static_cast<uint32_t>(std::clamp<int64_t>(value, 1, 1 << 20));
    else if (name == "shade_sort_from_bounce") mImpl->optShadeSortFromBounce = static_cast<uint32_t>(std::max<int64_t>(value, 0));
    else if (name == "chunk_early_bounces") mImpl->optChunkEarlyBounces = static_cast<uint32_t>(std::max<int64_t>(value, 0));
    else if (name == "sample_sort") mImpl->optSampleSort = value != 0;
    else if (name == "accumulate_runs") mImpl->optAccumulateRuns = value != 0;
    else if (name == "uniform_fetch") mImpl->optUniformFetch = static_cast<int>(value);
    else if (name == "compact_from_bounce") mImpl->optCompactFromBounce = static_cast<uint32_t>(std::max<int64_t>(value, 0));
    else if (name == "compact_shadow_from_bounce") mImpl->optCompactShadowFromBounce = static_cast<uint32_t>(std::max<int64_t>(value, 0));
    else if (name == "query_compact") mImpl->optQueryCompact = static_cast<int>(value);
    else if (name == "quad_from_bounce") mImpl->optQuadFromBounce = static_cast<uint32_t>(std::max<int64_t>(value, 0));
    else if (name == "quad_shadow_from_bounce") mImpl->optQuadShadowFromBounce = static_cast<uint32_t>(std::max<int64_t>(value, 0));
    else if (name == "quad_local_from_bounce") mImpl->optQuadLocalFromBounce = static_cast<uint32_t>(std::max<int64_t>(value, 0));
    else if (name == "quad_local_shadow_from_bounce") mImpl->optQuadLocalShadowFromBounce = static_cast<uint32_t>(std::max<int64_t>(value, 0));
    else if (name == "quad_half_from_bounce") mImpl->optQuadHalfFromBounce = static_cast<uint32_t>(std::max<int64_t>(value, 0));
    else if (name == "quad_half_shadow_from_bounce") mImpl->optQuadHalfShadowFromBounce = static_cast<uint32_t>(std::max<int64_t>(value, 0));
    else if (name == "quad_except_mask") mImpl->optQuadExceptMask = static_cast<uint32_t>(value);
    else if (name == "quad_shadow_except_mask") mImpl->optQuadShadowExceptMask = static_cast<uint32_t>(value);
    else if (name == "oct_from_bounce") mImpl->optOctFromBounce = static_cast<uint32_t>(std::max<int64_t>(value, 0));
    else if (name == "hot_from_bounce") mImpl->optHotFromBounce = static_cast<uint32_t>(std::max<int64_t>(value, 0));
    else if (name == "hot_shadow_from_bounce") mImpl->optHotShadowFromBounce = static_cast<uint32_t>(std::max<int64_t>(value, 0));
    else if (name == "packet_bounces") mImpl->optPacketBounces = static_cast<uint32_t>(std::max<int64_t>(value, 0));
    else if (name == "shade_blocks") mImpl->optShadeBlocks = static_cast<uint32_t>(value);
    else if (name == "const_primary_origin") mImpl->optConstPrimaryOrigin = value != 0;
    else if (name == "shadow_self_test") mImpl->optShadowSelfTest = value != 0;
    // 0 (default): sin / cos / acos / exp / pow as the f32 rounding of a specified f64 evaluation (bit-identical to the test oracle); 1: the device math library's f32 functions
    // (kRaygen's lens / cone angle, kSky's dome: rf_device.hpp tSin ...), graded by SURVEY 8(d)'s tolerance.  Set it before the first sample of an accumulation.
    else if (name == "transcendentals") mImpl->optTranscendentalsF32 = value != 0;
    else if (name == "dense_raygen") mImpl->optDenseRaygen = value != 0; // 0: kRaygen appends to the first queue with one atomic per 1 024 slots (until round 6)
    else if (name == "slot_group_shift") mImpl->optSlotGroupShift = value < 0 || value > 10 ? kSlotSampleMajor : static_cast<uint32_t>(value); // -1: sample-major
    else if (name == "shadow_nearest_first") mImpl->shadowNearestFirst = value != 0;
    else if (name == "shadow_sign_order" || name == "shadow_record_order") mImpl->optShadowSignOrder = value != 0;
    else if (name == "shadow_first_look_from_bounce") mImpl->optShadowFirstLookFromBounce = static_cast<uint32_t>(std::max<int64_t>(value, 0));
    else if (name == "occluder_grid_log2_cells") mImpl->optOccluderGridLog2Cells = static_cast<uint32_t>(std::clamp<int64_t>(value, 4, 26));
    else if (name == "occluder_grid_cells") mImpl->optOccluderGridCells = static_cast<uint32_t>(std::clamp<int64_t>(value, 0, 1 << 16));
    else if (name == "occluder_cache_bounces") mImpl->optOccluderCacheBounces = static_cast<uint32_t>(std::max<int64_t>(value, 0));
    else if (name == "reserve_samples")
    {
        // allocate path state for batches of up to `value` samples now (otherwise it grows on first use)
        Impl&          m = *mImpl;
        const uint64_t pixelsPadded = static_cast<uint64_t>(m.tiles.size()) * 1024;
        RF_HIP(hipSetDevice(m.device));
        if (pixelsPadded)
        {
            uint64_t samples = std::min<uint64_t>(std::max<uint64_t>(static_cast<uint64_t>(value), 1), std::max(m.maxPaths / pixelsPadded, uint64_t{1}));
            samples = std::max<uint64_t>(std::min(samples, m.pathsThatFit() / pixelsPadded), 1);
            (void)m.ensurePathState(samples * pixelsPadded); // best effort: render() falls back to smaller batches
        }
    }
    else if (name == "query_variant") mImpl->queryVariant = mImpl->wideUsable ? static_cast<int>(value) : 0;
    else if (name == "persistent_blocks")
    {
        // > 0: every persistent launch takes that grid; 0: back to what the device holds of the kernel launched (residentBlocks)
        mImpl->wideBlocksForced = value > 0;
        mImpl->wideBlocks = value > 0 ? static_cast<uint32_t>(value) : mImpl->residentBlocks(traceWideKernel(false, false, false, 0, false), mImpl->optExtraLds);
    }
    else if (name == "extra_lds")
    {
        Impl& m = *mImpl;
        m.optExtraLds = static_cast<uint32_t>(std::max<int64_t>(value, 0));
        hipDeviceProp_t prop{};
        RF_HIP(hipGetDeviceProperties(&prop, m.device));
        int perCu = 0;
        RF_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, reinterpret_cast<const void*>(traceWideKernel(false, false, false, 0, false)), kBlock, m.optExtraLds));
        m.wideBlocks = static_cast<uint32_t>(std::max(perCu, 1)) * static_cast<uint32_t>(prop.multiProcessorCount);
    }
    else throw std::invalid_argument("unknown option " + name);
}
void Renderer::setTiming(bool enabled) { mImpl->timing = enabled; }

void Renderer::resetStats()
{
    Impl& m = *mImpl;
    synchronize();
    m.collectTimings();
    DeviceCounters zero{};
    RF_HIP(hipMemcpy(m.counters.ptr, &zero, sizeof zero, hipMemcpyHostToDevice));
    RF_HIP(hipMemset(m.bounceTotals.ptr, 0, 4 * RenderStats::kMaxBounceStats * sizeof(unsigned long long)));
    m.hostStats = RenderStats{};
    m.primaryRaysHost = 0;
}

RenderStats Renderer::stats()
{
    Impl& m = *mImpl;
    synchronize();
    m.collectTimings();
    DeviceCounters c{};
    RF_HIP(hipMemcpy(&c, m.counters.ptr, sizeof c, hipMemcpyDeviceToHost));
    RenderStats s = m.hostStats;
    s.primaryRays = m.primaryRaysHost;
    s.closestRays = c.closestRays;
    s.shadowRays = c.shadowRays;
    s.closestNodeVisits = c.closestNodeVisits;
    s.closestTriangleTests = c.closestTriangleTests;
    s.shadowNodeVisits = c.shadowNodeVisits;
    s.shadowTriangleTests = c.shadowTriangleTests;
    s.stackHighWater = c.stackHigh;
    s.closestRecordFetches = c.closestRecordFetches;
    s.shadowRecordFetches = c.shadowRecordFetches;
    s.paths = m.primaryRaysHost;
    s.abandonedRays = c.abandonedRays;
    s.scalarRedoRays = c.scalarRedo[0] + c.scalarRedo[1];
    if (std::getenv("RF_DEBUG_COUNTERS"))
        std::fprintf(stderr, "[rf] rays redone by the scalar traversal: closest %llu of %llu, shadow %llu of %llu\n", c.scalarRedo[0], c.closestRays, c.scalarRedo[1], c.shadowRays);
#if defined(RF_EXP_PHASE)
    if (std::getenv("RF_DEBUG_COUNTERS") && !m.counting)
    {
        for (int k = 0; k < 2; ++k)
        {
            const double rays = static_cast<double>(k ? c.shadowRays : c.closestRays), steps = static_cast<double>(k ? c.shadowRecordFetches : c.closestRecordFetches);
            std::fprintf(stderr,
                         "[rf-phase] %s: rays %.0f | per ray: steps %.2f leaf visits %.2f triangle tests %.2f | wave trips per 64 rays: outer %.2f descend %.2f leaf phase %.2f refill %.2f"
                         " | lanes busy: descend %.3f leaf phase %.3f\n",
                         k ? "shadow " : "closest", rays, steps / rays, c.leafTrips[k] / rays, c.popLaneTrips[k] / rays, c.outerTrips[k] * 64.0 / rays,
                         c.descendTrips[k] * 64.0 / rays, c.leafPhases[k] * 64.0 / rays, c.refillTrips[k] * 64.0 / rays, steps / (64.0 * c.descendTrips[k]),
                         c.leafTrips[k] / (64.0 * c.leafPhases[k]));
            std::fprintf(stderr, "[rf-phase] %s: lanes of a descend trip: stepping %.3f, parked at a leaf %.3f, without a ray %.3f | lanes of a leaf pass: at a leaf %.3f, at an interior node %.3f, without a ray %.3f\n",
                         k ? "shadow " : "closest", steps / (64.0 * c.descendTrips[k]), c.descendParked[k] / (64.0 * c.descendTrips[k]), c.descendIdle[k] / (64.0 * c.descendTrips[k]),
                         c.leafTrips[k] / (64.0 * c.leafPhases[k]), c.leafInterior[k] / (64.0 * c.leafPhases[k]), c.leafIdle[k] / (64.0 * c.leafPhases[k]));
        }
        std::fprintf(stderr, "[rf-phase] occluder cache: shadow rays %llu, occluded %llu, cache tried %llu, answered at the cached leaf %llu\n", c.shadowRays, c.occludedRays, c.occluderTried, c.occluderHit);
    }
#endif
    if (std::getenv("RF_DEBUG_COUNTERS") && m.counting)
    {
        for (int k = 0; k < 2; ++k)
        {
            const double rays = static_cast<double>(k ? c.shadowRays : c.closestRays), steps = static_cast<double>(k ? c.shadowRecordFetches : c.closestRecordFetches);
            const double tris = static_cast<double>(k ? c.shadowTriangleTests : c.closestTriangleTests);
            std::fprintf(stderr,
                         "[rf] %s: rays %.0f | per ray: steps %.2f tris %.2f pops %.2f | wave trips per 64 rays: outer %.2f descend %.2f leafphase %.2f leaf %.2f refill %.2f"
                         " | lane utilisation: descend %.3f leaf %.3f\n",
                         k ? "shadow " : "closest", rays, steps / rays, tris / rays, c.popLaneTrips[k] / rays, c.outerTrips[k] * 64.0 / rays,
                         c.descendTrips[k] * 64.0 / rays, c.leafPhases[k] * 64.0 / rays, c.leafTrips[k] * 64.0 / rays, c.refillTrips[k] * 64.0 / rays,
                         steps / (64.0 * c.descendTrips[k]), tris / (64.0 * c.leafTrips[k]));
        }
    }
    unsigned long long totals[4 * RenderStats::kMaxBounceStats];
    RF_HIP(hipMemcpy(totals, m.bounceTotals.ptr, sizeof totals, hipMemcpyDeviceToHost));
    for (uint32_t b = 0; b < RenderStats::kMaxBounceStats; ++b)
    {
        s.closestRaysByBounce[b] = totals[b];
        s.shadowRaysByBounce[b] = totals[RenderStats::kMaxBounceStats + b];
        s.shadowRaysHintAnswered += totals[2 * RenderStats::kMaxBounceStats + b];
        s.shadowRaysSelfAnswered += totals[3 * RenderStats::kMaxBounceStats + b];
    }
    return s;
}

void Renderer::tracePrimaryStats(const Camera& camera, uint32_t width, uint32_t height, uint32_t* nodesVisitedOut, uint8_t* hitOut,
                                 float* tOut, uint32_t* triangleTestsOut)
{
    Impl& m = *mImpl;
    synchronize();
    const size_t           n = static_cast<size_t>(width) * height;
    DeviceBuffer<uint32_t> nv, tt;
    DeviceBuffer<uint8_t>  hit;
    DeviceBuffer<float>    t;
    nv.alloc(n);
    tt.alloc(n);
    hit.alloc(n);
    t.alloc(n);
    const uint32_t waves = ((width + 7) / 8) * ((height + 7) / 8);
    hipLaunchKernelGGL(primaryStatsKernel(), dim3((waves * 64 + kBlock - 1) / kBlock), dim3(kBlock), 0, m.stream, m.scene, camera, width, height,
                       nv.ptr, hit.ptr, t.ptr, tt.ptr, m.counters.ptr);
    RF_HIP(hipGetLastError());
    RF_HIP(hipStreamSynchronize(m.stream));
    RF_HIP(hipMemcpy(nodesVisitedOut, nv.ptr, n * 4, hipMemcpyDeviceToHost));
    if (hitOut) RF_HIP(hipMemcpy(hitOut, hit.ptr, n, hipMemcpyDeviceToHost));
    if (tOut) RF_HIP(hipMemcpy(tOut, t.ptr, n * 4, hipMemcpyDeviceToHost));
    if (triangleTestsOut) RF_HIP(hipMemcpy(triangleTestsOut, tt.ptr, n * 4, hipMemcpyDeviceToHost));
}

void Renderer::intersectRays(const float* rays6, uint64_t numRays, float tMax, uint32_t* triangleOut, float* tOut, float* uvOut, float* pOut,
                             uint32_t* nodesVisitedOut, uint32_t* triangleTestsOut)
{
    Impl& m = *mImpl;
    synchronize();
    if (numRays == 0) return;
    if (m.queryVariant == 2)
    {
        std::vector<float4> hit, p4;
        m.queryWide(rays6, numRays, tMax, false, hit, p4);
        for (uint64_t i = 0; i < numRays; ++i)
        {
            const uint32_t tri = floatBits(hit[i].x);
            const bool     found = tri != kMiss;
            triangleOut[i] = tri;
            if (tOut) tOut[i] = found ? hit[i].w : 0.0f;
            if (uvOut) uvOut[2 * i] = found ? hit[i].y : 0.0f, uvOut[2 * i + 1] = found ? hit[i].z : 0.0f;
            if (pOut) pOut[3 * i] = found ? p4[i].x : 0.0f, pOut[3 * i + 1] = found ? p4[i].y : 0.0f, pOut[3 * i + 2] = found ? p4[i].z : 0.0f;
            if (nodesVisitedOut) nodesVisitedOut[i] = 0;
            if (triangleTestsOut) triangleTestsOut[i] = 0;
        }
        return;
    }
    DeviceBuffer<float>    rays, t, uv, p;
    DeviceBuffer<uint32_t> tri, nv, tt;
    rays.upload(rays6, 6 * numRays);
    tri.alloc(numRays);
    t.alloc(numRays);
    uv.alloc(2 * numRays);
    p.alloc(3 * numRays);
    nv.alloc(numRays);
    tt.alloc(numRays);
    hipLaunchKernelGGL(intersectRaysKernel(), dim3(static_cast<uint32_t>((numRays + kBlock - 1) / kBlock)), dim3(kBlock), 0, m.stream, m.scene, rays.ptr,
                       numRays, tMax, tri.ptr, t.ptr, uv.ptr, p.ptr, nv.ptr, tt.ptr, m.counters.ptr);
    RF_HIP(hipGetLastError());
    RF_HIP(hipStreamSynchronize(m.stream));
    RF_HIP(hipMemcpy(triangleOut, tri.ptr, numRays * 4, hipMemcpyDeviceToHost));
    if (tOut) RF_HIP(hipMemcpy(tOut, t.ptr, numRays * 4, hipMemcpyDeviceToHost));
    if (uvOut) RF_HIP(hipMemcpy(uvOut, uv.ptr, numRays * 8, hipMemcpyDeviceToHost));
    if (pOut) RF_HIP(hipMemcpy(pOut, p.ptr, numRays * 12, hipMemcpyDeviceToHost));
    if (nodesVisitedOut) RF_HIP(hipMemcpy(nodesVisitedOut, nv.ptr, numRays * 4, hipMemcpyDeviceToHost));
    if (triangleTestsOut) RF_HIP(hipMemcpy(triangleTestsOut, tt.ptr, numRays * 4, hipMemcpyDeviceToHost));
}

void Renderer::occludedRays(const float* rays6, uint64_t numRays, float tMax, float* visibilityOut)
{
    Impl& m = *mImpl;
    synchronize();
    if (numRays == 0) return;
    if (m.queryVariant == 2)
    {
        std::vector<float4> rad, unused;
        m.queryWide(rays6, numRays, tMax, true, rad, unused);
        for (uint64_t i = 0; i < numRays; ++i) visibilityOut[i] = rad[i].x != 0.0f ? 1.0f : 0.0f;
        return;
    }
    DeviceBuffer<float> rays, vis;
    rays.upload(rays6, 6 * numRays);
    vis.alloc(numRays);
    hipLaunchKernelGGL(occludedRaysKernel(), dim3(static_cast<uint32_t>((numRays + kBlock - 1) / kBlock)), dim3(kBlock), 0, m.stream, m.scene, rays.ptr,
                       numRays, tMax, vis.ptr, m.counters.ptr);
    RF_HIP(hipGetLastError());
    RF_HIP(hipStreamSynchronize(m.stream));
    RF_HIP(hipMemcpy(visibilityOut, vis.ptr, numRays * 4, hipMemcpyDeviceToHost));
}
uint32_t checkWideLayouts(std::span<const BvhNode> nodes, float* quadHalfAreaRatio)
{
    if (nodes.empty()) throw std::runtime_error("checkWideLayouts: no nodes");
    const WideBuild wb = buildWide(nodes.data(), nodes.size());
    if (quadHalfAreaRatio) *quadHalfAreaRatio = wb.quadHalf.empty() ? 0.0f : wb.quadHalfAreaRatio;
    const uint32_t  flags = (wb.boxesRegular ? 1u : 0u) | (!wb.compact.empty() ? 2u : 0u) | (!wb.hot.empty() ? 4u : 0u) | (!wb.quad.empty() ? 8u : 0u) | (!wb.quadHalf.empty() ? 16u : 0u) | (!wb.quadLocal.empty() ? 32u : 0u) | (!wb.oct.empty() ? 64u : 0u);
    const size_t    records = wb.nodes.size() / 4;
    const auto      fail = [](size_t r, const char* what) { throw std::runtime_error("wide layout mismatch at record " + std::to_string(r) + ": " + what); };
    if (!wb.quadHalf.empty())
    {
        // The half-precision quad records: same words as the f32 quad records, every binary16 plane on the conservative side of the f32
        // plane by at least the margin of the proof (rf_wide.hpp), never subnormal, finite.
        if (wb.quadHalf.size() * 2 != wb.quad.size()) throw std::runtime_error("wide layout mismatch: half-precision quad records do not pair up with the quad records");
        double R = 0.0;
        for (const float c : {wb.rootLo.x, wb.rootLo.y, wb.rootLo.z, wb.rootHi.x, wb.rootHi.y, wb.rootHi.z}) R = std::max(R, static_cast<double>(std::fabs(c)));
        const double margin = 1.1920928955078125e-07 * (4.0 * static_cast<double>(wb.originBound) + 3.0 * R) * 1.0001; // what the proof needs (the builder leaves twice that)
        if (!(static_cast<double>(wb.originBound) >= 4.0 * R)) throw std::runtime_error("wide layout mismatch: origin bound of the half-precision quad records");
        for (size_t r = 0; r < wb.quadHalf.size() / 4; ++r)
        {
            const float4* q = &wb.quad[8 * r];
            const uint4*  h = &wb.quadHalf[4 * r];
            const uint32_t d[16] = {h[0].x, h[0].y, h[0].z, h[0].w, h[1].x, h[1].y, h[1].z, h[1].w, h[2].x, h[2].y, h[2].z, h[2].w, h[3].x, h[3].y, h[3].z, h[3].w};
            if (d[12] != floatBits(q[6].x) || d[13] != floatBits(q[6].y) || d[14] != floatBits(q[6].z) || d[15] != floatBits(q[6].w)) fail(r, "half-precision quad record: words differ");
            for (int k = 0; k < 2; ++k)
            {
                const float4 a = q[3 * k], z = q[3 * k + 1], b = q[3 * k + 2];
                const float  lo[2][3] = {{a.x, a.y, z.x}, {b.x, b.y, z.z}}, hi[2][3] = {{a.z, a.w, z.y}, {b.z, b.w, z.w}};
                for (int j = 0; j < 2; ++j)
                    for (int ax = 0; ax < 3; ++ax)
                    {
                        const uint32_t w = d[3 * (2 * k + j) + ax];
                        if (d[12 + 2 * k + j] == kQuadEmpty)
                        {
                            // (an empty slot holds the inverted box that no ray passes: the traversal kernel does not test its word)
                            if (w != kHalfEmptyPlanes) fail(r, "half-precision quad record: an empty slot does not hold the inverted box");
                            continue;
                        }
                        const uint16_t l16 = static_cast<uint16_t>(w & 0xFFFFu), h16 = static_cast<uint16_t>(w >> 16);
                        for (const uint16_t v : {l16, h16})
                            if (((v >> 10) & 0x1Fu) == 0x1Fu || (((v >> 10) & 0x1Fu) == 0u && (v & 0x3FFu) != 0u)) fail(r, "half-precision quad record: a plane is not a normal number or zero");
                        if (!(static_cast<double>(halfBitsToFloat(l16)) <= static_cast<double>(lo[j][ax]) - margin)) fail(r, "half-precision quad record: a lower plane is not below its f32 plane by the margin");
                        if (!(static_cast<double>(halfBitsToFloat(h16)) >= static_cast<double>(hi[j][ax]) + margin)) fail(r, "half-precision quad record: an upper plane is not above its f32 plane by the margin");
                    }
            }
        }
    }
    if (!wb.quadLocal.empty())
    {
        // The local-grid quad records: same words, power-of-two scales, every decoded plane (anchor + byte * scale, exact in double) on the
        // conservative side of the f32 plane by at least the margin of the proof (rf_wide.hpp).
        if (wb.quadLocal.size() * 2 != wb.quad.size()) throw std::runtime_error("wide layout mismatch: local-grid quad records do not pair up with the quad records");
        double R = 0.0;
        for (const float c : {wb.rootLo.x, wb.rootLo.y, wb.rootLo.z, wb.rootHi.x, wb.rootHi.y, wb.rootHi.z}) R = std::max(R, static_cast<double>(std::fabs(c)));
        for (size_t r = 0; r < wb.quadLocal.size() / 4; ++r)
        {
            const float4*  q = &wb.quad[8 * r];
            const uint4*   l = &wb.quadLocal[4 * r];
            const uint32_t words[4] = {l[3].x, l[3].y, l[3].z, l[3].w};
            if (words[0] != floatBits(q[6].x) || words[1] != floatBits(q[6].y) || words[2] != floatBits(q[6].z) || words[3] != floatBits(q[6].w)) fail(r, "local-grid quad record: words differ");
            const float    anchor[3] = {bitsFloat(l[0].x), bitsFloat(l[0].y), bitsFloat(l[0].z)}, scale[3] = {bitsFloat(l[0].w), bitsFloat(l[1].x), bitsFloat(l[1].y)};
            const uint32_t axisWords[3][2] = {{l[1].z, l[1].w}, {l[2].x, l[2].y}, {l[2].z, l[2].w}};
            for (int ax = 0; ax < 3; ++ax)
            {
                int          ex = 0;
                const double m = std::frexp(static_cast<double>(scale[ax]), &ex);
                if (!(m == 0.5) || !std::isfinite(anchor[ax])) fail(r, "local-grid quad record: scale is not a power of two, or the anchor is not finite");
                const double margin = 5.9604644775390625e-08 * (6.03 * (static_cast<double>(wb.originBound) + R) + 1024.0 * static_cast<double>(scale[ax])) * 1.0001;
                for (int e = 0; e < 4; ++e)
                {
                    if (words[e] == kQuadEmpty) continue;
                    const int    k = e / 2, j = e % 2;
                    const float4 a = q[3 * k], z = q[3 * k + 1], b = q[3 * k + 2];
                    const double lo = ax == 0 ? (j ? b.x : a.x) : ax == 1 ? (j ? b.y : a.y) : (j ? z.z : z.x), hi = ax == 0 ? (j ? b.z : a.z) : ax == 1 ? (j ? b.w : a.w) : (j ? z.w : z.y);
                    const uint32_t pair = (axisWords[ax][e / 2] >> (16 * (e % 2))) & 0xFFFFu;
                    const double   dlo = static_cast<double>(anchor[ax]) + static_cast<double>(pair & 0xFFu) * static_cast<double>(scale[ax]),
                                 dhi = static_cast<double>(anchor[ax]) + static_cast<double>(pair >> 8) * static_cast<double>(scale[ax]);
                    if (!(dlo <= lo - margin)) fail(r, "local-grid quad record: a lower plane is not below its f32 plane by the margin");
                    if (!(dhi >= hi + margin)) fail(r, "local-grid quad record: an upper plane is not above its f32 plane by the margin");
                }
            }
        }
    }
    if (!wb.quad.empty())
    {
        // The leaf boxes and occluder-cache entries leafBoxesIntoTriangles writes into the triangle records (round 4): every leaf's box is its node's; an entry is 0
        // ("the leaf itself") or the index of a quad record from which the leaf is reached within `levels` steps -- in range, and really above THAT leaf.
        size_t numTriangles = 0;
        for (const BvhNode& n : nodes)
            if (n.triangleCount != 0) numTriangles = std::max(numTriangles, static_cast<size_t>(n.trianglesOffset) + n.triangleCount);
        const size_t numQuad = wb.quad.size() / 8;
        for (uint32_t levels = 0; levels <= 3; ++levels)
        {
            std::vector<float4> tri(4 * numTriangles, make_float4(0.0f, 0.0f, 0.0f, 0.0f));
            const bool          distinct = leafBoxesIntoTriangles(nodes.data(), nodes.size(), tri.data(), numTriangles, levels, &wb.quadIndexOfNode);
            if (!distinct) break; // (leaves sharing a first triangle: the renderer keeps the layouts that read these slots off)
            for (size_t i = 0; i < nodes.size(); ++i)
            {
                const BvhNode& n = nodes[i];
                if (n.triangleCount == 0) continue;
                const float4* t = &tri[4 * static_cast<size_t>(n.trianglesOffset)];
                if (floatBits(t[0].w) != floatBits(n.aabb.min.x) || floatBits(t[1].w) != floatBits(n.aabb.min.y) || floatBits(t[2].w) != floatBits(n.aabb.min.z) ||
                    floatBits(t[3].x) != floatBits(n.aabb.max.x) || floatBits(t[3].y) != floatBits(n.aabb.max.y) || floatBits(t[3].z) != floatBits(n.aabb.max.z))
                    fail(i, "leaf box in the triangle record differs from the node's");
                const uint32_t hint = floatBits(t[3].w);
                if (levels == 0 && hint != 0u) fail(i, "occluder-cache entry of a leaf: not 0 at level 0");
                if (hint == 0u) continue;
                if (hint >= numQuad) fail(i, "occluder-cache entry of a leaf: quad record index out of range");
                std::vector<uint32_t> frontier{hint};
                bool                  found = false;
                for (uint32_t l = 0; l < levels && !found; ++l)
                {
                    std::vector<uint32_t> next;
                    for (const uint32_t r : frontier)
                    {
                        const float4*  q = &wb.quad[8 * static_cast<size_t>(r)];
                        const uint32_t raw[4] = {floatBits(q[6].x), floatBits(q[6].y), floatBits(q[6].z), floatBits(q[6].w)};
                        for (int e = 0; e < 4; ++e)
                        {
                            if (raw[e] == kQuadEmpty) continue;
                            const uint32_t w = e == 2 ? raw[e] : raw[e] & ~(3u << kWideAxisShift); // (entries 0, 1 and 3 carry split axes)
                            if ((w & kWideLeafBit) == 0u)
                            {
                                if (w >= numQuad) fail(i, "occluder-cache entry of a leaf: a record below it names a record out of range");
                                next.push_back(w);
                                continue;
                            }
                            uint32_t first = w & ((1u << kWideIndexBits) - 1u);
                            if (((w >> kWideIndexBits) & 7u) == 7u) first = wb.bigLeaves[first].x;
                            if (first == n.trianglesOffset) found = true;
                        }
                    }
                    frontier.swap(next);
                }
                if (!found) fail(i, "occluder-cache entry of a leaf: the leaf is not below the record it names");
            }
        }
    }
    if (!wb.quad.empty())
    {
        // Walk the quad records from the root next to the 64-byte records: the entries of a quad record must be the children of
        // the children the plain record of the same node names (a leaf child filling one slot), with the same leaf words, the
        // split axes of both levels, and the skipped child's box must be the union of its entries (what makes the skip exact).
        struct Pair
        {
            uint32_t plain, quad;
        };
        std::vector<Pair> todo{{0u, 0u}};
        size_t            visited = 0;
        while (!todo.empty())
        {
            const Pair at = todo.back();
            todo.pop_back();
            ++visited;
            if (at.plain >= records || 8 * static_cast<size_t>(at.quad) + 8 > wb.quad.size()) fail(at.plain, "quad record index out of range");
            const float4*  w = &wb.nodes[4 * static_cast<size_t>(at.plain)];
            const float4*  q = &wb.quad[8 * static_cast<size_t>(at.quad)];
            const uint32_t qw[4] = {floatBits(q[6].x), floatBits(q[6].y), floatBits(q[6].z), floatBits(q[6].w)};
            const uint32_t pw[2] = {floatBits(w[3].x) & ~(3u << kWideAxisShift), floatBits(w[3].y)};
            if (((qw[0] >> kWideAxisShift) & 3u) != ((floatBits(w[3].x) >> kWideAxisShift) & 3u)) fail(at.plain, "quad record: split axis of the node differs");
            const float cbox[2][6] = {{w[0].x, w[0].y, w[0].z, w[0].w, w[1].x, w[1].y}, {w[2].x, w[2].y, w[2].z, w[2].w, w[1].z, w[1].w}};
            for (int k = 0; k < 2; ++k)
            {
                const float4*  e = q + 3 * k;
                const float    b0[6] = {e[0].x, e[0].y, e[0].z, e[0].w, e[1].x, e[1].y}, b1[6] = {e[2].x, e[2].y, e[2].z, e[2].w, e[1].z, e[1].w};
                const uint32_t wa = qw[2 * k], wb2 = qw[2 * k + 1];
                if (pw[k] & kWideLeafBit)
                {
                    // a leaf child fills slot 2k with itself
                    if ((wa & ~(3u << kWideAxisShift)) != pw[k] || wb2 != kQuadEmpty) fail(at.plain, "quad record: leaf child not passed through");
                    for (int j = 0; j < 6; ++j)
                        if (!(b0[j] == cbox[k][j])) fail(at.plain, "quad record: leaf child's box differs");
                    continue;
                }
                const float4*  cw = &wb.nodes[4 * static_cast<size_t>(pw[k])]; // the child's own plain record = its children
                const float    g0[6] = {cw[0].x, cw[0].y, cw[0].z, cw[0].w, cw[1].x, cw[1].y}, g1[6] = {cw[2].x, cw[2].y, cw[2].z, cw[2].w, cw[1].z, cw[1].w};
                const uint32_t gw[2] = {floatBits(cw[3].x) & ~(3u << kWideAxisShift), floatBits(cw[3].y)};
                const uint32_t childAxis = (floatBits(cw[3].x) >> kWideAxisShift) & 3u;
                for (int j = 0; j < 6; ++j)
                {
                    if (!(b0[j] == g0[j]) || !(b1[j] == g1[j])) fail(at.plain, "quad record: grandchild box differs");
                    const float u = (j == 0 || j == 1 || j == 4) ? std::min(g0[j], g1[j]) : std::max(g0[j], g1[j]);
                    if (!(u == cbox[k][j])) fail(at.plain, "quad record: skipped child's box is not the union of its children's");
                }
                const uint32_t tagged = k == 0 ? wb2 : wb2; // slot 2k+1 carries the child's split axis
                if (((tagged >> kWideAxisShift) & 3u) != childAxis) fail(at.plain, "quad record: split axis of a child differs");
                const uint32_t ea = k == 0 ? (wa & ~(3u << kWideAxisShift)) : wa, eb = wb2 & ~(3u << kWideAxisShift);
                const uint32_t ent[2] = {ea, eb};
                for (int j = 0; j < 2; ++j)
                {
                    if (gw[j] & kWideLeafBit)
                    {
                        if (ent[j] != gw[j]) fail(at.plain, "quad record: grandchild leaf word differs");
                    }
                    else
                    {
                        if (ent[j] & kWideLeafBit) fail(at.plain, "quad record: interior grandchild encoded as a leaf");
                        todo.push_back(Pair{gw[j], ent[j]});
                    }
                }
            }
        }
        if (visited != wb.quad.size() / 8) fail(0, "quad records: not every record is reachable from the root exactly once");
    }
    if (!wb.oct.empty())
    {
        // Oct records (round 5): walk them from the root next to the reference nodes.  Slot e = 4 c + 2 g + k must hold what the three levels below the member hold
        // at that place (a leaf fills the first slot of its range), every plane must lie outside its f32 plane by the margin, an empty slot must fail for every
        // ray (lo = 255, hi = 0), and for each of the eight direction-sign patterns the positions the KERNEL derives from the order table (kTraceWide, COMPACT == 6)
        // must put the slots into the order in which the reference's walk (wgsl:409-417 at each of the three levels) reaches them.
        double R = 0.0;
        for (const float c : {wb.rootLo.x, wb.rootLo.y, wb.rootLo.z, wb.rootHi.x, wb.rootHi.y, wb.rootHi.z}) R = std::max(R, static_cast<double>(std::fabs(c)));
        const double margin = (static_cast<double>(wb.originBound) + R) * 3.814697265625e-06 * 0.999; // 2^-18 (a hair less: the builder's own rounding)
        const size_t numOct = wb.oct.size() / 8;
        struct Item
        {
            uint32_t node, rec;
        };
        std::vector<Item> todo{{0u, 0u}};
        size_t            visited = 0;
        while (!todo.empty())
        {
            const Item at = todo.back();
            todo.pop_back();
            ++visited;
            if (at.rec >= numOct || at.node >= nodes.size() || nodes[at.node].triangleCount != 0) fail(at.rec, "oct record: index out of range or not an interior node");
            uint32_t slotNode[8];
            for (uint32_t& v : slotNode) v = kQuadEmpty;
            const std::function<void(uint32_t, int, int)> place = [&](uint32_t nd, int depth, int base) {
                if (depth == 3 || (depth > 0 && nodes[nd].triangleCount != 0)) { slotNode[base] = nd; return; }
                place(nd + 1, depth + 1, base);
                place(nodes[nd].secondChildOffset, depth + 1, base + (4 >> depth));
            };
            place(at.node, 0, 0);
            const uint4*   o = &wb.oct[8 * static_cast<size_t>(at.rec)];
            const float    anchor[3] = {bitsFloat(o[0].x), bitsFloat(o[0].y), bitsFloat(o[0].z)}, scale[3] = {bitsFloat(o[0].w), bitsFloat(o[1].x), bitsFloat(o[1].y)};
            const uint32_t words[8] = {o[5].x, o[5].y, o[5].z, o[5].w, o[6].x, o[6].y, o[6].z, o[6].w};
            const uint32_t axisWords[3][4] = {{o[2].x, o[2].y, o[2].z, o[2].w}, {o[3].x, o[3].y, o[3].z, o[3].w}, {o[4].x, o[4].y, o[4].z, o[4].w}};
            for (int e = 0; e < 8; ++e)
            {
                for (int ax = 0; ax < 3; ++ax)
                {
                    const uint32_t pair = (axisWords[ax][e / 2] >> (16 * (e % 2))) & 0xFFFFu;
                    if (slotNode[e] == kQuadEmpty)
                    {
                        if (pair != 0x00FFu) fail(at.rec, "oct record: an empty slot does not hold the planes no ray passes");
                        continue;
                    }
                    const BvhNode& en = nodes[slotNode[e]];
                    const double   lo = ax == 0 ? en.aabb.min.x : ax == 1 ? en.aabb.min.y : en.aabb.min.z, hi = ax == 0 ? en.aabb.max.x : ax == 1 ? en.aabb.max.y : en.aabb.max.z;
                    const double   dlo = static_cast<double>(anchor[ax]) + static_cast<double>(pair & 0xFFu) * static_cast<double>(scale[ax]),
                                 dhi = static_cast<double>(anchor[ax]) + static_cast<double>(pair >> 8) * static_cast<double>(scale[ax]);
                    if (!(dlo <= lo - margin)) fail(at.rec, "oct record: a lower plane is not below its f32 plane by the margin");
                    if (!(dhi >= hi + margin)) fail(at.rec, "oct record: an upper plane is not above its f32 plane by the margin");
                }
                if (slotNode[e] == kQuadEmpty)
                {
                    if (words[e] != kQuadEmpty) fail(at.rec, "oct record: an empty slot names something");
                    continue;
                }
                const BvhNode& en = nodes[slotNode[e]];
                if (en.triangleCount != 0)
                {
                    if ((words[e] & kWideLeafBit) == 0u) fail(at.rec, "oct record: a leaf encoded as an interior entry");
                    uint32_t first = words[e] & ((1u << kWideIndexBits) - 1u), cnt = ((words[e] >> kWideIndexBits) & 7u) + 1u;
                    if (cnt == 8u) cnt = wb.bigLeaves[first].y, first = wb.bigLeaves[first].x;
                    if (first != en.trianglesOffset || cnt != en.triangleCount) fail(at.rec, "oct record: leaf word names other triangles");
                }
                else
                {
                    if ((words[e] & kWideLeafBit) != 0u || words[e] != wb.octIndexOfNode[slotNode[e]]) fail(at.rec, "oct record: interior entry names another record");
                    todo.push_back(Item{slotNode[e], words[e]});
                }
            }
            const unsigned long long table = (static_cast<unsigned long long>(o[1].w) << 32) | o[1].z;
            for (uint32_t signs = 0; signs < 8; ++signs)
            {
                std::vector<int>                              want; // slots in the order the reference reaches them
                const std::function<void(uint32_t, int, int)> walk = [&](uint32_t nd, int depth, int base) {
                    if (depth == 3 || (depth > 0 && nodes[nd].triangleCount != 0)) { want.push_back(base); return; }
                    const bool     neg = ((signs >> (nodes[nd].splitAxis & 3u)) & 1u) != 0u;
                    const uint32_t kid[2] = {nd + 1, nodes[nd].secondChildOffset};
                    const int      kidBase[2] = {base, base + (4 >> depth)};
                    for (int k = 0; k < 2; ++k) walk(kid[neg ? 1 - k : k], depth + 1, kidBase[neg ? 1 - k : k]);
                };
                walk(at.node, 0, 0);
                // the kernel's rule (refill + oct step)
                const uint32_t signXY = signs & 3u, negZ = signs >> 2;
                const uint32_t octKey = negZ ? ((16u * (3u - signXY)) | (0x7777u << 8)) : 16u * signXY;
                const uint32_t ord = static_cast<uint32_t>(table >> (octKey & 63u)) ^ (octKey >> 8);
                int            slotAt[8];
                for (int& v : slotAt) v = -1;
                for (int j = 0; j < 4; ++j)
                {
                    const uint32_t p0 = (ord >> (4 * j)) & 7u;
                    if (slotNode[2 * j] != kQuadEmpty) slotAt[p0] = 2 * j;
                    if (slotNode[2 * j + 1] != kQuadEmpty) slotAt[p0 ^ 1u] = 2 * j + 1;
                }
                std::vector<int> got;
                for (const int v : slotAt)
                    if (v >= 0) got.push_back(v);
                if (got != want) fail(at.rec, "oct record: the order table does not reproduce the reference's visit order");
            }
        }
        if (visited != numOct) fail(0, "oct records: not every record is reachable from the root exactly once");
    }
    for (size_t r = 0; r < records; ++r)
    {
        const float4* w = &wb.nodes[4 * r];
        // child planes of the plain record, in the order {lo.x lo.y hi.x hi.y lo.z hi.z}
        const float    c0[6] = {w[0].x, w[0].y, w[0].z, w[0].w, w[1].x, w[1].y}, c1[6] = {w[2].x, w[2].y, w[2].z, w[2].w, w[1].z, w[1].w};
        const uint32_t word0 = floatBits(w[3].x), word1 = floatBits(w[3].y);
        float          own[6]; // the node's own box: the union of its children's (lo: min, hi: max)
        for (int k = 0; k < 6; ++k) own[k] = (k == 0 || k == 1 || k == 4) ? std::min(c0[k], c1[k]) : std::max(c0[k], c1[k]);
        if (!wb.compact.empty())
        {
            const float4*  c = &wb.compact[4 * r];
            const uint32_t cw0 = floatBits(c[2].x), cw1 = floatBits(c[2].z);
            const bool     selLo = (cw1 >> kWideAxisShift) & 1u, selHi = (cw1 >> (kWideAxisShift + 1)) & 1u;
            const float    d0[6] = {selLo ? c[3].x : c[0].x, c[0].y, selHi ? c[3].y : c[0].z, c[0].w, c[1].x, c[1].y};
            const float    d1[6] = {selLo ? c[0].x : c[3].x, c[2].y, selHi ? c[0].z : c[3].y, c[2].w, c[1].z, c[1].w};
            for (int k = 0; k < 6; ++k)
                if (!(d0[k] == c0[k]) || !(d1[k] == c1[k])) fail(r, "compact-capable record decodes to other planes");
            if (cw0 != word0 || (cw1 & ~(3u << kWideAxisShift)) != word1) fail(r, "compact-capable record holds other child words");
            if (!(c[3].x == own[0]) || !(c[3].y == own[2])) fail(r, "compact-capable record: outer x planes are not the node's");
        }
        if (!wb.hot.empty())
        {
            const float4*  h = &wb.hot[2 * r];
            const float4*  o = &wb.own[2 * r];
            const uint32_t hw0 = floatBits(h[1].z), hw1 = floatBits(h[1].w);
            const float    inner[6] = {h[0].x, h[0].y, h[0].z, h[0].w, h[1].x, h[1].y}, outer[6] = {o[0].x, o[0].y, o[0].z, o[0].w, o[1].x, o[1].y};
            const bool     sel[6] = {((hw0 >> 25) & 1u) != 0u, ((hw0 >> 24) & 1u) != 0u, ((hw1 >> 25) & 1u) != 0u,
                                     ((hw1 >> 24) & 1u) != 0u, ((hw1 >> 30) & 1u) != 0u, ((hw1 >> 29) & 1u) != 0u};
            for (int k = 0; k < 6; ++k)
            {
                if (!((sel[k] ? outer[k] : inner[k]) == c0[k]) || !((sel[k] ? inner[k] : outer[k]) == c1[k])) fail(r, "32-byte record decodes to other planes");
                if (!(outer[k] == own[k])) fail(r, "own-box array does not hold the union of the children");
            }
            if ((hw0 & ~(3u << 24)) != word0 || (hw1 & ~((3u << 24) | (3u << kWideAxisShift))) != word1) fail(r, "32-byte record holds other child words");
        }
    }
    return flags;
}
} // namespace rf

