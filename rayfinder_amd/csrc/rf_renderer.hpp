// rf_renderer.hpp -- host interface of the MI355X wavefront path tracer.
//
// Mirrors the reference's renderer seam (src/pt/reference_path_tracer.hpp:26-76):
//   ReferencePathTracer(desc, gpu, scene)  -> Renderer(desc, scene)      copies the scene to HBM
//   setRenderParameters(params)            -> setRenderParameters        resets accumulation on change
//   render(...)  (one sample per call)     -> render(numFrames)          n calls without host sync
//   averageRenderpassDurationMs()          -> averageRenderpassDurationMs (30-deep moving average)
//   renderProgressPercentage()             -> renderProgressPercentage
// plus what an offline renderer needs and the reference never exposed: read-back of the float
// accumulation buffer, the tonemapped image, per-kernel statistics, tile sharding for multi-GPU,
// and the bvh-visualizer node-visit pass (src/bvh-visualizer/main.cpp:60-78) on the GPU.
#pragma once

#include "rf_sky.hpp"
#include "rf_types.hpp"

#include <cstdint>
#include <deque>
#include <memory>
#include <span>
#include <string>
#include <vector>

namespace rf
{
struct SamplingParams
{
    uint32_t numSamplesPerPixel = 128;
    uint32_t numBounces = 4;
    bool     operator==(const SamplingParams&) const = default;
};

struct RenderParameters
{
    uint32_t       width = 0, height = 0;
    Camera         camera{};
    SamplingParams samplingParams;
    Sky            sky;
    float          exposure = 1.0f;
};
bool operator==(const RenderParameters& a, const RenderParameters& b);

struct TextureView
{
    const uint32_t* pixels;
    uint32_t        width, height;
};

struct SceneView
{
    std::span<const BvhNode>           bvhNodes;
    std::span<const PositionAttribute> positionAttributes;
    std::span<const VertexAttributes>  vertexAttributes;
    std::span<const TextureView>       baseColorTextures;
};

struct RendererDescriptor
{
    RenderParameters renderParams;
    uint32_t         maxWidth = 0, maxHeight = 0;
    int              deviceOrdinal = 0;
    // Paths kept in flight per batch (several samples of every pixel are traced together so that
    // deep bounces still fill the chip). 0 = default (about 8M).
    uint64_t maxPathsInFlight = 0;
};

// Ray/traversal statistics since the last resetStats().  Node visits and triangle tests are only
// counted while counting is enabled (it selects the counting build of the traversal kernels).
struct RenderStats
{
    uint64_t primaryRays = 0, closestRays = 0, shadowRays = 0;
    uint64_t closestNodeVisits = 0, closestTriangleTests = 0;
    uint64_t shadowNodeVisits = 0, shadowTriangleTests = 0;
    uint64_t paths = 0;
    uint32_t stackHighWater = 0;
    // batch depth actually used (round 6): samples of every pixel traced together in the MOST RECENT batch, paths in it, and batches traced since the last reset.  A render call
    // that does not get the configured depth (device memory short: another handle, a co-tenant) traces the same samples in more, shallower batches -- same image, shorter
    // launches: a loss of speed that used to be said on stderr only
    uint32_t batchSamplesUsed = 0, batchesTraced = 0;
    uint64_t batchPathsUsed = 0;
    uint64_t closestRecordFetches = 0, shadowRecordFetches = 0; // 64-B BVH records fetched (counting build)
    uint64_t abandonedRays = 0;  // rays whose traversal stack outgrew 96 entries (reference: undefined past 32); every build
    uint64_t scalarRedoRays = 0; // rays redone by the reference-ordered scalar traversal (irregular rays, LDS stack overflow); every build
    uint64_t shadowRaysSelfAnswered = 0; // of shadowRays: stopped by the triangle they start on, found by kShade's own test (kShadeSelfShadow); never queued for an any-hit launch
    uint64_t shadowRaysHintAnswered = 0; // of shadowRays: answered by kShadowHint at the leaf the occluder grid named (never entered the traversal); every build
    // hipEvent-timed kernel time (ms) and launch counts, per kernel class, while timing is enabled
    double   msRaygen = 0, msClosest = 0, msShade = 0, msShadow = 0, msAccumulate = 0;
    uint32_t launchesRaygen = 0, launchesClosest = 0, launchesShade = 0, launchesShadow = 0, launchesAccumulate = 0;
    // per bounce (index b = bounce b+1; bounces past kMaxBounceStats are folded into the last entry):
    // queue occupancy = rays traced, and kernel time while timing is enabled
    static constexpr uint32_t kMaxBounceStats = 32;
    uint64_t closestRaysByBounce[kMaxBounceStats] = {}, shadowRaysByBounce[kMaxBounceStats] = {};
    double   msClosestByBounce[kMaxBounceStats] = {}, msShadowByBounce[kMaxBounceStats] = {};
};

constexpr uint32_t kTileSize = 32; // shard tile edge in pixels (32x32 = 16 waves of 8x8 pixels)

// Deterministic tile -> rank assignment (tiles along a Z-order curve dealt round-robin: equal counts +-1, every compact block of the image split over all ranks).
std::vector<uint32_t> tilesForRank(uint32_t width, uint32_t height, uint32_t rank, uint32_t worldSize);
// compact tile-major float4 buffer -> row-major width*height*4 image (pixels of other ranks' tiles untouched)
void untileHost(const float* compact, const uint32_t* tileIds, uint32_t numTiles, uint32_t width, uint32_t height, float* image);

class Renderer
{
public:
    Renderer(const RendererDescriptor& desc, const SceneView& scene);
    ~Renderer();
    Renderer(const Renderer&) = delete;
    Renderer& operator=(const Renderer&) = delete;

    void  setRenderParameters(const RenderParameters& params);
    void  render(uint32_t numFrames);
    float averageRenderpassDurationMs() const;
    float renderProgressPercentage() const;

    // Multi-GPU: render only this rank's tiles. Resets accumulation.
    void setTileShard(uint32_t rank, uint32_t worldSize);
    std::span<const uint32_t> shardTiles() const;

    uint32_t accumulatedSampleCount() const;
    uint32_t width() const;
    uint32_t height() const;
    uint32_t shardRank() const;
    uint32_t shardWorldSize() const;
    int      deviceOrdinal() const;
    // The handle's HIP stream (a hipStream_t): work a caller wants ordered behind the frame's kernels (the RCCL
    // frame exchange, rf_comm.hpp) is enqueued here.
    void* streamHandle() const;
    // Row-major width*height*4 floats (sum of samples, 16-B stride as the reference's
    // array<vec3f>); pixels outside this rank's tiles are zero.
    void readAccumulation(float* dst);
    // Device pointer of the compact tile-major accumulation buffer (numTiles*1024 float4) and a
    // way to render into caller-owned device memory (e.g. a torch tensor used for the RCCL gather).
    void*    accumulationDevicePointer() const;
    // Zero the accumulation buffer (on the handle's stream) if nothing has been rendered into it since the last reset, so that a
    // reader on the device (the frame exchange) never sees the previous frame's sums.
    void     clearAccumulationIfStale();
    // Device memory held: path state + queues (allocated on demand, kBytesPerPath per path slot), the batch depth in use, and
    // the resident scene (BVH layouts, triangles, shading records, textures).
    // rf_renderer_layout_info: layouts[0..15] = closest-hit, [16..31] = any-hit, [32..47] = any-hit launch starts at the occluder cache; misc = {hint levels, first look from bounce,
    // dense leaf min, legacy build}
    void     layoutInfo(uint32_t (&layouts)[48], uint32_t (&misc)[4], float& quadHalfAreaRatio, uint64_t& treeBytes) const;
    void     memoryInfo(uint64_t& pathStateBytes, uint64_t& pathsAllocated, uint64_t& maxPathsPerBatch, uint64_t& sceneBytes) const;
    uint64_t accumulationBytes() const;
    void     bindAccumulationBuffer(void* devicePtr, uint64_t bytes);
    // BGRA8 swap-chain image (wgsl:59-63), row-major.
    void readTonemapped(uint32_t* dstBgra8);
    // The same display transform for any row-major float4 SUM image in device memory (e.g. the frame a gather
    // assembled on the root rank): numPixels texels, divided by `samples`, scaled by the handle's exposure.
    void tonemapDeviceImage(const void* imageDevice, uint64_t numPixels, uint32_t samples, uint32_t* dstBgra8Host);

    // Deferred-lighting variant: numFrames frames of lighting pass + exponential resolve
    // (src/pt/deferred_renderer_lighting_pass.wgsl:96-186, deferred_renderer_resolve_pass.wgsl:33-54) over a
    // primary-ray G-buffer; uses the handle's camera, sky and exposure.  Its frame counter starts at 0 (frame 0
    // initialises the accumulation, resolve_pass.wgsl:41-44) and is independent of render()'s.
    void     renderDeferred(uint32_t numFrames);
    void     resetDeferred();
    uint32_t deferredFrameCount() const;
    // sampleBuffer / accumulationBuffer (width*height*3 floats, row-major) and the resolve pass's BGRA8 output; NULL = skip
    void readDeferred(float* sampleRgb, float* accumulationRgb, uint32_t* bgra8);

    void        setCounting(bool enabled);
    // Tuning knobs for A/B measurements inside one process ("traversal_variant": 0 = one ray per
    // thread kernels, 1 = persistent waves with lane refill).  Results never depend on them.
    void        setOption(const std::string& name, int64_t value);
    uint32_t    numBounces() const;
    void        setTiming(bool enabled);
    void        resetStats();
    RenderStats stats();
    void        synchronize();

    // bvh-visualizer pass: pinhole camera, u = j/W, v = 1-(i+1)/H, tMax = FLT_MAX.
    void tracePrimaryStats(const Camera& camera, uint32_t width, uint32_t height, uint32_t* nodesVisitedOut,
                           uint8_t* hitOut, float* tOut, uint32_t* triangleTestsOut);
    // Batch closest-hit / any-hit of caller rays (6 floats each) -- the GPU twin of the
    // reference's CPU query rayIntersectBvh (src/common/ray_intersection.hpp:43-49).
    void intersectRays(const float* rays6, uint64_t numRays, float tMax, uint32_t* triangleOut, float* tOut, float* uvOut,
                       float* pOut, uint32_t* nodesVisitedOut, uint32_t* triangleTestsOut);
    void occludedRays(const float* rays6, uint64_t numRays, float tMax, float* visibilityOut);

private:
    struct Impl;
    std::unique_ptr<Impl> mImpl;
};

// Host-only check of the render path's BVH layouts (rf_wide.hpp) for a flattened tree: builds the 64-byte records and their
// compact-capable / 32-byte variants and verifies that every variant decodes to the same child planes and child words as the
// plain record, and that the carried "own" planes are the union of the children's.  Returns bit 0: boxes regular (wide layout
// usable), bit 1: compact-capable records usable, bit 2: 32-byte records usable; throws std::runtime_error on a mismatch.
uint32_t checkWideLayouts(std::span<const BvhNode> nodes, float* quadHalfAreaRatio = nullptr);
} // namespace rf
