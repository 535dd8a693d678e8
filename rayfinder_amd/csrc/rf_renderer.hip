// rf_renderer.hip -- wavefront path tracer for MI355X (gfx950): kernels + host driver.
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
#include "rf_renderer.hpp"

#include "rf_bvh.hpp"
#include "rf_camera.hpp"
#include "rf_data.hpp"
#include "rf_device.hpp"
#include "rf_wide.hpp"

#include <hip/hip_runtime.h>

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
#define RF_HIP(expr)                                                                                          \
    do                                                                                                        \
    {                                                                                                         \
        const hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                                 \
            throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " in " #expr);      \
    } while (0)

struct __attribute__((packed, aligned(4))) P3
{
    float x, y, z;
};

// Path state.  What one launch writes for the NEXT launch to stream through lives at QUEUE positions: entry q of a
// bounce's ray queue has its origin, direction, throughput, hit record and pending NEE term at index q of these arrays, so
// that every launch reads and writes them densely, in queue order (coalesced), however few of the batch's paths are still
// alive (by bounce 3 half of the slots are dead: slot-indexed, each surviving path cost a 64-byte line per stream).
// Direction and throughput are double-buffered: kShade reads entry q of the bounce's arrays while other workgroups already
// write entries of the next queue.  What belongs to the PATH for its whole life stays at its slot: the radiance sum
// (read by the accumulation in sample order) and the blue-noise pair.
struct PathStreams
{
    P3*     rayO;    // [queue position] origin.xyz of the ray to trace (kRaygen / kShade: the offset hit point)
    P3*     rayD;    // [queue position] direction.xyz, this bounce's
    P3*     thr;     // [queue position] throughput.rgb, this bounce's
    float4* rad;     // [slot] radiance.rgb
    float4* hit;     // [queue position] {triangle bits, u, v, t}
    P3*     pending; // [queue position] (throughput * solar radiance) * reflectance, waiting for visibility
    P3*     noise;   // [queue position] {u.x, cos(2 pi u.y), sin(2 pi u.y)}: the path's one blue-noise pair, this bounce's copy
    P3*     rayDOut; // [position in the NEXT queue] written by kShade
    P3*     thrOut;  // [position in the NEXT queue]
    P3*     noiseOut; // [position in the NEXT queue]: kShade copies the triple along; the shadow launch of the bounce reads it here
};

// 12-byte load of the xyz part of a float4 stream element (global_load_dwordx3): the L1 -> VGPR return path
// bounds the traversal kernels, so the unused .w lanes are not fetched
typedef float v3f __attribute__((ext_vector_type(3)));
__device__ __forceinline__ Vec3 load3(const float4* p)
{
    const v3f v = *reinterpret_cast<const v3f*>(p);
    return vec3(v.x, v.y, v.z);
}
// The queue-position arrays hold PACKED xyz triples (12-byte stride, global_load/store_dwordx3 at 4-byte alignment): they are
// streamed densely by every launch, so a quarter of their bytes would be padding otherwise.
__device__ __forceinline__ Vec3 load3(const P3* p)
{
    const P3 v = *p;
    return vec3(v.x, v.y, v.z);
}
__device__ __forceinline__ void store3(P3* p, Vec3 v)
{
    P3 o;
    o.x = v.x, o.y = v.y, o.z = v.z;
    *p = o;
}
// The same with the non-temporal hint, for kShade's streams: 100 B per hit written once and read once by the next launches -- tens of GB per bounce that would
// otherwise push what IS reused (shading records, texels, BVH records, the occluder grid) out of L2 / Infinity Cache: kShade -4.4 % (profiles/r04_occluder/
// nt_shade2.log).  The hint on kRaygen's stores, kShadowFirstLook's loads and the traversal kernels' hit records as well measured nothing more; on the traversal
// kernels' own path-state accesses it measured 1 % slower (round 2) -- those stay plain.
typedef float v3fu __attribute__((ext_vector_type(3), aligned(4)));
__device__ __forceinline__ void store3nt(P3* p, Vec3 v)
{
    v3fu o;
    o.x = v.x, o.y = v.y, o.z = v.z;
    __builtin_nontemporal_store(o, reinterpret_cast<v3fu*>(p));
}
__device__ __forceinline__ Vec3 load3nt(const P3* p)
{
    const v3fu v = __builtin_nontemporal_load(reinterpret_cast<const v3fu*>(p));
    return vec3(v.x, v.y, v.z);
}

// Path-state accesses of the traversal kernels (queue entry, origin, direction, result: touched once per ray).  A build with
// the non-temporal hint on them measured 1 % slower (shadow kernel 32.2 -> 33.3 ms per 32 spp), so they are plain.
__device__ __forceinline__ Vec3     load3s(const float4* p) { return load3(p); }
__device__ __forceinline__ Vec3     load3s(const P3* p) { return load3(p); }
__device__ __forceinline__ uint32_t loadQ(const uint32_t* p) { return *p; }
__device__ __forceinline__ void     store4s(float4* p, float x, float y, float z, float w) { *p = make_float4(x, y, z, w); }

struct DeviceCounters
{
    unsigned long long primaryRays, closestRays, shadowRays;
    unsigned long long closestNodeVisits, closestTriangleTests, shadowNodeVisits, shadowTriangleTests;
    unsigned int       stackHigh;
    unsigned int       pad;
    unsigned long long closestRecordFetches, shadowRecordFetches; // 64-byte wide records actually fetched (counting build)
    // wave-level trip counts of kTraceWide's loops (counting build): lane utilisation = lane work / (64 * trips)
    unsigned long long descendTrips[2], leafTrips[2], leafPhases[2], refillTrips[2], popLaneTrips[2], outerTrips[2];
    unsigned long long scalarRedo[2]; // rays redone by the scalar traversal (irregular or stack overflow), all builds
    unsigned long long abandonedRays; // rays whose traversal stack outgrew 96 entries (result = what was found until then), all builds
    unsigned long long occluderTried, occluderHit, occludedRays; // RF_EXP_PHASE builds: the any-hit launches' occluder cache
};

struct FrameParams
{
    uint32_t width, height;
    Camera   camera;
    uint32_t samplesPerPixel, numBounces;
    uint32_t firstFrame; // frameCount of sample 0 of this batch
    uint32_t numSamples; // samples traced in this batch
    uint32_t numTiles;
    uint32_t pixelsPadded; // numTiles * 1024
    // Path slot <-> (sample k of the batch, local pixel lp).  Groups of 2^g consecutive local pixels (g = 0: one pixel,
    // 6: one 8x8 block, 10: one tile) keep all their samples together:
    //     slot = (((lp >> g) * numSamples + k) << g) + (lp & (2^g - 1)),
    // so that neighbours in the ray queues (= in a wave, on a CU) are the same few pixels' other samples rather than the
    // same sample's other pixels.  kSlotSampleMajor: the round-1 order, slot = k * pixelsPadded + lp.
    uint32_t slotGroupShift;
    // Order of a pixel group's samples inside its run of slots: position p holds sample samplePerm[p] (inverse:
    // sampleInvPerm).  nullptr = identity.  kSamplePermutation sorts the batch's samples along a Z-order curve through their
    // R2 points, so that a wave's rays (a few neighbouring pixels x consecutive positions) leave the same surface in similar
    // directions (u = fract(blueNoise(pixel) + r2(sample)): the same pair drives every bounce, wgsl:52-55,194,209).
    const uint32_t* samplePerm;
    const uint32_t* sampleInvPerm;
    uint32_t tilesX;
};

constexpr uint32_t kSlotSampleMajor = 31u;

__device__ __forceinline__ void slotToSamplePixel(const FrameParams& fp, uint32_t slot, uint32_t& k, uint32_t& lp)
{
    if (fp.slotGroupShift == kSlotSampleMajor)
    {
        k = slot / fp.pixelsPadded;
        lp = slot % fp.pixelsPadded;
    }
    else
    {
        const uint32_t g = fp.slotGroupShift, chunk = slot >> g;
        k = chunk % fp.numSamples;
        lp = ((chunk / fp.numSamples) << g) + (slot & ((1u << g) - 1u));
    }
}
__device__ __forceinline__ size_t samplePixelToSlot(const FrameParams& fp, uint32_t k, uint32_t lp)
{
    if (fp.slotGroupShift == kSlotSampleMajor) return static_cast<size_t>(k) * fp.pixelsPadded + lp;
    const uint32_t g = fp.slotGroupShift;
    return ((static_cast<size_t>(lp >> g) * fp.numSamples + k) << g) + (lp & ((1u << g) - 1u));
}

// local pixel index (tile-major, 8x8 pixel blocks = one wave) -> image coordinates
__device__ __forceinline__ bool localPixelToXY(const FrameParams& fp, const uint32_t* tileIds, uint32_t lp, uint32_t& x, uint32_t& y)
{
    const uint32_t tile = tileIds[lp >> 10];
    const uint32_t w = lp & 1023u;
    const uint32_t block = w >> 6, lane = w & 63u;
    x = (tile % fp.tilesX) * kTileSize + (block & 3u) * 8u + (lane & 7u);
    y = (tile / fp.tilesX) * kTileSize + (block >> 2) * 8u + (lane >> 3);
    return x < fp.width && y < fp.height;
}

// Block-wide append of up to ITEMS candidates per thread with ONE atomic per block.  A single
// device-scope counter saturates near 90 atomics/us (MI355X_MICROARCH.md "dequeue"), so a
// one-atomic-per-wave append made the shade and raygen launches atomic-bound (130 k waves per
// launch = 1.5 ms); per block of 1024 entries it is 8 k atomics.  Must be reached by every thread of
// the block.  Output order: wave-major, then item, then lane (stays local to the block's entries).
constexpr int kItems = 4; // queue entries per thread in the per-entry kernels

template<int ITEMS>
__device__ __forceinline__ void blockAppend(const bool (&keep)[ITEMS], const uint32_t (&slot)[ITEMS], uint32_t* queue, uint32_t* count, uint32_t* sScratch,
                                            uint32_t (*position)[ITEMS] = nullptr)
{
    const uint32_t lane = __lane_id(), wave = threadIdx.x >> 6;
    uint32_t       offs[ITEMS];
    uint32_t       waveTotal = 0;
#pragma unroll
    for (int k = 0; k < ITEMS; ++k)
    {
        const unsigned long long mask = __ballot(keep[k]);
        offs[k] = waveTotal + __popcll(mask & ((1ull << lane) - 1ull));
        waveTotal += __popcll(mask);
    }
    if (lane == 0) sScratch[wave] = waveTotal;
    __syncthreads();
    if (threadIdx.x == 0)
    {
        const uint32_t total = sScratch[0] + sScratch[1] + sScratch[2] + sScratch[3];
        sScratch[4] = total ? atomicAdd(count, total) : 0u;
    }
    __syncthreads();
    uint32_t base = sScratch[4];
    for (uint32_t w = 0; w < wave; ++w) base += sScratch[w];
#pragma unroll
    for (int k = 0; k < ITEMS; ++k)
    {
        if (keep[k]) queue[base + offs[k]] = slot[k];
        if (position) (*position)[k] = base + offs[k]; // where the entry went (meaningful where keep[k])
    }
    __syncthreads(); // sScratch may be reused by the next append
}

__device__ __forceinline__ unsigned long long waveSum(unsigned long long v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}
__device__ __forceinline__ uint32_t waveMax(uint32_t v)
{
    for (int off = 32; off > 0; off >>= 1) v = max(v, static_cast<uint32_t>(__shfl_down(v, off)));
    return v;
}

// Inclusive scan over the 64 lanes of a wave, DPP only (no LDS): row_shr 1 / 2 / 4 / 8 inside the rows of 16, then row_bcast 15 / 31 across them.
// Must run with all 64 lanes enabled.  MAX: running maximum (of unsigned values; identity 0), else running sum.
template<bool MAX>
__device__ __forceinline__ uint32_t waveScanInclusive(uint32_t x)
{
    const auto step = [&](auto ctrl, auto rowMask) {
        const uint32_t y = static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), decltype(ctrl)::value, decltype(rowMask)::value, 0xF, true));
        x = MAX ? max(x, y) : x + y;
    };
    step(std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xF>{}); // row_shr:1
    step(std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xF>{}); // row_shr:2
    step(std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xF>{}); // row_shr:4
    step(std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xF>{}); // row_shr:8
    step(std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xA>{}); // row_bcast:15 into rows 1 and 3
    step(std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xC>{}); // row_bcast:31 into rows 2 and 3
    return x;
}
__device__ __forceinline__ uint32_t laneGather(uint32_t value, uint32_t srcLane) { return static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(static_cast<int>(srcLane << 2), static_cast<int>(value))); }
__device__ __forceinline__ float    laneGather(float value, uint32_t srcLane) { return __uint_as_float(laneGather(__float_as_uint(value), srcLane)); }

// Z-order key of sample k's R2 point (the temporal part of animatedBlueNoise, wgsl:606-615; only the ORDER matters)
__device__ __forceinline__ uint32_t sampleKey(uint32_t firstFrame, uint32_t spp, uint32_t k)
{
    const uint32_t n = (firstFrame + k) % spp;
    const float    rx = wFract(0.7548776662466927f * static_cast<float>(n)), ry = wFract(0.5698402909980532f * static_cast<float>(n));
    uint32_t       x = static_cast<uint32_t>(rx * 65536.0f) & 0xFFFFu, y = static_cast<uint32_t>(ry * 65536.0f) & 0xFFFFu;
    const auto     spread = [](uint32_t v) {
        v = (v | (v << 8)) & 0x00FF00FFu;
        v = (v | (v << 4)) & 0x0F0F0F0Fu;
        v = (v | (v << 2)) & 0x33333333u;
        v = (v | (v << 1)) & 0x55555555u;
        return v;
    };
    return spread(x) | (spread(y) << 1);
}

// perm / inverse of the batch's samples by (key, k): S is at most a few thousand, one thread per sample counts its rank
__global__ void kSamplePermutation(uint32_t firstFrame, uint32_t spp, uint32_t numSamples, uint32_t* perm, uint32_t* inv)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= numSamples) return;
    const uint32_t mine = sampleKey(firstFrame, spp, k);
    uint32_t       rank = 0;
    for (uint32_t j = 0; j < numSamples; ++j)
    {
        const uint32_t other = sampleKey(firstFrame, spp, j);
        rank += (other < mine || (other == mine && j < k)) ? 1u : 0u;
    }
    perm[rank] = k;
    inv[k] = rank;
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void kRaygen(FrameParams fp, DeviceScene scene, const uint32_t* tileIds, PathStreams ps,
                                                   uint32_t* queue, uint32_t* queueCount, DeviceCounters* counters)
{
    __shared__ uint32_t sScratch[8];
    const uint32_t      total = fp.numSamples * fp.pixelsPadded;
    bool                keep[kItems];
    uint32_t            slots[kItems], pos[kItems], px[kItems], py[kItems], sample[kItems];
    // pass 1: which slots are pixels of the image -> their positions in the first queue
#pragma unroll
    for (int k = 0; k < kItems; ++k)
    {
        const uint32_t slot = (blockIdx.x * kItems + k) * kBlock + threadIdx.x;
        bool           valid = slot < total;
        uint32_t       x = 0, y = 0, sampleIdx = 0, lp = 0;
        if (valid) slotToSamplePixel(fp, slot, sampleIdx, lp);
        if (valid) valid = localPixelToXY(fp, tileIds, lp, x, y);
        keep[k] = valid;
        slots[k] = slot;
        px[k] = x, py[k] = y, sample[k] = sampleIdx;
    }
    blockAppend<kItems>(keep, slots, queue, queueCount, sScratch, &pos);
    // pass 2: the rays, written at their queue positions
#pragma unroll
    for (int k = 0; k < kItems; ++k)
    {
        if (!keep[k]) continue;
        const uint32_t x = px[k], y = py[k];
        const uint32_t frame = fp.firstFrame + (fp.samplePerm ? fp.samplePerm[sample[k]] : sample[k]);
        float          nx, ny;
        animatedBlueNoise(scene.blueNoise, x, y, frame, fp.samplesPerPixel, nx, ny);

        // fragment centre (wgsl:36-43); v runs down the image
        const float u = (static_cast<float>(x) + 0.5f) / static_cast<float>(fp.width);
        const float v = (static_cast<float>(y) + 0.5f) / static_cast<float>(fp.height);
        const float s = u + nx / static_cast<float>(fp.width);
        const float t = (1.0f - v) + ny / static_cast<float>(fp.height);

        const float phi = 2.0f * kPi * ny;
        const float cosPhi = wCos(phi), sinPhi = wSin(phi);
        const float r = rf_sqrt(nx);
        const float lensX = fp.camera.lensRadius * (r * cosPhi);
        const float lensY = fp.camera.lensRadius * (r * sinPhi);
        const Vec3  origin = fp.camera.origin + (lensX * fp.camera.right + lensY * fp.camera.up);
        const Vec3  dir = normalize(fp.camera.lowerLeftCorner + s * fp.camera.horizontal + t * fp.camera.vertical - origin);

        // throughput = 1 and radiance = 0 (wgsl:183-184) are not stored: bounce 1 knows them (kFlagFirstBounce, kSky's
        // first-bounce flag), which saves 32 of the 80 bytes a path costs here and the reads back
        store3(ps.rayO + pos[k], origin);
        store3(ps.rayD + pos[k], dir);
        store3(ps.noise + pos[k], vec3(nx, cosPhi, sinPhi));
    }
    // primary rays are counted on the host (samples x valid pixels of the shard): one atomic per wave on a single counter
    // was what bound this kernel -- 261 k waves at ~90 same-address atomics/us = 2.9 of its 3.2 ms (MI355X_MICROARCH.md "dequeue")
    (void)counters;
}

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

// p = p0 + u*e1 + v*e2 offset along normalize(e1 x e2) (wgsl:511-519,523-544)
__device__ __forceinline__ Vec3 hitPoint(const DeviceScene& scene, uint32_t tri, float u, float v)
{
    const Vec3 p0 = load3(scene.triangles + kTriStride * tri), p1 = load3(scene.triangles + kTriStride * tri + 1),
               p2 = load3(scene.triangles + kTriStride * tri + 2);
    const Vec3   e1 = p1 - p0, e2 = p2 - p0;
    const Vec3   p = p0 + u * e1 + v * e2;
    return offsetRay(p, normalize(cross(e1, e2)));
}

// Sun direction sample for this path (wgsl:194,287-292,568-579): cone about sunDirection.
// The orthonormal basis about the sun direction (wgsl:309-319 applied to sunDirection) is the same for
// every sample of a frame: computed once on the host with the same f32 expressions and passed as kernel
// arguments (SGPRs) instead of ~15 VALU instructions per sample.
struct SunBasis
{
    Vec3 u, v;
};

__device__ __forceinline__ Vec3 sunSample(const SkyStateGpu& sky, const SunBasis& basis, float nx, float cosPhi, float sinPhi)
{
    const float cosThetaMax = __uint_as_float(kSolarCosThetaMaxBits);
    const float cosTheta = 1.0f - nx * (1.0f - cosThetaMax);
    const float sinTheta = rf_sqrt(1.0f - cosTheta * cosTheta);
    const Vec3  local = vec3(cosPhi * sinTheta, sinPhi * sinTheta, cosTheta);
    const Vec3  sun = vec3(sky.sunDirection[0], sky.sunDirection[1], sky.sunDirection[2]);
    return basisTimes(basis.u, basis.v, sun, local);
}

constexpr uint32_t kShadeLastBounce = 1u, kShadeFirstBounce = 2u;


// SORTED (option shade_sort_from_bounce): a tile's surviving paths are appended to the next queue in the order of the triangles they hit
// (counting sort over kSortBins ranges of triangle ids in LDS; triangles are in BVH leaf order, so that is an order by region of the
// scene) instead of input order: the 64 rays a wave of the next launches picks up then start close to each other.  The tile still
// occupies ONE contiguous run of the queue, so queue order stays slot order at the scale of 1024 entries (what is indexed by slot --
// the blue-noise triple, the radiance sum -- is touched by the same workgroups as before).  `sortScale`: bin of triangle t =
// (t * sortScale) >> 32.
constexpr uint32_t kSortBins = 256;
#if defined(RF_EXP_SHADE_WAVES)
#define RF_SHADE_BOUNDS __launch_bounds__(kBlock, RF_EXP_SHADE_WAVES)
#else
#define RF_SHADE_BOUNDS __launch_bounds__(kBlock) // (SORTED: 137 registers, three waves per SIMD; forced into 128 for four it is 2.5 % slower)
#endif
template<bool SORTED>
__global__ RF_SHADE_BOUNDS void kShade(DeviceScene scene, SkyStateGpu sky, SunBasis sunBasis, PathStreams ps, const uint32_t* queue,
                                                  const uint32_t* queueCount, uint32_t* hitQueue, uint32_t* hitCount, uint32_t* missQueue,
                                                  uint32_t* missCount, uint32_t bounceFlags, uint32_t sortScale)
{
    static_assert(kSortBins == kBlock, "one bin per thread");
    __shared__ uint32_t sScratch[8];
    __shared__ uint32_t sHist[SORTED ? kSortBins : 1], sStart[SORTED ? kSortBins : 1], sPerm[SORTED ? kItems * kBlock : 1];
    __shared__ float    sLut[256];
    constexpr uint32_t  kTile = kItems * kBlock;
    __shared__ float    sIn[SORTED ? 10 * kTile : 1]; // SORTED: throughput, blue-noise triple, {triangle, u, v} and slot of the tile's hits, [component][entry of the tile]
    const uint32_t      count = *queueCount;
    // grid-stride over tiles of kItems * kBlock queue entries: the grid is capped (kShadeMaxBlocks), so late bounces, whose
    // queues hold a sixth of the paths, do not pay for hundreds of thousands of empty workgroups
    const uint32_t tiles = (count + kItems * kBlock - 1) / (kItems * kBlock);
    if (blockIdx.x >= tiles) return; // whole block out of range (uniform)
    static_assert(kBlock == 256, "one table entry per thread");
    sLut[threadIdx.x] = scene.albedoLut[threadIdx.x];
    __syncthreads();
    const bool isLastBounce = (bounceFlags & kShadeLastBounce) != 0u, isFirstBounce = (bounceFlags & kShadeFirstBounce) != 0u;
  for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x)
  {
    // Pass 1: which entries hit, which left the scene -> both output queues are appended FIRST, so that every surviving path
    // knows its position in the next queue before it is shaded: what only the next two launches read (the NEE term) is
    // written there, densely, instead of at the path's slot (whose neighbours are mostly dead by bounce 3).
    bool       isHit[kItems], isMiss[kItems];
    uint32_t   slots[kItems], missEntries[kItems], outPos[kItems], hitTri[kItems];
#pragma unroll
    for (int k = 0; k < kItems; ++k)
    {
        const uint32_t i = (tile * kItems + k) * kBlock + threadIdx.x;
        isHit[k] = isMiss[k] = false;
        slots[k] = missEntries[k] = 0;
        hitTri[k] = kMiss;
        if (i >= count) continue;
        slots[k] = queue[i];
        missEntries[k] = i; // the miss list holds QUEUE positions: kSky finds the ray's direction and throughput there
        const Vec3     hitRec = SORTED ? load3(ps.hit + i) : vec3(ps.hit[i].x, 0.0f, 0.0f); // hit records sit at QUEUE positions (dense)
        const uint32_t tri = __float_as_uint(hitRec.x);
        hitTri[k] = tri;
        isMiss[k] = tri == kMiss; // the path ends in the sky: evaluated densely by this bounce's kSky launch
        isHit[k] = tri != kMiss;
        if constexpr (SORTED)
        {
            if (isHit[k])
            {
                // what pass 2 needs of this entry, read here in INPUT order (coalesced) and handed over in LDS: pass 2 works in the
                // tile's sorted order, where the 64 lanes of a wave would gather from ~57 different lines per stream
                const uint32_t l = static_cast<uint32_t>(k) * kBlock + threadIdx.x;
                const Vec3     t = isFirstBounce ? vec3(1.0f, 1.0f, 1.0f) : load3nt(ps.thr + i), z = load3nt(ps.noise + i);
                sIn[l] = t.x, sIn[kTile + l] = t.y, sIn[2 * kTile + l] = t.z;
                sIn[3 * kTile + l] = z.x, sIn[4 * kTile + l] = z.y, sIn[5 * kTile + l] = z.z;
                sIn[6 * kTile + l] = hitRec.x, sIn[7 * kTile + l] = hitRec.y, sIn[8 * kTile + l] = hitRec.z;
                sIn[9 * kTile + l] = __uint_as_float(slots[k]);
            }
        }
    }
    uint32_t sortedHits = 0, sortedBase = 0; // SORTED: hits of the tile, and where its run starts in the next queue
    if constexpr (SORTED)
    {
        // counting sort of the tile's hits by triangle range: rank inside the bin from an LDS counter, bin starts from a block scan
        uint32_t bin[kItems], rank[kItems];
        sHist[threadIdx.x] = 0u;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kItems; ++k)
        {
            bin[k] = rank[k] = 0u;
            if (!isHit[k]) continue;
            bin[k] = min(__umulhi(hitTri[k], sortScale), kSortBins - 1u);
            rank[k] = atomicAdd(&sHist[bin[k]], 1u);
        }
        __syncthreads();
        {
            const uint32_t n = sHist[threadIdx.x], lane = __lane_id(), wave = threadIdx.x >> 6;
            uint32_t       incl = n;
            for (int off = 1; off < 64; off <<= 1)
            {
                const uint32_t up = __shfl_up(incl, off);
                if (static_cast<int>(lane) >= off) incl += up;
            }
            if (lane == 63) sScratch[wave] = incl;
            __syncthreads();
            uint32_t before = 0;
            for (uint32_t w = 0; w < wave; ++w) before += sScratch[w];
            sStart[threadIdx.x] = before + incl - n;
            if (threadIdx.x == 0)
            {
                const uint32_t total = sScratch[0] + sScratch[1] + sScratch[2] + sScratch[3];
                sScratch[5] = total;
                sScratch[4] = total ? atomicAdd(hitCount, total) : 0u; // the tile's run in the next queue
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kItems; ++k)
            if (isHit[k]) sPerm[sStart[bin[k]] + rank[k]] = static_cast<uint32_t>(k) * kBlock + threadIdx.x;
        __syncthreads();
        // the thread's work from here on: entries k * 256 + tid of the SORTED order
        const uint32_t tileHits = sScratch[5], base = sScratch[4];
        sortedHits = tileHits, sortedBase = base;
#pragma unroll
        for (int k = 0; k < kItems; ++k)
        {
            const uint32_t p = static_cast<uint32_t>(k) * kBlock + threadIdx.x;
            isHit[k] = p < tileHits;
            outPos[k] = base + p;
            if (!isHit[k]) continue;
            const uint32_t local = sPerm[p];
            hitTri[k] = local; // (reused: which entry of the tile)
            slots[k] = __float_as_uint(sIn[9 * kTile + local]);
            hitQueue[outPos[k]] = slots[k];
        }
        __syncthreads(); // LDS is reused by the miss append and the next tile
    }
    else
        blockAppend<kItems>(isHit, slots, hitQueue, hitCount, sScratch, &outPos);
    blockAppend<kItems>(isMiss, missEntries, missQueue, missCount, sScratch);

    // Pass 2: shade the hits.  An entry is a chain of dependent gathers -- hit record -> shading record -> texture descriptor -> texel --
    // and four entries one after the other were four such chains end to end: the kernel waited.  Now the hit records of all
    // four entries are requested up front, and the shading record of entry k + 1 while entry k is shaded (its texel fetch included).
    // (SORTED only: bounce 1 -- coherent records, no sort -- streams at its memory rate as one entry at a time with fewer registers)
    constexpr bool kPipelined = SORTED;
    Vec3           hits[kPipelined ? kItems : 1]; // {triangle, u, v} of the hit records (t is not needed here)
    const auto     entryIndex = [&](int k) -> uint32_t {
        return SORTED ? (tile * kItems + hitTri[k] / kBlock) * kBlock + (hitTri[k] % kBlock) : (tile * kItems + static_cast<uint32_t>(k)) * kBlock + threadIdx.x;
    };
    if constexpr (kPipelined)
    {
#pragma unroll
        for (int k = 0; k < kItems; ++k) hits[k] = isHit[k] ? vec3(sIn[6 * kTile + hitTri[k]], sIn[7 * kTile + hitTri[k]], sIn[8 * kTile + hitTri[k]]) : Vec3{};
    }
    // everything this stage needs of the triangle sits in ONE 128-byte record (positions + packed attributes): one L2 line
    // per shaded hit instead of a triangle line and an attribute line (kShade 56.1 -> 53.0 ms per 128 spp)
    struct ShadeRecord
    {
        Vec3   p0, p1, p2;
        float4 a0, a1, a2, a3; // packed vertex attributes (one 64-byte sector): {n0.xyz n1.x} {n1.yz n2.xy} {n2.z uv0.xy uv1.x} {uv1.y uv2.xy textureIdx}
    };
    const auto fetchRecord = [&](uint32_t tri) {
        ShadeRecord   r;
        const float4* rec = scene.shadeRecords + 8 * static_cast<size_t>(tri);
#if defined(RF_EXP_SHADE_ABLATE) && RF_EXP_SHADE_ABLATE >= 3
        if constexpr (SORTED) rec = scene.shadeRecords + 8 * static_cast<size_t>(tri & 63u); // ablation (timing only): 64 records, all L1 hits
#endif
        r.p0 = load3(rec), r.p1 = load3(rec + 1), r.p2 = load3(rec + 2);
        r.a0 = rec[3], r.a1 = rec[4], r.a2 = rec[5], r.a3 = rec[6];
        return r;
    };
    const auto shade = [&](int k, float hu, float hv, const ShadeRecord& cur, uint32_t out) {
        const uint32_t i = entryIndex(k);
        (void)i;
        {
            // hit point pushed off the surface along the geometric normal (wgsl:511-519,523-544): origin of
            // the shadow ray and of the next bounce; same arithmetic as the scalar traversal (rf_device.hpp)
            const Vec3 p0 = cur.p0, p1 = cur.p1, p2 = cur.p2;
            const Vec3 e1 = p1 - p0, e2 = p2 - p0;
            const Vec3 hp = offsetRay(p0 + hu * e1 + hv * e2, normalize(cross(e1, e2)));
            store3nt(ps.rayO + out, hp); // (this bounce's origins have been consumed by the closest-hit launch)
        }
        // SORTED: this thread's entry is the tile's `local`-th in input order; its throughput and blue-noise triple were read in input
        // order (coalesced) by pass 1 and wait in LDS -- gathered from memory, the 64 lanes of a wave would touch ~57 different lines of
        // the tile's 12 KB per stream
        Vec3 throughput, nz;
        if constexpr (SORTED)
        {
            const uint32_t local = hitTri[k];
            throughput = vec3(sIn[local], sIn[kTile + local], sIn[2 * kTile + local]);
            nz = vec3(sIn[3 * kTile + local], sIn[4 * kTile + local], sIn[5 * kTile + local]);
        }
        else
        {
            throughput = isFirstBounce ? vec3(1.0f, 1.0f, 1.0f) : load3nt(ps.thr + i); // wgsl:184
            nz = load3nt(ps.noise + i);
        }
        const float nx = nz.x, cosPhi = nz.y, sinPhi = nz.z;
        store3nt(ps.noiseOut + out, nz); // travels with the path: dense for this bounce's shadow launch and for the next kShade
        const float4  a0 = cur.a0, a1 = cur.a1, a2 = cur.a2, a3 = cur.a3;
        const Vec3    n0 = vec3(a0.x, a0.y, a0.z), n1 = vec3(a0.w, a1.x, a1.y), n2 = vec3(a1.z, a1.w, a2.x);
        const float   b0 = 1.0f - hu - hv, b1 = hu, b2 = hv; // wgsl:515
        const Vec3    n = (b0 * n0 + b1 * n1) + b2 * n2;         // not normalised, wgsl:396
        const float   uvx = (b0 * a2.y + b1 * a2.w) + b2 * a3.y;
        const float   uvy = (b0 * a2.z + b1 * a3.x) + b2 * a3.z;
#if defined(RF_EXP_SHADE_ABLATE) && (RF_EXP_SHADE_ABLATE == 1 || RF_EXP_SHADE_ABLATE == 4)
        const Vec3    albedo = SORTED ? vec3(sLut[__float_as_uint(a3.w) & 255u], uvx - floorf(uvx), uvy - floorf(uvy)) : evalTexture(scene, sLut, __float_as_uint(a3.w), uvx, uvy); // ablation (timing only): no texel fetch
#else
        const Vec3    albedo = evalTexture(scene, sLut, __float_as_uint(a3.w), uvx, uvy);
#endif

        // next-event estimation towards the sun, wgsl:194-203 (cosine is not clamped)
        const Vec3 lightDirection = sunSample(sky, sunBasis, nx, cosPhi, sinPhi);
        const Vec3 lightIntensity = vec3(sky.solarRadiances[0], sky.solarRadiances[1], sky.solarRadiances[2]);
        const Vec3 brdf = albedo * kFrac1Pi;
        const Vec3 reflectance = brdf * dot(n, lightDirection);
        const Vec3 pend = (throughput * lightIntensity) * reflectance;
        store3nt(ps.pending + out, pend); // read by the shadow launch at the same queue position

        if (!isLastBounce)
        {
            // cosine-weighted bounce about the interpolated normal, wgsl:209-211,294-301,582-592
            const float sinTheta = rf_sqrt(1.0f - nx);
            const Vec3  local = vec3(cosPhi * sinTheta, sinPhi * sinTheta, rf_sqrt(nx));
            Vec3        bu, bv;
            pixarOnb(n, bu, bv);
            const Vec3 wi = basisTimes(bu, bv, n, local); // not renormalised
            const Vec3 t2 = throughput * albedo;
            store3nt(ps.rayDOut + out, wi);
            store3nt(ps.thrOut + out, t2);
        }
    };
    if constexpr (kPipelined)
    {
        const uint32_t tileHits = sortedHits, base = sortedBase;
        ShadeRecord    cur{};
        if (threadIdx.x < tileHits) cur = fetchRecord(__float_as_uint(hits[0].x));
#pragma unroll
        for (int k = 0; k < kItems; ++k)
        {
            const uint32_t p = static_cast<uint32_t>(k) * kBlock + threadIdx.x, pNext = p + kBlock; // positions in the tile's sorted order
            ShadeRecord    next{};
            if (k + 1 < kItems && pNext < tileHits) next = fetchRecord(__float_as_uint(hits[k + 1 < kItems ? k + 1 : k].x));
            if (p < tileHits) shade(k, hits[k].y, hits[k].z, cur, base + p);
            cur = next;
        }
    }
    else
    {
#pragma unroll 1
        for (int k = 0; k < kItems; ++k)
        {
            if (!isHit[k]) continue;
            const Vec3 h = load3(ps.hit + entryIndex(k));
            shade(k, h.y, h.z, fetchRecord(__float_as_uint(h.x)), outPos[k]);
        }
    }
    if constexpr (SORTED) __syncthreads(); // (a block that takes another tile refills sIn)
  }
}

// Paths that left the scene at this bounce: radiance += throughput * sky (wgsl:212-228,247-275).  One dense launch per
// bounce over that bounce's miss list (queue positions) instead of a divergent f64 branch inside kShade; it runs right
// after kShade, while the bounce's direction / throughput arrays and its queue are still intact.  All NEE terms of the path
// have been added by then (the shadow launch of the previous bounce is complete).  Grid-stride: the list length is only
// known on the device, and a worst-case grid of empty workgroups per bounce would cost more than the work.
__global__ __launch_bounds__(kBlock) void kSky(SkyStateGpu sky, PathStreams ps, const uint32_t* queue, const uint32_t* missQueue, const uint32_t* missCount,
                                                uint32_t firstBounce)
{
    const uint32_t n = *missCount;
    const bool     first = firstBounce != 0u; // left the scene at bounce 1: throughput 1, radiance 0, neither in memory
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
    {
        const uint32_t q = missQueue[i];
        const uint32_t slot = queue[q];
        const Vec3     v = load3(ps.rayD + q);
        const Vec3     thr = first ? vec3(1.0f, 1.0f, 1.0f) : load3(ps.thr + q);
        const Vec3     rad = first ? vec3(0.0f, 0.0f, 0.0f) : load3(ps.rad + slot);
        const Vec3     s = vec3(sky.sunDirection[0], sky.sunDirection[1], sky.sunDirection[2]);
        const float    theta = wAcos(v.y);
        const float    gamma = wAcos(minf(maxf(dot(v, s), -1.0f), 1.0f));
        // cos(gamma) and |cos(theta)| do not depend on the channel: evaluated once instead of three times (same values)
        const float cosGamma = wCos(gamma), cosTheta = fabsf(wCos(theta));
        const Vec3  dome = vec3(skyRadiance(sky, cosTheta, gamma, cosGamma, 0), skyRadiance(sky, cosTheta, gamma, cosGamma, 1), skyRadiance(sky, cosTheta, gamma, cosGamma, 2));
        const Vec3  radiance = rad + thr * dome;
        ps.rad[slot] = make_float4(radiance.x, radiance.y, radiance.z, 0.0f);
    }
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

// ------------------------------------------------------------------------------------------------
// Scheduling constants of the persistent traversal kernel (tuned on the atrium, tools/gpu_ab.py).
// ------------------------------------------------------------------------------------------------
constexpr uint32_t kChunk = 128;   // queue entries claimed per atomic (small enough that the tail stays balanced)
constexpr uint32_t kShards = 16;   // work cursors per launch, one 64-byte line each: a single cursor
                                   // saturates near 90 claims/us (8 M rays / 64 per 1.2 ms = 100/us)
constexpr uint32_t kLineWords = 16;
constexpr uint32_t kRefillMin = 40; // refill once this many lanes are idle (r02 sweep: 32 -> 40 = +1 %)
constexpr uint32_t kLeafVote = 20; // leave the descent loop when fewer lanes than this are descending (16..24 measure the same)

// record layouts (kTraceWide's COMPACT parameter) + the two paths outside kTraceWide
constexpr int kLayoutBinary = 0, kLayoutCompact = 1, kLayoutHot = 2, kLayoutQuad = 3, kLayoutQuadHalf = 4, kLayoutQuadLocal = 5, kLayoutOct = 6, kLayoutScalar = 7, kLayoutPacket = 8;
constexpr uint32_t kFlagShadowDirFromStream = 1u; // any-hit: direction from ps.rayD instead of the sun sample
constexpr uint32_t kFlagUniformTri = 8u;          // the same for the triangles of a leaf phase
constexpr uint32_t kFlagUniformFetch = 4u;        // try the scalar-cache path for records that every descending lane shares
constexpr uint32_t kFlagFirstBounce = 2u;         // any-hit: radiance so far is 0 and not in memory yet (kRaygen does not store it)
constexpr uint32_t kFlagOccluderCache = 16u;      // any-hit: a new ray first visits the leaves that stopped the last rays from its cell of the scene (see kTraceWide)
constexpr uint32_t kFlagOccluderNoTry = 32u;      // ... the launch runs behind kShadowFirstLook: its rays have had their first look, it only records what stopped them
constexpr uint32_t kFlagNoRayCount = 64u;
constexpr uint32_t kFlagDenseLeafShift = 8u;       // bits 11..8: leaf phases in which a parked lane holds this many triangles or more run over dense (lane, triangle) pairs (0: never; see kTraceWide)         // the launch's rays are counted elsewhere (kShadowFirstLook counted the whole queue)
#if defined(RF_EXP_OCC_SLOTS)
constexpr int kOccSlots = RF_EXP_OCC_SLOTS;
#else
constexpr int kOccSlots = 4; // entries per cell of the occluder grid (1, 2 or 4: shadow launches of the atrium -19 / -29 / -34 %, profiles/r04_occluder)
#endif
static_assert(kOccSlots == 1 || kOccSlots == 2 || kOccSlots == 4, "one aligned load per cell");
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

// Lane state of kTraceWide lives in ONE register, the next thing to visit: a child word of
// rf_wide.hpp (bit 31 clear: interior record index; set: leaf descriptor) or one of two sentinels
// (no leaf word reaches them: that would take count field 7 with big-leaf index 0x0FFFFFFE).
constexpr uint32_t kNodeIdle = 0xFFFFFFFFu; // no ray
constexpr uint32_t kNodeDone = 0xFFFFFFFEu; // ray finished, result not yet written

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

template<bool ANY_HIT, bool COUNT, bool NEAREST_FIRST = false, int COMPACT = 0, bool DENSE_LEAVES = false>
__global__ __launch_bounds__(kBlock, (COUNT && !NEAREST_FIRST) ? 2 : kWideWaves) void kTraceWide(DeviceScene scene, WideScene wide, SkyStateGpu sky, SunBasis sunBasis, PathStreams ps,
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
    __shared__ uint2 sStack[kDepth * kBlock];
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
    uint32_t hrot = 0u; // ... and (1/d.x < 0) << 4 | (1/d.y < 0) << 12 | (1/d.z < 0) << 20: rotate amounts that bring a plane word's NEAR plane into its low half
    PackedRay pr{};        // origin and 1/direction in the pairings of the record (rf_wide.hpp)
    Vec3      rayDir{};    // for the triangle tests
    uint32_t  negMask = 0; // bit a: 1/direction[a] < 0 (reference child order); bit 3: class B ray (rf_wide.hpp)
    float     rayTMax = tMax;
    // Closest-hit launches keep the stack top as a BYTE offset into sStack (lane * 8 + depth * kBlock * 8): a push is one ds_write + one add, no
    // shift-or for the address, and -- in the quad steps -- one bound check per step instead of one per push: closest-hit launches -1.5 % (round 4,
    // gpurun_out A/B in profiles/r04_lanes).  The any-hit launches measured +2.5 % with it and keep the plain depth, as do the counting builds
    // (they report it).
    constexpr bool kPtrStack = !COUNT && !ANY_HIT;
    const int     spBase = kPtrStack ? static_cast<int>(threadIdx.x * sizeof(uint2)) : 0;
    constexpr int kSpStep = kPtrStack ? static_cast<int>(kBlock * sizeof(uint2)) : 1;
    constexpr int kSpLimit = kPtrStack ? kDepth * static_cast<int>(kBlock * sizeof(uint2)) : kDepth; // (depth == kDepth <=> offset >= this: lane * 8 < kBlock * 8)
    int       stackSize = spBase;
    const auto stackAt = [&](int s) -> uint2& {
        if constexpr (kPtrStack) return *reinterpret_cast<uint2*>(reinterpret_cast<char*>(sStack) + s);
        else return sStack[s * kBlock + threadIdx.x];
    };
    bool      needScalar = false; // irregular ray or stack overflow: redo with the scalar traversal
    // An any-hit ray's rayTMax never changes, so an entry that passed `tmin < rayTMax` when it was pushed passes it when it is popped: such a
    // kernel keeps only the words on its stack (no tmin to select, store and compare) -- except the reference-bookkeeping build, which
    // pushes missed children with tmin = +inf to count them.
    constexpr bool kStackWordsOnly = ANY_HIT && !kRefCount;
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
            if constexpr (kStackWordsOnly) spillBuf[spilled + i] = stackAt(slotS(i)).x;
            else spillBuf[spilled + i] = stackAt(slotS(i));
        }
        const int depth = kPtrStack ? (stackSize - spBase) / kSpStep : stackSize;
        for (int i = kEvict; i < depth; ++i)
        {
            if constexpr (kStackWordsOnly) stackAt(slotS(i - kEvict)).x = stackAt(slotS(i)).x;
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
            if constexpr (kStackWordsOnly) stackAt(slotS(i)).x = spillBuf[spilled + i];
            else stackAt(slotS(i)) = spillBuf[spilled + i];
        }
        stackSize = slotS(kEvict);
    };
    auto      push = [&](uint32_t word, float tmin) -> bool {
        if (stackSize >= kSpLimit && !evict()) return false;
        if constexpr (kStackWordsOnly) stackAt(stackSize).x = word;
        else stackAt(stackSize) = make_uint2(word, __float_as_uint(tmin));
        stackSize += kSpStep;
        return true;
    };
    auto      pushUnchecked = [&](uint32_t word, float tmin) {
        if constexpr (kStackWordsOnly) stackAt(stackSize).x = word;
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
                node = stackAt(stackSize).x;
                if (COUNT) ++wPop;
            }
            return;
        }
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
                    base = shardBegin + __shfl(base, 0);
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
                const Vec3 o = load3s(ps.rayO + resultIndex);
                Vec3       dir;
                if (ANY_HIT && !shadowDirFromStream)
                {
                    const Vec3 nz = load3s(ps.noiseOut + resultIndex);
                    dir = sunSample(sky, sunBasis, nz.x, nz.y, nz.z);
                }
                else dir = load3s(ps.rayD + resultIndex);
                const RayPrep ray = prepareRay(o, dir);
                pr = packRay(ray);
                rayDir = dir;
                const uint32_t rayClass = classifyRay(ray);
                negMask = ray.negX | (ray.negY << 1) | (ray.negZ << 2) | (rayClass == kRayHasInf ? 8u : 0u) | (triedCell ? 16u : 0u);
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
                    hbx = -(o.x * pr.iXY.x);
                    hby = -(o.y * pr.iXY.y);
                    hbz = -(o.z * pr.iZ);
                    hrot = (ray.negX << 4) | (ray.negY << 12) | (ray.negZ << 20);
                    lselX = ray.negX ? 0x00040005u : 0x00050004u, lselY = ray.negY ? 0x00040005u : 0x00050004u, lselZ = ray.negZ ? 0x00040005u : 0x00050004u;
                    const uint32_t signXY = ray.negX | (ray.negY << 1);
                    octKey = ray.negZ ? ((16u * (3u - signXY)) | (0x7777u << 8)) : 16u * signXY;
                }
                float      rootTMin;
                const bool rootOk = slabBounds(ray, wide.rootLo, wide.rootHi, rootTMin) && rootTMin < rayTMax;
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
        do
        {
            if (kPhase) ++wDescend;
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
                                    if constexpr (kStackWordsOnly) stackAt(stackSize + rank * kSpStep).x = words[e];
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
                            const uint4* n = wide.quadLocal + 4 * static_cast<size_t>(node);
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
                        const uint32_t rx = hrot, ry = hrot >> 8, rz = hrot >> 16;
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
                            const uint4* n = wide.quadHalf + 4 * static_cast<size_t>(node);
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
                    const bool     h0 = okq0 && (kSkipTMax || tq0 < rayTMax), h1 = okq1 && (kSkipTMax || tq1 < rayTMax) && w1 != kQuadEmpty,
                                   h2 = okq2 && (kSkipTMax || tq2 < rayTMax), h3 = okq3 && (kSkipTMax || tq3 < rayTMax) && w3 != kQuadEmpty;
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

        // ---- leaves
#if defined(RF_EXP_PHASE)
        if (__ballot(node - kWideLeafBit < kNodeDone - kWideLeafBit) != 0ull) ++phaseLeafWave;
#endif
        uint32_t occluderWord = 0u; // kOccluderCache: the leaf in which this lane has just found an occluder
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
                const float4* t0 = scene.triangles + kTriStride * static_cast<size_t>(first);
                firstA = t0[0], firstB = t0[1], firstC = t0[2];
                float4 hi;
                if constexpr (kOccluderCache) hi = t0[3]; // .w: what the occluder cache remembers for this leaf (leafBoxesIntoTriangles)
                else
                {
                    const v3f h3 = *reinterpret_cast<const v3f*>(t0 + 3);
                    hi = make_float4(h3.x, h3.y, h3.z, 0.0f);
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
__global__ __launch_bounds__(kBlock) void kShadowFirstLook(DeviceScene scene, WideScene wide, SkyStateGpu sky, SunBasis sunBasis, PathStreams ps, const uint32_t* queue,
                                                            const uint32_t* queueCount, uint32_t* list, uint32_t* listCount, DeviceCounters* counters, float tMax, uint32_t firstBounce)
{
    __shared__ uint32_t sScratch[8];
    const uint32_t      count = *queueCount;
    const uint32_t      tiles = (count + kItems * kBlock - 1) / (kItems * kBlock);
    if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&counters->shadowRays, static_cast<unsigned long long>(count));
    // (one entry after the other: staging the kItems entries of a thread -- four cells, then four triangle records in flight per lane -- takes 163
    // registers, three waves per SIMD instead of eight, and measured 17 % slower: profiles/r04_occluder/firstlook2.log)
    for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x)
    {
        bool     keep[kItems];
        uint32_t entry[kItems];
#pragma unroll
        for (int k = 0; k < kItems; ++k)
        {
            const uint32_t i = (tile * kItems + k) * kBlock + threadIdx.x;
            keep[k] = i < count;
            entry[k] = i;
            if (i >= count) continue;
            const Vec3     o = load3(ps.rayO + i);
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
                const float4   a = t0[0], b = t0[1], c = t0[2];
                const v3f      hi = *reinterpret_cast<const v3f*>(t0 + 3);
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

// Queue occupancy per bounce: Q[b-1] paths enter bounce b (closest-hit rays), Q[b] of them hit
// something (shadow rays).  Folded into running totals at the end of every batch.
// `listCounts` / `lookMask`: bounces whose any-hit launch ran behind kShadowFirstLook (bit b) -- Q[b] minus the length of its list is what that kernel answered.
__global__ void kBounceTotals(const uint32_t* queueCounts, uint32_t numBounces, unsigned long long* totals, const uint32_t* listCounts, unsigned long long lookMask, unsigned long long* lookBatch)
{
    const uint32_t b = threadIdx.x;
    if (b >= numBounces) return;
    const uint32_t k = min(b, RenderStats::kMaxBounceStats - 1);
    atomicAdd(&totals[k], static_cast<unsigned long long>(queueCounts[kLineWords * b]));
    atomicAdd(&totals[RenderStats::kMaxBounceStats + k], static_cast<unsigned long long>(queueCounts[kLineWords * (b + 1)]));
    if ((lookMask >> b) & 1ull)
    {
        const unsigned long long rays = queueCounts[kLineWords * (b + 1)], answered = rays - listCounts[kLineWords * b];
        atomicAdd(&totals[2 * RenderStats::kMaxBounceStats + k], answered);
        atomicAdd(&lookBatch[0], answered); // this batch alone: the host decides from it whether the first look pays (Impl::firstLookHoldOff)
        atomicAdd(&lookBatch[1], rays);
    }
}

// image[lp] += radiance of samples 0..numSamples-1 in order (f32, wgsl:55); image is the compact
// tile-major float4 buffer.
__global__ __launch_bounds__(kBlock) void kAccumulate(FrameParams fp, const uint32_t* tileIds, PathStreams ps, float4* image)
{
    const uint32_t lp = blockIdx.x * kBlock + threadIdx.x;
    if (lp >= fp.pixelsPadded) return;
    uint32_t x, y;
    if (!localPixelToXY(fp, tileIds, lp, x, y)) return;
    float4 acc = image[lp];
    for (uint32_t k = 0; k < fp.numSamples; ++k)
    {
        const float4 r = ps.rad[samplePixelToSlot(fp, fp.sampleInvPerm ? fp.sampleInvPerm[k] : k, lp)];
        acc.x += r.x;
        acc.y += r.y;
        acc.z += r.z;
    }
    image[lp] = acc;
}

// The same sum for the pixel-major slot order (slotGroupShift = 0), where a pixel's samples sit in one contiguous run of
// numSamples float4: there kAccumulate's per-thread reads are a 16-byte gather at a stride of numSamples * 16 bytes (8.2 ms per
// 320 spp of a 1080p frame).  Here one wave takes kAccPixels pixels: their runs are read coalesced (1 KiB per load) into LDS, then
// one lane per (pixel, channel) adds its samples in sample-index order -- the order is the result (f32, H15), so the
// additions stay sequential; only the memory traffic changes.  Dynamic LDS: kAccPixels * (numSamples + 1) * 12 bytes (rows padded by one float: bank-conflict-free sums).
#if defined(RF_EXP_ACC_PIXELS)
constexpr uint32_t kAccPixels = RF_EXP_ACC_PIXELS;
#else
constexpr uint32_t kAccPixels = 4;
#endif
constexpr uint32_t kAccMaxSamples = 1024;

__global__ __launch_bounds__(64) void kAccumulateRuns(FrameParams fp, const uint32_t* tileIds, PathStreams ps, float4* image)
{
    extern __shared__ float sRun[]; // [pixel][channel][sample], rows of S + 1 floats: the twelve lanes that sum walk twelve different banks
    const uint32_t S = fp.numSamples, R = S + 1u, lane = threadIdx.x;
    const uint32_t lp0 = blockIdx.x * kAccPixels;
    for (uint32_t px = 0; px < kAccPixels; ++px)
    {
        const uint32_t lp = lp0 + px;
        if (lp >= fp.pixelsPadded) break;
        const float4* run = ps.rad + static_cast<size_t>(lp) * S;
        float*        dst = sRun + px * 3u * R;
        for (uint32_t p = lane; p < S; p += 64u)
        {
            // position p of the run holds sample samplePerm[p]: stored at ITS index, so that the sums below walk LDS in order
            const Vec3     v = load3(run + p);
            const uint32_t k = fp.samplePerm ? fp.samplePerm[p] : p;
            dst[k] = v.x;
            dst[R + k] = v.y;
            dst[2u * R + k] = v.z;
        }
    }
    __syncthreads();
    if (lane >= kAccPixels * 3u) return;
    const uint32_t px = lane / 3u, c = lane % 3u, lp = lp0 + px;
    if (lp >= fp.pixelsPadded) return;
    uint32_t x, y;
    if (!localPixelToXY(fp, tileIds, lp, x, y)) return;
    float*       out = reinterpret_cast<float*>(image + lp) + c;
    float        acc = *out;
    const float* src = sRun + (px * 3u + c) * R;
#pragma unroll 8
    for (uint32_t k = 0; k < S; ++k) acc += src[k]; // sample order (wgsl:56-57): one dependent chain of f32 additions per channel
    *out = acc;
}

// wgsl:59-63,277-285 -> BGRA8Unorm texel
__global__ void kTonemap(const float4* image, uint32_t n, uint32_t accumulatedSamples, float exposure, uint32_t* out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 px = image[i];
    const float  in[3] = {px.x, px.y, px.z};
    uint32_t     q[3];
    for (int c = 0; c < 3; ++c)
    {
        const float est = in[c] / static_cast<float>(accumulatedSamples);
        const float x = exposure * est;
        const float a = 2.51f, b = 0.03f, cc = 2.43f, d = 0.59f, e = 0.14f;
        float       y = (x * (a * x + b)) / (x * (cc * x + d) + e);
        y = minf(maxf(y, 0.0f), 1.0f);
        const float srgb = wPow(y, 1.0f / 2.2f);
        q[c] = static_cast<uint32_t>(floorf(srgb * 255.0f + 0.5f));
    }
    out[i] = q[2] | (q[1] << 8) | (q[0] << 16) | (255u << 24);
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


// ------------------------------------------------------------------------------------------------
// Deferred-lighting variant: src/pt/deferred_renderer_lighting_pass.wgsl:96-186 and
// deferred_renderer_resolve_pass.wgsl:33-54 over a G-buffer that comes from ONE PRIMARY RAY per pixel instead of
// the reference's raster pass (deferred_renderer_gbuffer_pass.wgsl: needs a hardware rasteriser).  What differs from
// the reference by construction, and only there: the albedo is the nearest texel (the path tracer's textureLookup,
// the raster pass samples through a sampler), the shading normal and position are not quantised by a texture format,
// and visibility comes from the primary ray rather than the depth buffer.  Everything downstream is the WGSL's: the
// fixed 2-bounce surfaceColor with the solar disk in the sky term (:231-235), the OTHER self-intersection constants
// (1/16384 and 1024, :498-500), one blue-noise pair per pixel with a 2^20-frame cycle, the 0.1 / 0.9 exponential
// resolve.  An interactive-preview path: one thread per pixel, the scalar reference-ordered traversal.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ Vec3 offsetPositionDeferred(Vec3 p, Vec3 n)
{
    constexpr float kOrigin = 1.0f / 32.0f, kFloatScale = 1.0f / 16384.0f, kIntScale = 1024.0f; // lighting_pass.wgsl:498-500
    const int       ox = static_cast<int>(kIntScale * n.x), oy = static_cast<int>(kIntScale * n.y), oz = static_cast<int>(kIntScale * n.z);
    const Vec3      shifted = vec3(__int_as_float(__float_as_int(p.x) + (p.x < 0 ? -ox : ox)), __int_as_float(__float_as_int(p.y) + (p.y < 0 ? -oy : oy)),
                                   __int_as_float(__float_as_int(p.z) + (p.z < 0 ? -oz : oz)));
    return vec3(fabsf(p.x) < kOrigin ? p.x + kFloatScale * n.x : shifted.x, fabsf(p.y) < kOrigin ? p.y + kFloatScale * n.y : shifted.y,
                fabsf(p.z) < kOrigin ? p.z + kFloatScale * n.z : shifted.z);
}

struct DeferredSurface
{
    Vec3 plain, offset, normal, albedo;
};

// interpolated attributes of a hit (lighting_pass.wgsl:312-321) + hit point pushed along the geometric normal (:447-450)
__device__ __forceinline__ DeferredSurface deferredSurface(const DeviceScene& scene, const float* lut, const ClosestHit& h)
{
    DeferredSurface out;
    const Vec3 p0 = load3(scene.triangles + kTriStride * h.triangle), p1 = load3(scene.triangles + kTriStride * h.triangle + 1),
               p2 = load3(scene.triangles + kTriStride * h.triangle + 2);
    const Vec3 e1 = p1 - p0, e2 = p2 - p0;
    out.plain = p0 + h.u * e1 + h.v * e2;
    out.offset = offsetPositionDeferred(out.plain, normalize(cross(e1, e2)));
    const float4* va = scene.attributes + 4 * static_cast<size_t>(h.triangle);
    const float4  a0 = va[0], a1 = va[1], a2 = va[2], a3 = va[3];
    const Vec3    n0 = vec3(a0.x, a0.y, a0.z), n1 = vec3(a0.w, a1.x, a1.y), n2 = vec3(a1.z, a1.w, a2.x);
    const float   b0 = 1.0f - h.u - h.v, b1 = h.u, b2 = h.v;
    out.normal = (b0 * n0 + b1 * n1) + b2 * n2;
    const float uvx = (b0 * a2.y + b1 * a2.w) + b2 * a3.y, uvy = (b0 * a2.z + b1 * a3.x) + b2 * a3.z;
    out.albedo = evalTexture(scene, lut, __float_as_uint(a3.w), uvx, uvy);
    return out;
}

// lighting_pass.wgsl:200-238: the dome plus the solar disk; TERRESTRIAL_SOLAR_RADIUS = 0.255f * (PI / 180f) in f32
__device__ __forceinline__ Vec3 skyWithSun(const SkyStateGpu& sky, Vec3 v)
{
    const Vec3  s = vec3(sky.sunDirection[0], sky.sunDirection[1], sky.sunDirection[2]);
    const float theta = wAcos(v.y), gamma = wAcos(minf(maxf(dot(v, s), -1.0f), 1.0f));
    const float cosGamma = wCos(gamma), cosTheta = fabsf(wCos(theta));
    const bool  inDisk = gamma / __uint_as_float(0x3B91D640u) <= 1.0f;
    return vec3(skyRadiance(sky, cosTheta, gamma, cosGamma, 0) + (inDisk ? sky.solarRadiances[0] : 0.0f),
                skyRadiance(sky, cosTheta, gamma, cosGamma, 1) + (inDisk ? sky.solarRadiances[1] : 0.0f),
                skyRadiance(sky, cosTheta, gamma, cosGamma, 2) + (inDisk ? sky.solarRadiances[2] : 0.0f));
}

__global__ __launch_bounds__(kBlock) void kDeferredLighting(DeviceScene scene, SkyStateGpu sky, SunBasis sunBasis, Camera cam, uint32_t width, uint32_t height,
                                                             uint32_t frameCount, float jitterX, float jitterY, float exposure, float* sampleBuffer,
                                                             float* accumulationBuffer, uint32_t* bgraOut, DeviceCounters* counters)
{
    __shared__ uint32_t sStack[kLdsStack * kBlock];
    __shared__ float    sLut[256];
    sLut[threadIdx.x] = scene.albedoLut[threadIdx.x];
    __syncthreads();
    // 8x8-pixel blocks per wave
    const uint32_t blocksX = (width + 7u) / 8u;
    const uint32_t wave = (blockIdx.x * kBlock + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    const uint32_t x = (wave % blocksX) * 8u + (lane & 7u), y = (wave / blocksX) * 8u + (lane >> 3);
    // lanes outside the frame (sizes that are not multiples of 8) stay alive for the wave reduction at the end and contribute 0
    unsigned long long closest = 0, shadow = 0;
    if (x < width && y < height)
    {
    const float W = static_cast<float>(width), H = static_cast<float>(height);
    // pixel centre displaced by the frame's projection jitter (deferred_renderer.cpp:309-315: (r2 - 0.5) / size in NDC)
    const float su = (static_cast<float>(x) + 0.5f) / W - (jitterX - 0.5f) / (2.0f * W);
    const float tv = (1.0f - (static_cast<float>(y) + 0.5f) / H) - (jitterY - 0.5f) / (2.0f * H);
    const Vec3  rd = normalize(cam.lowerLeftCorner + cam.horizontal * su + cam.vertical * tv - cam.origin);
    TraversalCounters  tc;
    ClosestHit         h;
    closest = 1;
    Vec3               color;
    const Vec3         lightIntensity = vec3(sky.solarRadiances[0], sky.solarRadiances[1], sky.solarRadiances[2]);
    if (!traverse<false, false>(scene, cam.origin, rd, kTMax, &sStack[threadIdx.x], h, tc)) color = skyWithSun(sky, rd); // :106-117
    else
    {
        DeferredSurface sf = deferredSurface(scene, sLut, h);
        Vec3            normal = sf.normal, albedo = sf.albedo;
        Vec3            position = offsetPositionDeferred(sf.plain, normal); // :118-125: along the SHADING normal
        float           ux, uy;
        animatedBlueNoise(scene.blueNoise, x, y, frameCount, 1u << 20, ux, uy);
        const float phi = 2.0f * kPi * uy;
        const float cosPhi = wCos(phi), sinPhi = wSin(phi);
        const Vec3  light = sunSample(sky, sunBasis, ux, cosPhi, sinPhi);
        const auto  lightSample = [&](Vec3 pos, Vec3 n, Vec3 alb) { // :188-198
            const Vec3 reflectance = (alb * kFrac1Pi) * dot(n, light);
            ClosestHit unused;
            ++shadow;
            const float vis = traverse<true, false>(scene, pos, light, kTMax, &sStack[threadIdx.x], unused, tc) ? 0.0f : 1.0f;
            return ((lightIntensity * reflectance) * vis) * __uint_as_float(kSolarInvPdfBits);
        };
        Vec3 radiance = vec3(0.0f, 0.0f, 0.0f), throughput = vec3(1.0f, 1.0f, 1.0f);
        radiance = radiance + throughput * lightSample(position, normal, albedo);
        for (int bounce = 1; bounce < 2; ++bounce) // NUM_BOUNCES = 2 (:140)
        {
            const float sinTheta = rf_sqrt(1.0f - ux);
            Vec3        bu, bv;
            pixarOnb(normal, bu, bv);
            const Vec3 wi = basisTimes(bu, bv, normal, vec3(cosPhi * sinTheta, sinPhi * sinTheta, rf_sqrt(ux)));
            throughput = throughput * albedo;
            ++closest;
            if (traverse<false, false>(scene, position, wi, kTMax, &sStack[threadIdx.x], h, tc))
            {
                sf = deferredSurface(scene, sLut, h);
                position = sf.offset;
                normal = sf.normal;
                albedo = sf.albedo;
            }
            else
            {
                radiance = radiance + throughput * skyWithSun(sky, wi);
                break;
            }
            radiance = radiance + throughput * lightSample(position, normal, albedo);
        }
        color = radiance;
    }
    const size_t idx = static_cast<size_t>(y) * width + x;
    sampleBuffer[3 * idx] = color.x;
    sampleBuffer[3 * idx + 1] = color.y;
    sampleBuffer[3 * idx + 2] = color.z;
    // resolve_pass.wgsl:38-52
    Vec3 outc = color;
    if (frameCount != 0u)
    {
        const Vec3 prev = vec3(accumulationBuffer[3 * idx], accumulationBuffer[3 * idx + 1], accumulationBuffer[3 * idx + 2]);
        outc = 0.1f * color + 0.9f * prev;
    }
    accumulationBuffer[3 * idx] = outc.x;
    accumulationBuffer[3 * idx + 1] = outc.y;
    accumulationBuffer[3 * idx + 2] = outc.z;
    const float in[3] = {outc.x, outc.y, outc.z};
    uint32_t    q[3];
    for (int c = 0; c < 3; ++c)
    {
        const float xx = exposure * in[c];
        const float a = 2.51f, b = 0.03f, cc = 2.43f, d = 0.59f, e = 0.14f;
        float       yy = (xx * (a * xx + b)) / (xx * (cc * xx + d) + e);
        yy = minf(maxf(yy, 0.0f), 1.0f);
        q[c] = static_cast<uint32_t>(floorf(wPow(yy, 1.0f / 2.2f) * 255.0f + 0.5f));
    }
    bgraOut[idx] = q[2] | (q[1] << 8) | (q[0] << 16) | (255u << 24);
    if (tc.abandoned) atomicAdd(&counters->abandonedRays, 1ull);
    }
    const unsigned long long cr = waveSum(closest), sr = waveSum(shadow);
    if (__lane_id() == 0)
    {
        atomicAdd(&counters->closestRays, cr);
        atomicAdd(&counters->shadowRays, sr);
    }
}

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
    DeviceBuffer<float4>    ownedImage;
    float4*                 image = nullptr; // compact tile-major accumulation buffer
    uint64_t                imageBytes = 0;
    bool                    imageDirty = true; // needs zeroing before the next sample

    uint64_t                validPixels = 0;     // pixels of this rank's tiles that lie inside the frame
    unsigned long long      primaryRaysHost = 0; // samples traced x validPixels since the last resetStats()
    uint64_t                maxPaths = 0;
    DeviceBuffer<P3>        sRayO, sRayD, sRayD2, sThr, sThr2, sPending, sNoise, sNoise2; // queue-position arrays, packed xyz; rayD / thr / noise: double-buffered (PathStreams)
    DeviceBuffer<float4>    sRad, sHit;
    DeviceBuffer<uint32_t>  queueA, queueB, missQueue, queueCounts;
    DeviceBuffer<DeviceCounters> counters;
    DeviceBuffer<unsigned long long> bounceTotals; // 3 x kMaxBounceStats: closest-hit rays, shadow rays, shadow rays answered by kShadowFirstLook

    // deferred-lighting variant: its own frame counter and buffers (array<array<f32, 3>>)
    uint32_t               deferredFrameCount = 0;
    DeviceBuffer<float>    deferredSample, deferredAccum;
    DeviceBuffer<uint32_t> deferredBgra;

    bool counting = false, timing = false;
    int      traversalVariant = 2; // 0 = one ray per thread over 32-B nodes (A/B baseline), 2 = persistent waves over 64-B wide nodes
    uint32_t wideBlocks = 0;
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
    // leaf phases in which a parked lane holds this many triangles or more run over dense (lane, triangle) pairs (kTraceWide; 0: never), from this bounce on
    uint32_t optDenseLeafMin = 5, optDenseLeafFromBounce = 1; // (5: the plain atrium's leaves of up to 4 triangles keep the loop -- 3 measured +1 % there; the clutter scene gains the same with 2 .. 5)
    uint32_t maxLeafTriangles = 0; // of the scene (upload): scenes without a leaf as long as the threshold run the instantiations WITHOUT the dense block
    uint32_t optShadeSortFromBounce = 2, sortScale = 0;         // kShade of bounce >= this appends its tile's hits in triangle order (0: never)
    uint32_t optChunkEarly = 256, optChunkEarlyBounces = 2;      // queue entries per cursor claim at bounces 1-2
    uint32_t optRefillMinDeep = 22, optRefillMinDeepQuad = 40, optRefillDeepFromBounce = 3; // closest-hit launches of bounce >= 3 refill at another count: 22 idle lanes on the 64-byte and the
                                                                                             // half-precision quad records (VALU bound: idle lanes cost most), 40 on the exact quad records (L1 bound: a refill is a wave-wide stall)
    // kShade grid cap (0: one workgroup per tile of 1024 entries, the default: workgroups then append to the hit queue in
    // roughly queue order, which keeps neighbouring pixels' rays together -- a capped, grid-striding kShade saved its empty
    // workgroups but cost the traversal kernels 2-5 %)
    uint32_t optShadeBlocks = 0;
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

    // bytes of path state + queues per path slot (eight packed xyz streams, two float4 streams, three u32 queues)
    static constexpr uint64_t kBytesPerPath = 8 * sizeof(P3) + 2 * sizeof(float4) + 3 * sizeof(uint32_t);

    void releasePathState()
    {
        sRayO.release(), sRayD.release(), sRayD2.release(), sThr.release(), sThr2.release(), sRad.release(), sHit.release();
        sPending.release(), sNoise.release(), sNoise2.release(), queueA.release(), queueB.release(), missQueue.release();
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
                        tryAlloc(queueA, paths) && tryAlloc(queueB, paths) && tryAlloc(missQueue, paths);
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
            for (const uint32_t t : tiles)
            {
                const uint32_t x0 = (t % tilesX) * kTileSize, y0 = (t / tilesX) * kTileSize;
                validPixels += static_cast<uint64_t>(std::min(kTileSize, params.width - x0)) * std::min(kTileSize, params.height - y0);
            }
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
        float           tMax;
        dim3            grid;
        uint32_t        extraLds;
    };
    template<bool ANY_HIT, bool COUNT, bool NEAREST, int COMPACT, bool DENSE = false>
    void launchWide(const WideScene& w, const WideArgs& a, uint32_t flags)
    {
        hipLaunchKernelGGL((kTraceWide<ANY_HIT, COUNT, NEAREST, COMPACT, DENSE>), a.grid, dim3(kBlock), a.extraLds, stream, scene, w, sky, sunBasis, a.ps, a.queue, a.count, a.cursor, counters.ptr,
                           a.refillMin, optLeafVote, a.chunk, a.tMax, flags);
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
    // contain the block -- the layouts the renderer picks by itself -- are used only then)
    bool denseWanted(uint32_t flags) const
    {
        const uint32_t threshold = (flags >> kFlagDenseLeafShift) & 15u;
        return threshold != 0u && maxLeafTriangles >= threshold;
    }
    void launchClosestWide(int layout, bool count, const WideScene& w, const WideArgs& a, uint32_t flags)
    {
        if (count) return launchWide<false, true, false, 0>(w, a, flags);
        if (denseWanted(flags))
            switch (layout)
            {
            case kLayoutQuadLocal: return launchWide<false, false, false, 5, true>(w, a, flags);
            case kLayoutQuadHalf: return launchWide<false, false, false, 4, true>(w, a, flags);
            case kLayoutQuad: return launchWide<false, false, false, 3, true>(w, a, flags);
            default: break;
            }
        switch (layout)
        {
        case kLayoutOct: return launchWide<false, false, false, 6>(w, a, flags);
        case kLayoutQuadLocal: return launchWide<false, false, false, 5>(w, a, flags);
        case kLayoutQuadHalf: return launchWide<false, false, false, 4>(w, a, flags);
        case kLayoutQuad: return launchWide<false, false, false, 3>(w, a, flags);
#if defined(RF_EXP_LEGACY_LAYOUTS)
        case kLayoutHot: return launchWide<false, false, false, 2>(w, a, flags);
        case kLayoutCompact: return launchWide<false, false, false, 1>(w, a, flags);
#endif
        default: return launchWide<false, false, false, 0>(w, a, flags);
        }
    }
    // nearest: entries ordered by slab distance (NEAREST_FIRST); else record order (the conservative layouts' default: optShadowSignOrder)
    void launchShadowWide(int layout, bool nearest, bool count, const WideScene& w, const WideArgs& a, uint32_t flags)
    {
        if (count) return nearest ? launchWide<true, true, true, 0>(w, a, flags) : launchWide<true, true, false, 0>(w, a, flags);
        if (denseWanted(flags))
        {
            if (layout == kLayoutQuadLocal && !nearest) return launchWide<true, false, false, 5, true>(w, a, flags);
            if (layout == kLayoutQuadHalf && !nearest) return launchWide<true, false, false, 4, true>(w, a, flags);
            if (layout == kLayoutQuad && nearest) return launchWide<true, false, true, 3, true>(w, a, flags);
        }
        switch (layout)
        {
        case kLayoutQuadLocal: return nearest ? launchWide<true, false, true, 5>(w, a, flags) : launchWide<true, false, false, 5>(w, a, flags);
        case kLayoutQuadHalf: return nearest ? launchWide<true, false, true, 4>(w, a, flags) : launchWide<true, false, false, 4>(w, a, flags);
        case kLayoutQuad: return nearest ? launchWide<true, false, true, 3>(w, a, flags) : launchWide<true, false, false, 3>(w, a, flags);
#if defined(RF_EXP_LEGACY_LAYOUTS)
        case kLayoutHot: return launchWide<true, false, true, 2>(w, a, flags);
        case kLayoutCompact: return launchWide<true, false, true, 1>(w, a, flags);
#endif
        default: return nearest ? launchWide<true, false, true, 0>(w, a, flags) : launchWide<true, false, false, 0>(w, a, flags);
        }
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
        const WideArgs wa{ps, queueA.ptr, queueCounts.ptr, queueCounts.ptr + kLineWords, optRefillMin, optChunk, tMax, grid, 0u};
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
            launchClosestWide(layoutIfPresent(optQueryCompact), false, wide, wa, denseFlag);
            hipLaunchKernelGGL(kHitPoints, dim3((count + 255) / 256), dim3(256), 0, stream, scene, sHit.ptr, sRayO.ptr, count);
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
        const uint32_t words = kLine * (2 * numBounces + 1) + kLine * kShards * 2 * numBounces + kLine * numBounces;
        if (queueCounts.count < words) queueCounts.alloc(words);
        uint32_t* const missCounts = queueCounts.ptr + kLine * (numBounces + 1);
        uint32_t* const cursors = queueCounts.ptr + kLine * (2 * numBounces + 1);
        uint32_t* const listCounts = cursors + kLine * kShards * 2 * numBounces; // lengths of the lists kShadowFirstLook leaves to the any-hit launches, per bounce
        unsigned long long lookMask = 0ull;
        const uint32_t  itemBlocks = static_cast<uint32_t>((paths + kBlock * kItems - 1) / (kBlock * kItems));
        RF_HIP(hipMemsetAsync(queueCounts.ptr, 0, queueCounts.count * sizeof(uint32_t), stream));

        uint32_t* qIn = queueA.ptr;
        uint32_t* qOut = queueB.ptr;
        if (fp.samplePerm)
            hipLaunchKernelGGL(kSamplePermutation, dim3((numSamples + 255) / 256), dim3(256), 0, stream, firstFrame, fp.samplesPerPixel, numSamples,
                               const_cast<uint32_t*>(fp.samplePerm), const_cast<uint32_t*>(fp.sampleInvPerm));
        launchTimed(0, [&] {
            hipLaunchKernelGGL(kRaygen, dim3(itemBlocks), dim3(kBlock), 0, stream, fp, scene, tileIds.ptr, ps, qIn, queueCounts.ptr, counters.ptr);
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
            const uint32_t refillClosest = bounce >= optRefillDeepFromBounce ? (layoutClosest == kLayoutQuad ? optRefillMinDeepQuad : optRefillMinDeep) : optRefillMin;
            (void)conservativeClosest;
            launchTimed(1, [&] {
                if (layoutClosest == kLayoutScalar)
                {
                    if (counting)
                        hipLaunchKernelGGL(kTraceClosest<true>, dim3(blocks), dim3(kBlock), 0, stream, scene, ps, qIn, countIn, counters.ptr);
                    else
                        hipLaunchKernelGGL(kTraceClosest<false>, dim3(blocks), dim3(kBlock), 0, stream, scene, ps, qIn, countIn, counters.ptr);
                }
#if defined(RF_EXP_LEGACY_LAYOUTS)
                else if (layoutClosest == kLayoutPacket)
                    hipLaunchKernelGGL((kTracePacket<false>), persistentGrid, dim3(kBlock), 0, stream, scene, wide, sky, sunBasis, ps, qIn, countIn, counters.ptr, kTMax, 0u);
#endif
                else
                    launchClosestWide(layoutClosest, counting, wide, WideArgs{ps, qIn, countIn, cursorClosest, counting ? optRefillMin : refillClosest, chunkNow, kTMax, persistentGrid, counting ? 0u : optExtraLds},
                                      uniformFlag);
            }, bounce - 1);
            uint32_t* const missCount = missCounts + kLine * (bounce - 1);
            launchTimed(2, [&] {
                const uint32_t shadeFlags = (bounce == numBounces ? kShadeLastBounce : 0u) | (bounce == 1 ? kShadeFirstBounce : 0u);
                const dim3     shadeGrid(optShadeBlocks ? std::min(itemBlocks, optShadeBlocks) : itemBlocks);
                if (optShadeSortFromBounce != 0u && bounce >= optShadeSortFromBounce)
                    hipLaunchKernelGGL(kShade<true>, shadeGrid, dim3(kBlock), 0, stream, scene, sky, sunBasis, ps, qIn, countIn, qOut, countOut, missQueue.ptr, missCount, shadeFlags, sortScale);
                else
                    hipLaunchKernelGGL(kShade<false>, shadeGrid, dim3(kBlock), 0, stream, scene, sky, sunBasis, ps, qIn, countIn, qOut, countOut, missQueue.ptr, missCount, shadeFlags, 0u);
                // the paths that left the scene at this bounce, while its direction / throughput arrays and queue are intact
                hipLaunchKernelGGL(kSky, dim3(std::min(blocks, skyBlocks)), dim3(kBlock), 0, stream, sky, ps, qIn, missQueue.ptr, missCount, bounce == 1 ? 1u : 0u);
            });
            // occluder cache (kTraceWide, kFlagOccluderCache): the conservative-record any-hit launches of bounces 1..optOccluderCacheBounces; their rays are
            // short (a third of the steps), so the deep launches refill earlier
            const int      layoutShadow = shadowLayoutFor(bounce);
            const bool     cachedShadow = layoutShadow != kLayoutScalar && layoutShadow != kLayoutPacket && cachedShadowFor(bounce);
            // ... behind kShadowFirstLook (see there) from the second batch on: the first batch of a renderer fills the grid (the traversal kernel's own first look serves)
            const bool      firstLook = cachedShadow && occluderHintLevels == 0u && optShadowFirstLookFromBounce != 0u && bounce >= optShadowFirstLookFromBounce && bounce <= 64u && occluderGridWarm && firstLookHoldOff == 0u;
            uint32_t* const listCount = listCounts + kLine * (bounce - 1);
            uint32_t* const countShadow = firstLook ? listCount : countOut;
            if (firstLook) lookMask |= 1ull << (bounce - 1);
            const uint32_t shadowFlags = (bounce == 1 ? kFlagFirstBounce : 0u) | uniformFlag | (cachedShadow ? kFlagOccluderCache : 0u) | (firstLook ? (kFlagOccluderNoTry | kFlagNoRayCount) : 0u);
            launchTimed(3, [&] {
                WideScene wide = this->wide; // (the launches below name `wide`)
                if (firstLook)
                {
                    // (the bounce's input queue is free by now -- kShade and kSky have consumed it -- and holds the list)
                    const dim3 lookGrid(optShadeBlocks ? std::min(itemBlocks, optShadeBlocks) : itemBlocks);
                    hipLaunchKernelGGL(kShadowFirstLook, lookGrid, dim3(kBlock), 0, stream, scene, wide, sky, sunBasis, ps, qOut, countOut, qIn, listCount, counters.ptr, kTMax, bounce == 1 ? 1u : 0u);
                    wide.rayList = qIn;
                }
                if (layoutShadow == kLayoutScalar)
                {
                    if (counting)
                        hipLaunchKernelGGL(kTraceShadow<true>, dim3(blocks), dim3(kBlock), 0, stream, scene, sky, sunBasis, ps, qOut, countOut, counters.ptr, bounce == 1 ? 1u : 0u);
                    else
                        hipLaunchKernelGGL(kTraceShadow<false>, dim3(blocks), dim3(kBlock), 0, stream, scene, sky, sunBasis, ps, qOut, countOut, counters.ptr, bounce == 1 ? 1u : 0u);
                }
#if defined(RF_EXP_LEGACY_LAYOUTS)
                else if (layoutShadow == kLayoutPacket)
                    hipLaunchKernelGGL((kTracePacket<true>), persistentGrid, dim3(kBlock), 0, stream, scene, wide, sky, sunBasis, ps, qOut, countOut, counters.ptr, kTMax,
                                       shadowFlags & kFlagFirstBounce);
#endif
                else
                {
                    // the conservative layouts (VALU bound) visit a record's entries in record order unless asked otherwise; the exact and binary records nearest-first
                    const bool conservative = layoutShadow == kLayoutQuadLocal || layoutShadow == kLayoutQuadHalf;
                    const bool nearest = shadowNearestFirst && !(conservative && optShadowSignOrder);
                    launchShadowWide(layoutShadow, nearest, counting, wide, WideArgs{ps, qOut, countShadow, cursorShadow, optRefillMin, chunkNow, kTMax, persistentGrid, counting ? 0u : optExtraLds}, shadowFlags);
                }
            }, bounce - 1);
            std::swap(qIn, qOut);
            std::swap(ps.rayD, ps.rayDOut);
            std::swap(ps.thr, ps.thrOut);
            std::swap(ps.noise, ps.noiseOut);
        }
        hipLaunchKernelGGL(kBounceTotals, dim3(1), dim3(64), 0, stream, queueCounts.ptr, std::min(numBounces, 64u), bounceTotals.ptr, listCounts, lookMask, lookBatch.ptr);
        if (lookMask != 0ull && lookBatchHost != nullptr)
        {
            RF_HIP(hipMemcpyAsync(lookBatchHost, lookBatch.ptr, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
            RF_HIP(hipEventRecord(lookEvent, stream));
            lookPending = true;
        }
        if (wide.occGrid != nullptr) occluderGridWarm = true;
        launchTimed(4, [&] {
            if (fp.slotGroupShift == 0u && numSamples > 4u && numSamples <= kAccMaxSamples && optAccumulateRuns)
                hipLaunchKernelGGL(kAccumulateRuns, dim3((fp.pixelsPadded + kAccPixels - 1) / kAccPixels), dim3(64), kAccPixels * 3u * (numSamples + 1u) * sizeof(float), stream, fp,
                                   tileIds.ptr, ps, image);
            else
                hipLaunchKernelGGL(kAccumulate, dim3((fp.pixelsPadded + kBlock - 1) / kBlock), dim3(kBlock), 0, stream, fp, tileIds.ptr, ps, image);
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
        const std::vector<unsigned long long> z(3 * RenderStats::kMaxBounceStats, 0ull);
        m.bounceTotals.upload(z.data(), z.size());
    }
    {
        hipDeviceProp_t prop{};
        RF_HIP(hipGetDeviceProperties(&prop, m.device));
        int perCu = 0;
        RF_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, kTraceWide<false, false>, kBlock, 0));
        m.wideBlocks = static_cast<uint32_t>(std::max(perCu, 1)) * static_cast<uint32_t>(prop.multiProcessorCount);
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
    if (n) hipLaunchKernelGGL(kTonemap, dim3((n + 255) / 256), dim3(256), 0, m.stream, m.image, n, m.accumulated, m.params.exposure, out.ptr);
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
    hipLaunchKernelGGL(kTonemap, dim3((n + 255) / 256), dim3(256), 0, m.stream, static_cast<const float4*>(imageDevice), n, samples, m.params.exposure, out.ptr);
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
        hipLaunchKernelGGL(kDeferredLighting, dim3((waves * 64 + kBlock - 1) / kBlock), dim3(kBlock), 0, m.stream, m.scene, m.sky, m.sunBasis, m.params.camera, W, H,
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
    else if (name == "refill_min") mImpl->optRefillMin = mImpl->optRefillMinDeep = mImpl->optRefillMinDeepQuad = static_cast<uint32_t>(value); // (all: a sweep of one value covers every launch)
    else if (name == "refill_min_deep") mImpl->optRefillMinDeep = mImpl->optRefillMinDeepQuad = static_cast<uint32_t>(value);
    else if (name == "refill_deep_from_bounce") mImpl->optRefillDeepFromBounce = static_cast<uint32_t>(std::max<int64_t>(value, 1));
    else if (name == "leaf_vote") mImpl->optLeafVote = static_cast<uint32_t>(value);
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
    else if (name == "persistent_blocks") mImpl->wideBlocks = static_cast<uint32_t>(value);
    else if (name == "extra_lds")
    {
        Impl& m = *mImpl;
        m.optExtraLds = static_cast<uint32_t>(std::max<int64_t>(value, 0));
        hipDeviceProp_t prop{};
        RF_HIP(hipGetDeviceProperties(&prop, m.device));
        int perCu = 0;
        RF_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&perCu, kTraceWide<false, false>, kBlock, m.optExtraLds));
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
    RF_HIP(hipMemset(m.bounceTotals.ptr, 0, 3 * RenderStats::kMaxBounceStats * sizeof(unsigned long long)));
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
    unsigned long long totals[3 * RenderStats::kMaxBounceStats];
    RF_HIP(hipMemcpy(totals, m.bounceTotals.ptr, sizeof totals, hipMemcpyDeviceToHost));
    for (uint32_t b = 0; b < RenderStats::kMaxBounceStats; ++b)
    {
        s.closestRaysByBounce[b] = totals[b];
        s.shadowRaysByBounce[b] = totals[RenderStats::kMaxBounceStats + b];
        s.shadowRaysHintAnswered += totals[2 * RenderStats::kMaxBounceStats + b];
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
    hipLaunchKernelGGL(kPrimaryStats, dim3((waves * 64 + kBlock - 1) / kBlock), dim3(kBlock), 0, m.stream, m.scene, camera, width, height,
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
    hipLaunchKernelGGL(kIntersectRays, dim3(static_cast<uint32_t>((numRays + kBlock - 1) / kBlock)), dim3(kBlock), 0, m.stream, m.scene, rays.ptr,
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
    hipLaunchKernelGGL(kOccludedRays, dim3(static_cast<uint32_t>((numRays + kBlock - 1) / kBlock)), dim3(kBlock), 0, m.stream, m.scene, rays.ptr,
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
