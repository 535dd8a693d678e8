// rf_kernels.hpp -- what the three translation units of the wavefront path tracer share (round 5: rf_renderer.hip, one 4 600-line unit until then,
// is now rf_trace.hip -- the traversal kernels --, rf_shade.hip -- ray generation, shading, sky, accumulation, display, the deferred variant -- and
// rf_renderer.hip -- the host driver).  Path-state streams, queue / slot arithmetic, wave helpers, the launch flags and scheduling constants of the
// traversal kernels, and the typed entry points through which the host unit launches kernels that live in the other two: each kernel unit hands out the
// ADDRESS of its kernels (xxxKernel() accessors), the host launches through the pointer -- same arguments, same type checking as a direct launch.
#pragma once

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
namespace kern
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
    // RF_EXP_PHASE builds (round 6): what the lanes that do NOT take a step in a descend trip are doing -- parked at a leaf (waiting for the leaf phase) or without a ray
    // (finished / never filled: waiting for the refill); and, in the leaf passes, lanes that still stand at an interior node / lanes without a ray.  Lane-trips, [closest, shadow]
    unsigned long long descendParked[2], descendIdle[2], leafInterior[2], leafIdle[2];
};

// Division of a 32-bit unsigned by a divisor that is a RUN-TIME value but the same for a whole launch (samples per batch, tiles per row, samples per pixel): the exact quotient from
// one multiply-high, an add and two shifts (Granlund & Montgomery's round-up method) instead of the ~30-instruction software division the compiler emits for `n / d` -- kRaygen is
// VALU bound and divided three times per path (round 6).  Host: FastDiv::make(d); device: div(n) == n / d for every n < 2^32, d >= 1.
struct FastDiv
{
    uint32_t mul, shift1, shift2, divisor;
    static FastDiv make(uint32_t d)
    {
        FastDiv f{0u, 0u, 0u, d ? d : 1u};
        uint32_t l = 0;
        while ((1ull << l) < f.divisor) ++l; // ceil(log2 d)
        f.mul = static_cast<uint32_t>(((1ull << 32) * ((1ull << l) - f.divisor)) / f.divisor + 1ull);
        f.shift1 = l < 1u ? l : 1u;
        f.shift2 = l > 1u ? l - 1u : 0u;
        return f;
    }
    __host__ __device__ __forceinline__ uint32_t div(uint32_t n) const
    {
#if defined(__HIP_DEVICE_COMPILE__)
        const uint32_t t = __umulhi(mul, n);
#else
        const uint32_t t = static_cast<uint32_t>((static_cast<uint64_t>(mul) * n) >> 32);
#endif
        return (t + ((n - t) >> shift1)) >> shift2;
    }
    __host__ __device__ __forceinline__ void divmod(uint32_t n, uint32_t& q, uint32_t& r) const
    {
        q = div(n);
        r = n - q * divisor;
    }
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
    uint32_t skipOrigins; // a pinhole camera whose primary launch takes the origin as a constant (kFlagConstOrigin): kRaygen leaves ps.rayO alone
    // Round 6: kRaygen's queue positions WITHOUT an atomic.  One device-scope counter serves ~88 returning atomics per microsecond (MI355X_MICROARCH.md, "dequeue"), and a 320-spp
    // batch of a 1080p frame is 648 k blocks of 1 024 slots: 7.4 ms of the kernel's 7.6 were that one counter.  Which slots are pixels of the image is known in advance: the tiles'
    // valid extents.  tileValidBefore[t] = valid pixels in the shard's tiles before tile t (numTiles + 1 entries, host: configureShard); the rank of a pixel inside its tile follows
    // from the tile's valid width / height in closed form (validRankInTile).  Slot orders with samples of a pixel GROUP kept together (slotGroupShift 1 .. 10) keep the atomic append.
    const uint32_t* tileValidBefore; // nullptr: append with the atomic
    uint32_t        validPixels;     // of the shard
    FastDiv         divNumSamples, divPixelsPadded, divTilesX, divSamplesPerPixel; // the launch-invariant divisors of the slot / pixel / frame arithmetic (host: traceBatch)
};

// rank of local pixel w (tile-major: 8x8 blocks of 64 lanes, localPixelToXY) among the VALID pixels of its tile -- the pixels with x < vw && y < vh in that order
__device__ __forceinline__ uint32_t validRankInTile(uint32_t w, uint32_t vw, uint32_t vh)
{
    const uint32_t block = w >> 6, lane = w & 63u, bx = block & 3u, by = block >> 2, lx = lane & 7u, ly = lane >> 3;
    const uint32_t ch = min(8u, vh - min(vh, 8u * by)), cw = min(8u, vw - min(vw, 8u * bx)); // valid rows / columns of this 8x8 block
    return vw * min(vh, 8u * by) + ch * min(vw, 8u * bx) + cw * ly + lx;                     // block rows above + blocks to the left + rows above inside the block + pixels to the left
}

constexpr uint32_t kSlotSampleMajor = 31u;

__device__ __forceinline__ void slotToSamplePixel(const FrameParams& fp, uint32_t slot, uint32_t& k, uint32_t& lp)
{
    if (fp.slotGroupShift == kSlotSampleMajor) fp.divPixelsPadded.divmod(slot, k, lp);
    else
    {
        const uint32_t g = fp.slotGroupShift, chunk = slot >> g;
        uint32_t       group;
        fp.divNumSamples.divmod(chunk, group, k);
        lp = (group << g) + (slot & ((1u << g) - 1u));
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
    uint32_t       tileX, tileY;
    fp.divTilesX.divmod(tile, tileY, tileX);
    x = tileX * kTileSize + (block & 3u) * 8u + (lane & 7u);
    y = tileY * kTileSize + (block >> 2) * 8u + (lane >> 3);
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
                                            uint32_t (*position)[ITEMS] = nullptr, const uint32_t (*second)[ITEMS] = nullptr, uint32_t* queue2 = nullptr)
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
        if (second != nullptr && keep[k]) queue2[base + offs[k]] = (*second)[k]; // a second list with the same positions
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
// kShadeSelfShadow: kShade tests every hit's shadow ray against the hit triangle ITSELF (its leaf's exact box, then the triangle: both sit in the shading record it has in
// registers) and writes the positions of the hits that this does not settle to `shadowList`; the bounce's any-hit launches work through that list only (see kShade)
constexpr uint32_t kShadeSelfShadow = 4u;
constexpr uint32_t kLookFirstBounce = 1u, kLookNoRayCount = 2u; // kShadowFirstLook's flags


constexpr uint32_t kSortBins = 256; // kShade<SORTED>: triangle ranges of its counting sort (the host derives sortScale from it)
#if defined(RF_EXP_ACC_PIXELS)
constexpr uint32_t kAccPixels = RF_EXP_ACC_PIXELS;
#else
constexpr uint32_t kAccPixels = 4;
#endif
constexpr uint32_t kAccMaxSamples = 1024;

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
constexpr uint32_t kFlagConstOrigin = 128u;      // closest-hit, bounce 1: every ray starts at WideScene::constOrigin (a pinhole camera: kRaygen does not write the origins, the refill does not read them)
constexpr uint32_t kFlagDenseLeafShift = 8u;       // bits 11..8: leaf phases in which a parked lane holds this many triangles or more run over dense (lane, triangle) pairs (0: never; see kTraceWide)         // the launch's rays are counted elsewhere (kShadowFirstLook counted the whole queue)
#if defined(RF_EXP_OCC_SLOTS)
constexpr int kOccSlots = RF_EXP_OCC_SLOTS;
#else
constexpr int kOccSlots = 4; // entries per cell of the occluder grid (1, 2 or 4: shadow launches of the atrium -19 / -29 / -34 %, profiles/r04_occluder)
#endif
static_assert(kOccSlots == 1 || kOccSlots == 2 || kOccSlots == 4, "one aligned load per cell");
// Lane state of kTraceWide lives in ONE register, the next thing to visit: a child word of
// rf_wide.hpp (bit 31 clear: interior record index; set: leaf descriptor) or one of two sentinels
// (no leaf word reaches them: that would take count field 7 with big-leaf index 0x0FFFFFFE).
constexpr uint32_t kNodeIdle = 0xFFFFFFFFu; // no ray
constexpr uint32_t kNodeDone = 0xFFFFFFFEu; // ray finished, result not yet written


// ---- kernel entry points (defined in rf_trace.hip / rf_shade.hip; nullptr: that instantiation is not compiled into this build)
using SamplePermutationKernel = void (*)(uint32_t firstFrame, uint32_t spp, uint32_t numSamples, uint32_t* perm, uint32_t* inv);
using RaygenKernel = void (*)(FrameParams fp, DeviceScene scene, const uint32_t* tileIds, PathStreams ps, uint32_t* queue, uint32_t* queueCount, DeviceCounters* counters);
using TraceClosestKernel = void (*)(DeviceScene scene, PathStreams ps, const uint32_t* queue, const uint32_t* queueCount, DeviceCounters* counters);
using ShadeKernel = void (*)(DeviceScene scene, SkyStateGpu sky, SunBasis sunBasis, PathStreams ps, const uint32_t* queue, const uint32_t* queueCount, uint32_t* hitQueue, uint32_t* hitCount, uint32_t* missQueue, uint32_t* missSlots, uint32_t* missCount, uint32_t* shadowList, uint32_t* shadowListCount, uint32_t bounceFlags, uint32_t sortScale);
using SkyKernel = void (*)(SkyStateGpu sky, PathStreams ps, const uint32_t* missSlots, const uint32_t* missQueue, const uint32_t* missCount, uint32_t firstBounce);
using TraceShadowKernel = void (*)(DeviceScene scene, SkyStateGpu sky, SunBasis sunBasis, PathStreams ps, const uint32_t* queue, const uint32_t* queueCount, DeviceCounters* counters, uint32_t firstBounce);
using TraceWideKernel = void (*)(DeviceScene scene, WideScene wide, SkyStateGpu sky, SunBasis sunBasis, PathStreams ps, const uint32_t* queue, const uint32_t* queueCount, uint32_t* cursor, DeviceCounters* counters, uint32_t refillMin, uint32_t leafVote, uint32_t chunkMax, float tMax, uint32_t flags);
using ShadowFirstLookKernel = void (*)(DeviceScene scene, WideScene wide, SkyStateGpu sky, SunBasis sunBasis, PathStreams ps, const uint32_t* queue, const uint32_t* inList, const uint32_t* inCount, uint32_t* list, uint32_t* listCount, DeviceCounters* counters, float tMax, uint32_t flags);
using TracePacketKernel = void (*)(DeviceScene scene, WideScene wide, SkyStateGpu sky, SunBasis sunBasis, PathStreams ps, const uint32_t* queue, const uint32_t* queueCount, DeviceCounters* counters, float tMax, uint32_t flags);
using HitPointsKernel = void (*)(DeviceScene scene, const float4* hit, P3* rayO, uint32_t n);
using BounceTotalsKernel = void (*)(const uint32_t* queueCounts, uint32_t numBounces, unsigned long long* totals, const uint32_t* listCounts, unsigned long long lookMask, unsigned long long* lookBatch, const uint32_t* shadowListCounts, unsigned long long selfMask, DeviceCounters* counters);
using AccumulateKernel = void (*)(FrameParams fp, const uint32_t* tileIds, PathStreams ps, float4* image);
using AccumulateRunsKernel = void (*)(FrameParams fp, const uint32_t* tileIds, PathStreams ps, float4* image);
using TonemapKernel = void (*)(const float4* image, uint32_t n, uint32_t accumulatedSamples, float exposure, uint32_t* out);
using PrimaryStatsKernel = void (*)(DeviceScene scene, Camera cam, uint32_t width, uint32_t height, uint32_t* nodesVisited, uint8_t* hitOut, float* tOut, uint32_t* triTests, DeviceCounters* counters);
using IntersectRaysKernel = void (*)(DeviceScene scene, const float* rays, uint64_t n, float tMax, uint32_t* triOut, float* tOut, float* uvOut, float* pOut, uint32_t* nvOut, uint32_t* ttOut, DeviceCounters* counters);
using OccludedRaysKernel = void (*)(DeviceScene scene, const float* rays, uint64_t n, float tMax, float* visOut, DeviceCounters* counters);
using DeferredLightingKernel = void (*)(DeviceScene scene, SkyStateGpu sky, SunBasis sunBasis, Camera cam, uint32_t width, uint32_t height, uint32_t frameCount, float jitterX, float jitterY, float exposure, float* sampleBuffer, float* accumulationBuffer, uint32_t* bgraOut, DeviceCounters* counters);

// rf_trace.hip
TraceWideKernel       traceWideKernel(bool anyHit, bool count, bool nearestFirst, int compact, bool denseLeaves);
TraceClosestKernel    traceClosestKernel(bool count);
TraceShadowKernel     traceShadowKernel(bool count);
ShadowFirstLookKernel shadowFirstLookKernel();
TracePacketKernel     tracePacketKernel(bool anyHit); // RF_EXP_LEGACY_LAYOUTS builds only
HitPointsKernel       hitPointsKernel();
PrimaryStatsKernel    primaryStatsKernel();
IntersectRaysKernel   intersectRaysKernel();
OccludedRaysKernel    occludedRaysKernel();
// rf_shade.hip
SamplePermutationKernel samplePermutationKernel();
RaygenKernel            raygenKernel(bool f32Transcendentals);
ShadeKernel             shadeKernel(bool sorted);
SkyKernel               skyKernel(bool f32Transcendentals);
BounceTotalsKernel      bounceTotalsKernel();
AccumulateKernel        accumulateKernel();
AccumulateRunsKernel    accumulateRunsKernel(uint32_t pixelsPerWorkgroup); // 1, 2 or kAccPixels
TonemapKernel           tonemapKernel();
DeferredLightingKernel  deferredLightingKernel();
} // namespace kern
using namespace kern;
} // namespace rf
