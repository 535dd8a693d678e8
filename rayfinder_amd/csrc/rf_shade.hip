// rf_shade.hip -- everything of the wavefront pipeline that is not traversal (MI355X, gfx950): sample permutation + ray generation (wgsl:42-54,236-245,594-616),
// kShade (albedo, NEE term, cosine bounce: wgsl:189-228,294-319,546-592), kSky (wgsl:212-228,247-275), per-bounce totals, the k-ordered accumulation
// (wgsl:47-57), the display transform (wgsl:59-63,277-285) and the deferred-lighting variant.  Launched from rf_renderer.hip through the accessors at the
// end of this file (rf_kernels.hpp).
#include "rf_kernels.hpp"

namespace rf
{
namespace
{
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
// F32: the opt-in f32 evaluation of sin / cos (renderer option `transcendentals`, rf_device.hpp: tSin / tCos); default: the specified f64 evaluation
template<bool F32>
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
    if (fp.tileValidBefore != nullptr)
    {
        // (round 6) positions in closed form, no atomic: the valid pixels of the shard in local-pixel order, a pixel's samples next to each other (slot order: the queue is the
        // slot array with the pixels outside the image squeezed out) -- or, sample-major, sample k's pixels in one run
#pragma unroll
        for (int k = 0; k < kItems; ++k)
        {
            if (!keep[k]) continue;
            uint32_t sampleIdx, lp;
            slotToSamplePixel(fp, slots[k], sampleIdx, lp);
            const uint32_t tile = tileIds[lp >> 10];
            uint32_t       tileX, tileY;
            fp.divTilesX.divmod(tile, tileY, tileX);
            const uint32_t x0 = tileX * kTileSize, y0 = tileY * kTileSize;
            const uint32_t rank = fp.tileValidBefore[lp >> 10] + validRankInTile(lp & 1023u, min(kTileSize, fp.width - x0), min(kTileSize, fp.height - y0));
            pos[k] = fp.slotGroupShift == kSlotSampleMajor ? sampleIdx * fp.validPixels + rank : rank * fp.numSamples + sampleIdx;
            queue[pos[k]] = slots[k];
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) *queueCount = fp.validPixels * fp.numSamples;
    }
    else
        blockAppend<kItems>(keep, slots, queue, queueCount, sScratch, &pos);
    // pass 2: the rays, written at their queue positions
#pragma unroll
    for (int k = 0; k < kItems; ++k)
    {
        if (!keep[k]) continue;
        const uint32_t x = px[k], y = py[k];
        const uint32_t frame = fp.firstFrame + (fp.samplePerm ? fp.samplePerm[sample[k]] : sample[k]);
        float          nx, ny;
        animatedBlueNoiseN(scene.blueNoise, x, y, frame - fp.divSamplesPerPixel.div(frame) * fp.samplesPerPixel, nx, ny); // (n = frame % spp, wgsl:608, without the software division: FastDiv)

        // fragment centre (wgsl:36-43); v runs down the image
        const float u = (static_cast<float>(x) + 0.5f) / static_cast<float>(fp.width);
        const float v = (static_cast<float>(y) + 0.5f) / static_cast<float>(fp.height);
        const float s = u + nx / static_cast<float>(fp.width);
        const float t = (1.0f - v) + ny / static_cast<float>(fp.height);

        const float phi = 2.0f * kPi * ny;
        const float cosPhi = tCos<F32>(phi), sinPhi = tSin<F32>(phi);
        const float r = rf_sqrt(nx);
        const float lensX = fp.camera.lensRadius * (r * cosPhi);
        const float lensY = fp.camera.lensRadius * (r * sinPhi);
        const Vec3  origin = fp.camera.origin + (lensX * fp.camera.right + lensY * fp.camera.up);
        const Vec3  dir = normalize(fp.camera.lowerLeftCorner + s * fp.camera.horizontal + t * fp.camera.vertical - origin);

        // throughput = 1 and radiance = 0 (wgsl:183-184) are not stored: bounce 1 knows them (kFlagFirstBounce, kSky's
        // first-bounce flag), which saves 32 of the 80 bytes a path costs here and the reads back
        // (non-temporal stores measured neutral here: the kernel is bound by its f64 sin / cos and the normalisation, not by its 28 - 40 bytes per path: profiles/r05_raygen)
        if (!fp.skipOrigins) store3(ps.rayO + pos[k], origin);
        store3(ps.rayD + pos[k], dir);
        store3(ps.noise + pos[k], vec3(nx, cosPhi, sinPhi));
    }
    // primary rays are counted on the host (samples x valid pixels of the shard): one atomic per wave on a single counter
    // was what bound this kernel -- 261 k waves at ~90 same-address atomics/us = 2.9 of its 3.2 ms (MI355X_MICROARCH.md "dequeue")
    (void)counters;
}

// SORTED (option shade_sort_from_bounce): a tile's surviving paths are appended to the next queue in the order of the triangles they hit
// (counting sort over kSortBins ranges of triangle ids in LDS; triangles are in BVH leaf order, so that is an order by region of the
// scene) instead of input order: the 64 rays a wave of the next launches picks up then start close to each other.  The tile still
// occupies ONE contiguous run of the queue, so queue order stays slot order at the scale of 1024 entries (what is indexed by slot --
// the blue-noise triple, the radiance sum -- is touched by the same workgroups as before).  `sortScale`: bin of triangle t =
// (t * sortScale) >> 32.
#if defined(RF_EXP_SHADE_WAVES)
#define RF_SHADE_BOUNDS __launch_bounds__(kBlock, RF_EXP_SHADE_WAVES)
#else
#define RF_SHADE_BOUNDS __launch_bounds__(kBlock) // (SORTED: 154 registers -- 137 before the own-triangle test --, three waves per SIMD; forced into 128 for four it was 2.5 % slower)
#endif
template<bool SORTED>
__global__ RF_SHADE_BOUNDS void kShade(DeviceScene scene, SkyStateGpu sky, SunBasis sunBasis, PathStreams ps, const uint32_t* queue,
                                                  const uint32_t* queueCount, uint32_t* hitQueue, uint32_t* hitCount, uint32_t* missQueue, uint32_t* missSlots,
                                                  uint32_t* missCount, uint32_t* shadowList, uint32_t* shadowListCount, uint32_t bounceFlags, uint32_t sortScale)
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
    // ---- The shadow ray's FIRST candidate occluder is the triangle it starts on (round 5).  The reference pushes the hit point off the surface along the GEOMETRIC normal whatever
    // side the path arrived from (wgsl:511-519), and then asks shadowRay (wgsl:321-368) whether ANY triangle stops the ray towards the sun: wherever the sun stands behind that
    // normal the ray crosses the plane of its own triangle a hair's breadth from its origin -- inside the triangle -- and the reference reports it occluded (half of all surfaces).
    // kShade holds everything that test needs in registers: the origin it has just computed, the sun sample, the triangle's positions -- and, in the spare floats of the shading
    // record, the EXACT box of the triangle's leaf.  kShadeSelfShadow: it applies the leaf's box with the reference's formula, then the reference's triangle test; a ray that this
    // stops is FINISHED here (its NEE term times 0, exactly as the traversal's write-back adds it), and only the positions of the other hits go onto `shadowList`, which the
    // bounce's any-hit launches (kShadowFirstLook, kTraceWide) work through instead of the whole queue.
    // Same visibility bit as the reference's walk, by the argument of the occluder cache (rf_trace.hip): the reference tests this triangle iff its walk reaches the leaf, i.e. iff
    // the boxes of the leaf and of all its ancestors pass; an ancestor's box contains the leaf's and the slab arithmetic is monotone in the planes, so a ray that passes the leaf's
    // own test passes every ancestor's: the reference either reaches this leaf -- and finds this triangle, or an earlier one of the leaf -- or has found another occluder before.
    // Occluded either way.  A ray the test does NOT stop proves nothing and is traced as before.  (Trees whose boxes are not nested, triangles that sit in two leaves or in
    // none, rays that are not class A: no shortcut -- the record's flag is 0 / the ray goes onto the list.)
    const bool selfShadow = (bounceFlags & kShadeSelfShadow) != 0u;
  for (uint32_t tile = blockIdx.x; tile < tiles; tile += gridDim.x)
  {
    // Pass 1: which entries hit, which left the scene -> both output queues are appended FIRST, so that every surviving path
    // knows its position in the next queue before it is shaded: what only the next two launches read (the NEE term) is
    // written there, densely, instead of at the path's slot (whose neighbours are mostly dead by bounce 3).
    bool       isHit[kItems], isMiss[kItems];
    uint32_t   slots[kItems], missEntries[kItems], missSlot[kItems], outPos[kItems], hitTri[kItems];
    // (round 5: the loads of all kItems entries are ASKED FOR before any of them is looked at -- first the queue entries and hit records, then, for the hits, throughput and
    // blue-noise triple.  As one loop with its `continue` and `if (hit)` the compiler waited for each entry's loads inside that entry's own blocks: eight memory round trips one
    // after the other at three waves per SIMD.  An entry that has nothing to load reads element 0 of the stream: one line for the whole wave, and a valid address)
    Vec3       hitRecs[kItems];
#pragma unroll
    for (int k = 0; k < kItems; ++k)
    {
        const uint32_t i = (tile * kItems + k) * kBlock + threadIdx.x;
        const uint32_t iL = i < count ? i : 0u;
        slots[k] = queue[iL];
        hitRecs[k] = SORTED ? load3(ps.hit + iL) : vec3(ps.hit[iL].x, 0.0f, 0.0f); // hit records sit at QUEUE positions (dense)
    }
#pragma unroll
    for (int k = 0; k < kItems; ++k)
    {
        const uint32_t i = (tile * kItems + k) * kBlock + threadIdx.x;
        const bool     valid = i < count;
        const uint32_t tri = valid ? __float_as_uint(hitRecs[k].x) : kMiss;
        slots[k] = valid ? slots[k] : 0u;
        missSlot[k] = slots[k];
        missEntries[k] = valid ? i : 0u; // the miss list holds QUEUE positions: kSky finds the ray's direction and throughput there (and, in a second list, the path's slot:
                                         // kSky then needs nothing of the bounce's queue, which kShadowFirstLook reuses for its list while kSky may still be running)
        hitTri[k] = tri;
        isMiss[k] = valid && tri == kMiss; // the path ends in the sky: evaluated densely by this bounce's kSky launch
        isHit[k] = valid && tri != kMiss;
    }
    if constexpr (SORTED)
    {
        // what pass 2 needs of the hits, read here in INPUT order (coalesced) and handed over in LDS: pass 2 works in the
        // tile's sorted order, where the 64 lanes of a wave would gather from ~57 different lines per stream
        Vec3 thrIn[kItems], noiseIn[kItems];
#pragma unroll
        for (int k = 0; k < kItems; ++k)
        {
            const uint32_t i = (tile * kItems + k) * kBlock + threadIdx.x;
            const uint32_t iL = isHit[k] ? i : 0u;
            thrIn[k] = isFirstBounce ? vec3(1.0f, 1.0f, 1.0f) : load3nt(ps.thr + iL);
            noiseIn[k] = load3nt(ps.noise + iL);
        }
#pragma unroll
        for (int k = 0; k < kItems; ++k)
        {
            if (!isHit[k]) continue;
            const uint32_t l = static_cast<uint32_t>(k) * kBlock + threadIdx.x;
            const Vec3     t = thrIn[k], z = noiseIn[k], hitRec = hitRecs[k];
            sIn[l] = t.x, sIn[kTile + l] = t.y, sIn[2 * kTile + l] = t.z;
            sIn[3 * kTile + l] = z.x, sIn[4 * kTile + l] = z.y, sIn[5 * kTile + l] = z.z;
            sIn[6 * kTile + l] = hitRec.x, sIn[7 * kTile + l] = hitRec.y, sIn[8 * kTile + l] = hitRec.z;
            sIn[9 * kTile + l] = __uint_as_float(slots[k]);
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
    blockAppend<kItems>(isMiss, missEntries, missQueue, missCount, sScratch, nullptr, &missSlot, missSlots);

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
        Vec3   boxLo;          // kShadeSelfShadow: the exact box of the triangle's leaf (the .w of the three position float4s) ...
        float4 boxHi;          // ... {hi.xyz, 1.0f if the box may be used (the triangle sits in exactly one leaf of a tree with nested boxes), else 0}
    };
    const auto fetchRecord = [&](uint32_t tri) {
        ShadeRecord   r;
        const float4* rec = scene.shadeRecords + 8 * static_cast<size_t>(tri);
#if defined(RF_EXP_SHADE_ABLATE) && RF_EXP_SHADE_ABLATE >= 3
        if constexpr (SORTED) rec = scene.shadeRecords + 8 * static_cast<size_t>(tri & 63u); // ablation (timing only): 64 records, all L1 hits
#endif
        // (round 5 measured the same fetch as straight-line code -- eight loads, no branch, every lane fetching A record -- so that entry k + 1's record really is in flight
        // while entry k is shaded (as written here the compiler waits for each group of four loads on the spot): `kShade` +3.5 %.  The kernel is bound by what the fabric
        // delivers, not by the latency of its requests; profiles/r05_shademlp)
        if (selfShadow)
        {
            const float4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
            r.p0 = vec3(q0.x, q0.y, q0.z), r.p1 = vec3(q1.x, q1.y, q1.z), r.p2 = vec3(q2.x, q2.y, q2.z);
            r.boxLo = vec3(q0.w, q1.w, q2.w);
            r.boxHi = rec[7];
        }
        else
        {
            r.p0 = load3(rec), r.p1 = load3(rec + 1), r.p2 = load3(rec + 2);
            r.boxLo = Vec3{}, r.boxHi = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        r.a0 = rec[3], r.a1 = rec[4], r.a2 = rec[5], r.a3 = rec[6];
        return r;
    };
    const auto shade = [&](int k, float hu, float hv, const ShadeRecord& cur, uint32_t out) -> bool {
        const uint32_t i = entryIndex(k);
        (void)i;
        Vec3 shadowOrigin;
        {
            // hit point pushed off the surface along the geometric normal (wgsl:511-519,523-544): origin of
            // the shadow ray and of the next bounce; same arithmetic as the scalar traversal (rf_device.hpp)
            const Vec3 p0 = cur.p0, p1 = cur.p1, p2 = cur.p2;
            const Vec3 e1 = p1 - p0, e2 = p2 - p0;
            const Vec3 hp = offsetRay(p0 + hu * e1 + hv * e2, normalize(cross(e1, e2)));
            store3nt(ps.rayO + out, hp); // (this bounce's origins have been consumed by the closest-hit launch)
            shadowOrigin = hp;
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
        bool       settled = false; // kShadeSelfShadow: the shadow ray is stopped by the triangle it starts on
        if (selfShadow && cur.boxHi.w != 0.0f)
        {
            const RayPrep ray = prepareRay(shadowOrigin, lightDirection);
            if (classifyRay(ray) == kRayPlain)
            {
                const PackedRay pr = packRay(ray);
                float           bn, bf;
                bool            boxNaN;
                slabSingleBounds(pr, cur.boxLo.x, cur.boxLo.y, cur.boxLo.z, cur.boxHi.x, cur.boxHi.y, cur.boxHi.z, bn, bf, boxNaN);
                if (bn <= bf && bf > 0.0f && bn < kTMax) // the reference reaches this leaf (kShadowFirstLook applies the same test to the leaves its cell names)
                {
                    TriangleHit th;
                    settled = intersectTriangle(shadowOrigin, lightDirection, cur.p0, cur.p1, cur.p2, kTMax, th);
                }
            }
        }
        if (settled)
        {
            // the traversal's write-back for an occluded ray (kTraceWide): radiance += (pending * 0) * invPdf -- a sum that keeps its bits unless the product is NaN, or the
            // sum is not in memory yet (bounce 1)
            const Vec3 add = (pend * 0.0f) * __uint_as_float(kSolarInvPdfBits);
            const bool unchanged = !isFirstBounce && add.x == 0.0f && add.y == 0.0f && add.z == 0.0f;
            if (!unchanged)
            {
                const uint32_t slot = slots[k];
                const Vec3     radiance = (isFirstBounce ? vec3(0.0f, 0.0f, 0.0f) : load3(ps.rad + slot)) + add;
                ps.rad[slot] = make_float4(radiance.x, radiance.y, radiance.z, 0.0f);
            }
        }
        else store3nt(ps.pending + out, pend); // read by the shadow launch at the same queue position

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
        return !settled;
    };
    bool needShadow[kItems]; // hits whose shadow ray is still to be traced
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
            needShadow[k] = p < tileHits && shade(k, hits[k].y, hits[k].z, cur, base + p);
            cur = next;
        }
    }
    else
    {
#pragma unroll
        for (int k = 0; k < kItems; ++k) needShadow[k] = false;
#pragma unroll 1
        for (int k = 0; k < kItems; ++k)
        {
            if (!isHit[k]) continue;
            const Vec3 h = load3(ps.hit + entryIndex(k));
            const bool need = shade(k, h.y, h.z, fetchRecord(__float_as_uint(h.x)), outPos[k]);
            // (k is a run-time index in this loop: the flag goes in through selects, not through an indexed register array)
#pragma unroll
            for (int j = 0; j < kItems; ++j) needShadow[j] = j == k ? need : needShadow[j];
        }
    }
    // the shadow rays that are still to be traced, by position in the next queue (= where kShade has put their origin, NEE term and blue-noise triple)
    if (selfShadow) blockAppend<kItems>(needShadow, outPos, shadowList, shadowListCount, sScratch);
    if constexpr (SORTED) __syncthreads(); // (a block that takes another tile refills sIn)
  }
}

// Paths that left the scene at this bounce: radiance += throughput * sky (wgsl:212-228,247-275).  One dense launch per
// bounce over that bounce's miss list (queue positions) instead of a divergent f64 branch inside kShade; it runs right
// after kShade, while the bounce's direction / throughput arrays and its queue are still intact.  All NEE terms of the path
// have been added by then (the shadow launch of the previous bounce is complete).  Grid-stride: the list length is only
// known on the device, and a worst-case grid of empty workgroups per bounce would cost more than the work.
// (F32: see kRaygen)
template<bool F32>
__global__ __launch_bounds__(kBlock) void kSky(SkyStateGpu sky, PathStreams ps, const uint32_t* missSlots, const uint32_t* missQueue, const uint32_t* missCount,
                                                uint32_t firstBounce)
{
    const uint32_t n = *missCount;
    const bool     first = firstBounce != 0u; // left the scene at bounce 1: throughput 1, radiance 0, neither in memory
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
    {
        const uint32_t q = missQueue[i];
        const uint32_t slot = missSlots[i];
        // (all three asked for at once: behind the wave-uniform `first` the compiler waited for the direction before it asked for the other two.  At bounce 1 the two
        // arrays hold nothing yet -- allocated memory, values not used)
        const Vec3     v = load3(ps.rayD + q);
        const Vec3     thrIn = load3(ps.thr + q), radIn = load3(ps.rad + slot);
        const Vec3     thr = first ? vec3(1.0f, 1.0f, 1.0f) : thrIn;
        const Vec3     rad = first ? vec3(0.0f, 0.0f, 0.0f) : radIn;
        const Vec3     s = vec3(sky.sunDirection[0], sky.sunDirection[1], sky.sunDirection[2]);
        const float    theta = tAcos<F32>(v.y);
        const float    gamma = tAcos<F32>(minf(maxf(dot(v, s), -1.0f), 1.0f));
        // cos(gamma) and |cos(theta)| do not depend on the channel: evaluated once instead of three times (same values)
        const float cosGamma = tCos<F32>(gamma), cosTheta = fabsf(tCos<F32>(theta));
        const Vec3  dome = vec3(skyRadiance<F32>(sky, cosTheta, gamma, cosGamma, 0), skyRadiance<F32>(sky, cosTheta, gamma, cosGamma, 1), skyRadiance<F32>(sky, cosTheta, gamma, cosGamma, 2));
        const Vec3  radiance = rad + thr * dome;
        ps.rad[slot] = make_float4(radiance.x, radiance.y, radiance.z, 0.0f);
    }
}

// Queue occupancy per bounce: Q[b-1] paths enter bounce b (closest-hit rays), Q[b] of them hit
// something (shadow rays).  Folded into running totals at the end of every batch.
// `listCounts` / `lookMask`: bounces whose any-hit launch ran behind kShadowFirstLook (bit b) -- Q[b] minus the length of its list is what that kernel answered.
// `selfMask` (bit b): kShade settled part of that bounce's shadow rays itself (kShadeSelfShadow) and left shadowListCounts[b] to the any-hit launches, which then do not count
// rays themselves: all Q[b] of them are counted here; totals[3 K + b] = how many kShade settled.
__global__ void kBounceTotals(const uint32_t* queueCounts, uint32_t numBounces, unsigned long long* totals, const uint32_t* listCounts, unsigned long long lookMask, unsigned long long* lookBatch,
                              const uint32_t* shadowListCounts, unsigned long long selfMask, DeviceCounters* counters)
{
    const uint32_t b = threadIdx.x;
    if (b >= numBounces) return;
    const uint32_t k = min(b, RenderStats::kMaxBounceStats - 1);
    atomicAdd(&totals[k], static_cast<unsigned long long>(queueCounts[kLineWords * b]));
    atomicAdd(&totals[RenderStats::kMaxBounceStats + k], static_cast<unsigned long long>(queueCounts[kLineWords * (b + 1)]));
    const bool self = ((selfMask >> b) & 1ull) != 0ull;
    if (self)
    {
        const unsigned long long rays = queueCounts[kLineWords * (b + 1)];
        atomicAdd(&totals[3 * RenderStats::kMaxBounceStats + k], rays - shadowListCounts[kLineWords * b]);
        atomicAdd(&counters->shadowRays, rays);
    }
    if ((lookMask >> b) & 1ull)
    {
        const unsigned long long rays = self ? shadowListCounts[kLineWords * b] : queueCounts[kLineWords * (b + 1)], answered = rays - listCounts[kLineWords * b];
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

// PIXELS per 64-lane workgroup: the runs of PIXELS pixels are staged in LDS (PIXELS x 3 x (S + 1) floats), so deep batches take fewer pixels per workgroup to keep workgroups
// resident (round 6: 320 spp per batch: 4 pixels = 15 KB, ten workgroups per CU, 3.36 ms; 2 pixels: 2.24 ms; 64 spp: 4 pixels 0.37 ms, 2 pixels 0.43: profiles/r06_raygen)
template<uint32_t PIXELS>
__global__ __launch_bounds__(64) void kAccumulateRuns(FrameParams fp, const uint32_t* tileIds, PathStreams ps, float4* image)
{
    extern __shared__ float sRun[]; // [pixel][channel][sample], rows of S + 1 floats: the twelve lanes that sum walk twelve different banks
    const uint32_t S = fp.numSamples, R = S + 1u, lane = threadIdx.x;
    const uint32_t lp0 = blockIdx.x * PIXELS;
    for (uint32_t px = 0; px < PIXELS; ++px)
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
    if (lane >= PIXELS * 3u) return;
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

} // namespace

namespace kern
{
SamplePermutationKernel samplePermutationKernel() { return kSamplePermutation; }
RaygenKernel            raygenKernel(bool f32) { return f32 ? kRaygen<true> : kRaygen<false>; }
ShadeKernel             shadeKernel(bool sorted) { return sorted ? kShade<true> : kShade<false>; }
SkyKernel               skyKernel(bool f32) { return f32 ? kSky<true> : kSky<false>; }
BounceTotalsKernel      bounceTotalsKernel() { return kBounceTotals; }
AccumulateKernel        accumulateKernel() { return kAccumulate; }
AccumulateRunsKernel    accumulateRunsKernel(uint32_t pixels) { return pixels == 1u ? kAccumulateRuns<1> : pixels == 2u ? kAccumulateRuns<2> : kAccumulateRuns<kAccPixels>; }
TonemapKernel           tonemapKernel() { return kTonemap; }
DeferredLightingKernel  deferredLightingKernel() { return kDeferredLighting; }
} // namespace kern
} // namespace rf
