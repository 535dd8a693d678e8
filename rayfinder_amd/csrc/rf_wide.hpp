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
//     one cache line, four dwordx4 loads, one dependent fetch per two box tests), laid out
//     {c0.lo.xy c0.hi.xy | c0.lo.z c0.hi.z c1.lo.z c1.hi.z | c1.lo.xy c1.hi.xy | word0 word1 - -} so that
//     consecutive float pairs are (x,y) or (z,z): six v_pk_add_f32 + six v_pk_mul_f32 do all twelve slab
//     planes, and the lane holds its ray ONCE ((x,y) as a register pair, z splatted by op_sel) -- the
//     (x,y) (z,x) (y,z) pairing of a plain xyz order needs every component in two pairs, 6 VGPRs more;
//   * the near child (reference order: dirNeg[splitAxis], wgsl:409-417) is handled at once; the
//     far child is pushed together with its tmin only if P holds, and when popped it is accepted
//     iff tmin < rayTMax *then* -- exactly the test the reference performs at pop time;
//   * a leaf is described by its parent's child word, so leaf nodes are never fetched.
//
// Child word: bit 31 = leaf; bits 30..29 = split axis of the record's node (first child word only);
// interior: bits 25..0 = index of the child's wide record.  Leaf: bits 28..26 = min(count-1, 7),
// bits 25..0 = first triangle (count <= 7) or index into the big-leaf table {first triangle, count}
// (count >= 8, or offsets >= 2^26).  A tree with 2^26 or more interior nodes is not representable
// (WideBuild::boxesRegular is cleared: the renderer then uses the 32-byte node kernels).
//
// nodesVisited bookkeeping (the counting build): the reference counts a node when it is visited,
// i.e. root once, the near child at its parent's step, the far child when popped -- also when its
// box test then fails.  The counting build therefore pushes every far child (tmin = +inf when P
// fails) so that visit counts AND the stack high-water mark equal the reference's exactly.
#pragma once

#include "rf_device.hpp"

#include <algorithm>
#include <cmath>
#include <initializer_list>
#include <limits>
#include <vector>

namespace rf
{
constexpr uint32_t kWideLeafBit = 0x80000000u;
constexpr uint32_t kWideIndexBits = 26;  // record / first-triangle / big-leaf index
constexpr uint32_t kWideAxisShift = 29;  // bits 30..29 of the FIRST child word carry the node's split axis
constexpr uint32_t kWideNone = 0xFFFFFFFFu; // scene.rootLeaf when the root is interior
// Which 64-byte quad records a scene gets by default, from WideBuild::quadHalfAreaRatio (how coarse binary16 of absolute coordinates is for
// its boxes).  tools/half_ratio_sweep.py on the atrium at tessellation scales 1 / 2 / 3 / 4 / 6 / 8 = ratios 1.035 / 1.063 / 1.089 / 1.117 /
// 1.166 / 1.223, launch time against the exact quad records:
//     closest-hit   half-precision  -17 / -17 / -14 / -10 /  -3 /  +7 %      local grid  -14 / -15 / -15 / -15 / -13 /  -9 %
//     shadow        half-precision   -7 /  -5 /  -1 / +10 / +28 / +54 %      local grid   -8 /  -8 /  -5 /  -1 /  +4 / +20 %
// (a shadow ray crosses the whole scene: every leaf box it only grazes conservatively costs a triangle line, from HBM in a large scene)
// -> closest-hit: half-precision up to kQuadHalfMaxAreaRatio, local grid beyond; shadow: local grid up to kQuadLocalShadowMaxAreaRatio,
//    exact quad records beyond.
constexpr float    kQuadHalfMaxAreaRatio = 1.075f, kQuadLocalShadowMaxAreaRatio = 1.10f;
constexpr uint32_t kHalfEmptyPlanes = 0x7BFFu | (0xFBFFu << 16); // half-precision quad records: the plane word {lower = +65504, upper = -65504} of an empty slot (an inverted box)
constexpr uint32_t kQuadEmpty = 0xFFFFFFFFu; // quad records: an entry slot that holds no node (its child is a leaf and fills one slot only)
#if defined(RF_EXP_WAVES)
constexpr int      kWideWaves = RF_EXP_WAVES;                 // experiment builds: resident workgroups per CU of kTraceWide
constexpr int      kWideLdsStack = 160 * 1024 / RF_EXP_WAVES / 2048; // as deep as the LDS allows at that occupancy (7: 11, 8: 10)
#else
constexpr int      kWideWaves = 6;
#if defined(RF_EXP_STACK)
constexpr int      kWideLdsStack = RF_EXP_STACK; // experiment builds: a shallower LDS stack (what would it cost to free LDS for other lane state?)
#else
constexpr int      kWideLdsStack = 12;
#endif
#endif
// kWideLdsStack: (child word, tmin) pairs per lane in LDS; a full stack evicts its oldest entries to a per-lane scratch array (kTraceWide, rf_renderer.hip);
// only a ray with more than kWideLdsStack + 36 pending entries is redone by the scalar traversal

struct WideScene
{
    const float4* nodes;     // 4 float4 per interior node
    const float4* compact;   // the same records in the compact-capable layout (see buildWide), or nullptr
    const float4* hot;       // 2 float4 per interior node: the 32-byte records of the all-planes-carried layout (see buildWide), or nullptr
    const float4* own;       // 2 float4 per interior node: the node's own box {lo.x lo.y hi.x hi.y} {lo.z hi.z - -} (read after a pop only)
    const float4* quad;      // 8 float4 per quad record: TWO levels in one 128-byte record (see buildWide), or nullptr
    const uint4*  quadHalf;  // 4 uint4 per quad record: the same records with CONSERVATIVE half-precision planes, 64 bytes (see buildWide), or nullptr
    float         originBound; // quadHalf / quadLocal: rays whose origin has a coordinate beyond this magnitude take the scalar traversal (margin proofs)
    const uint4*  quadLocal; // 4 uint4 per quad record: conservative 8-bit planes on a per-record grid, 64 bytes (see buildWide), or nullptr
    const uint4*  oct;       // 8 uint4 per oct record (128-byte aligned, 112 bytes read): THREE levels per fetch, local-grid planes (see WideBuild::oct), or nullptr
    const uint2*  bigLeaves; // {first triangle, count}
    float4        rootLo;    // root box (w unused)
    float4        rootHi;
    uint32_t      rootLeaf;  // child word of the root if the whole tree is one leaf, else kWideNone
    uint32_t      numRecords;
    // occluder grid of the any-hit launches (kTraceWide, kFlagOccluderCache): a hash table over cells of the scene's space -- entry = the leaf (child word) in
    // which the last shadow ray that left that cell found its occluder, 0: nothing known.  Hints only: any leaf word a kernel wrote is a valid first visit.
    uint32_t*     occGrid;   // or nullptr
    float         occScale;  // cells per unit length
    uint32_t      occMask;   // cells - 1 (a power of two)
    // any-hit launches behind kShadowFirstLook: the queue POSITIONS of the rays still to trace (bit 31: the ray has tried its cell's leaves), or nullptr = every
    // position of the queue; the launch's count argument is then the length of this list
    const uint32_t* rayList;
    // closest-hit launch of bounce 1 under a pinhole camera (kFlagConstOrigin): the one origin of all its rays
    float constOriginX, constOriginY, constOriginZ;
};

struct WideBuild
{
    std::vector<float4> nodes;
    std::vector<float4> compact;       // compact-capable layout of the same records; empty when !compactUsable
    bool                compactUsable = true; // every node's x planes are attained by one of its children (true for boxes built as unions)
    std::vector<float4> hot, own;      // the 32-byte layout (all six planes carried) and the nodes' own boxes; empty when !hotUsable
    bool                hotUsable = true; // every plane of every node is attained by a child AND every index fits 24 bits
    // Quad records (kTraceWide<..., COMPACT = 3>): one 128-byte record per interior node at an EVEN level below the root holds the
    // boxes of the node's (up to four) GRANDCHILDREN -- a child that is a leaf fills one slot with itself -- i.e. two levels of
    // the reference's tree per dependent fetch:
    //     {E0.lo.xy E0.hi.xy | E0.lo.z E0.hi.z E1.lo.z E1.hi.z | E1.lo.xy E1.hi.xy |     entries of the FIRST child  (node + 1)
    //      E2.lo.xy E2.hi.xy | E2.lo.z E2.hi.z E3.lo.z E3.hi.z | E3.lo.xy E3.hi.xy |     entries of the SECOND child (secondChildOffset)
    //      word0 word1 word2 word3 | - - - -}
    // word k: child word of entry k (leaf descriptor as in the 64-byte records, or the index of the entry's own quad record);
    // kQuadEmpty in slots 1 / 3 when the child is a leaf.  Bits 30..29 carry split axes: word0 the node's, word1 the first
    // child's, word3 the second child's.
    // Decision-identical to the reference for the render path.  The reference visits a child c iff P(c) && tmin(c) < rayTMax
    // and then its children g under the same test; boxes are unions, so every plane of c is a plane of one of its children,
    // t = (plane - o) * inv is monotone in the plane under rounding, hence per axis the slab interval of g lies inside that of
    // c:  P(g) => P(c)  and  tmin(c) <= tmin(g).  So "g passes" already implies "c passes" (with the same rayTMax: no triangle
    // is tested between c's visit and its children's tests), a child none of whose children pass contributes no triangle test,
    // and the entries in the order [near child's near, far | far child's near, far] (each level by dirNeg[its own split axis])
    // with the later ones pushed together with their tmin and re-tested against rayTMax when popped are exactly the leaves /
    // subtrees the reference enters, in its order.  Not for the counting build (nodesVisited counts the skipped level).
    std::vector<float4> quad;
    bool                quadUsable = true; // every plane of every interior node is attained by one of its children (unions), indices fit
    // Half-precision quad records (kTraceWide<..., COMPACT = 4>): the quad records with their 24 planes stored as IEEE binary16,
    // 64 bytes -- four 16-byte loads per step instead of seven, half the bytes from L2:
    //     dword 3e + a (entry e = 0..3, axis a = 0..2) = {lo plane in the low half, hi plane in the high half};  dwords 12..15 = the quad record's words.
    // The planes are CONSERVATIVE: lo' <= lo - margin (rounded down to binary16), hi' >= hi + margin (rounded up); never subnormal.
    // The step evaluates t' = fma(plane', 1/d, -(o * 1/d)) in one v_fma_mix_f32 per plane (binary16 source, f32 result: no decode
    // instruction) where the reference evaluates t = ((plane - o) * 1/d) with two roundings.  With u = 2^-24, |o| <= originBound and
    // |plane| <= R (the root box):  t' <= T' + u |o / d| + u |t'|  and  t >= T - 2u |T|  for the real values T' = (plane' - o) / d,
    // T = (plane - o) / d, so  margin >= u (4 originBound + 3 R + margin)  makes every near t' <= its t and every far t' >= its t
    // (margin = 2^-21 (originBound + R): twice that).  NaNs (an axis-parallel ray: inf - inf) drop out of v_min / v_max, i.e. that
    // axis does not constrain: conservative as well.  So a quad-half step accepts a SUPERSET of what the exact step accepts --
    // which is all an interior test has to do, because every LEAF is then tested against its EXACT box (kept in the spare floats
    // of the leaf's first 64-byte triangle record: leafBoxesIntoTriangles) with the reference's formula, a leaf's exact slab
    // interval lies inside every ancestor's (boxes are unions, rounding is monotone), and entries are visited in the same order:
    // the same leaves have their triangles tested, in the same order, against the same rayTMax.
    std::vector<uint4>  quadHalf;
    float               originBound = 0.0f;
    // sum of the surface areas of the half-precision boxes / of the exact boxes, over all entries: how many more boxes a random ray
    // hits because of the binary16 grid (absolute coordinates: ~2^-11 of the coordinate's magnitude).  The renderer uses the half records
    // by default only where this stays below kQuadHalfMaxAreaRatio (a 17 M-triangle scene of centimetre triangles: 1.22 -> the local grid).
    float               quadHalfAreaRatio = 0.0f;
    // Local-grid quad records (kTraceWide<..., COMPACT = 5>): the same idea with the planes as 8-bit steps of a PER-RECORD power-of-two grid,
    // for scenes whose triangles are small against their coordinates (where binary16 of absolute coordinates is too coarse):
    //     {anchor.x anchor.y anchor.z scale.x} {scale.y scale.z X0 X1} {Y0 Y1 Z0 Z1} {word0..3}
    // anchor: a point below every lower plane of the record; scale = 2^e per axis (f32); axis words: X0 = {lo0 hi0 lo1 hi1} (one byte each,
    // entries 0 and 1), X1 = entries 2 and 3; plane' = anchor + byte * scale, lo' <= lo - margin, hi' >= hi + margin.
    // The step evaluates t' = fma(1024 + byte, A, B') with A = scale / d (exact: a power of two times 1/d), B' = ((anchor - o) / d) - 1024 A
    // -- "1024 + byte" is the binary16 0x6400 | byte, built by one v_perm_b32 per pair of planes (which also puts the NEAR plane into the
    // low half: the selector depends on the sign of 1/d), and v_fma_mix_f32 takes it straight into the f32 FMA.  Roundings: two in
    // (anchor - o) / d, one in B', one in t', against two in the reference's t: with u = 2^-24 and everything within originBound / R,
    //     |t' - T'| <= u (3.02 |anchor - o| + 1024 scale + |plane' - o|) / |d|,   |t - T| <= 2.01 u |plane - o| / |d|,
    // so  margin >= u (6.03 (originBound + R) + 1024 scale)  keeps every near t' <= its t and every far t' >= its t; 1024 scale <= 16.2 R,
    // i.e. u (46 R + 6) with originBound = 4 R + 1, and the builder leaves 2^-18 (originBound + R) = u (320 R + 64): 6.9 x that.
    std::vector<uint4>  quadLocal;
    float               quadLocalAreaRatio = 0.0f;
    // Oct records (kTraceWide<..., COMPACT = 6>, round 5): the local-grid idea one level further -- one 128-byte-aligned record per interior node reachable
    // from the root in steps of THREE levels holds the boxes of the node's (up to eight) GREAT-GRANDCHILDREN, i.e. three levels of the reference's tree per
    // dependent fetch.  Built for scenes whose tree does not fit the Infinity Cache, on the calibration that a random 128-byte line costs the memory system about
    // what a 64-byte one does (tools/microbench/fetch_calib.hip `wide`, 2-GiB table: 64-byte records 60 G/s = 3.8 TB/s, 128-byte records 51 G/s = 6.5 TB/s) -- and
    // MEASURED SLOWER in the kernel (out-of-cache atrium, bounces 3-8: +17 %; profiles/r05_hbm): those launches sit on the L1 -> L2 request rate (17.7 requests per ray
    // at 78 G/s), a 112-byte record is two 64-byte requests where the quad record is one, and 0.65 x the steps x 2 = 1.2 x the requests.  Kept as an option
    // (oct_from_bounce), bit-identical and tested; not selected by default.  Slot e = 4 c + 2 g + k: child c of the node, its child g, that one's child k; a child or
    // grandchild that is a leaf fills the FIRST slot of its range, the rest of the range is empty.  Layout (uint4 pieces; the eighth is padding):
    //     {anchor.x anchor.y anchor.z scale.x} {scale.y scale.z orderLo orderHi} {X01 X23 X45 X67} {Y01 Y23 Y45 Y67} {Z01 Z23 Z45 Z67} {word0..3} {word4..7}
    // planes as in the local-grid quad records (axis word of slots 2 j, 2 j + 1 = {lo hi lo hi} bytes; an EMPTY slot holds lo = 255, hi = 0 on every axis: it fails
    // the slab test for every ray, so the step has no "is the slot empty" test at all); words = child words without axis bits (a leaf descriptor or the index of
    // the entry's own oct record; kQuadEmpty in empty slots).
    // Visit order.  The reference orders the two children of every node by dirNeg[its split axis] (wgsl:409-417); applied at the three levels inside a record
    // that is a permutation of the eight slots which depends on the record's seven split axes and on the ray's three direction signs only -- so the builder
    // tabulates it: order{Lo,Hi} = four 16-bit fields, one per sign pattern (negX | negY << 1) with negZ = 0; nibble j = 2 c + g of a field = the POSITION (0 = first)
    // at which slot (c, g, 0) is visited, slot (c, g, 1) sits at that position ^ 1.  A ray with negZ = 1 reads the field of the opposite sign pattern and flips
    // every position (^ 7): with all three signs inverted every swap is inverted, i.e. the order is reversed.
    std::vector<uint4>  oct;
    std::vector<uint32_t> octIndexOfNode; // per reference node: its oct record (kQuadEmpty: none)
    float               octAreaRatio = 0.0f; // as quadLocalAreaRatio
    std::vector<uint2>  bigLeaves;
    float4              rootLo, rootHi;
    uint32_t            rootLeaf = kWideNone;
    bool                boxesRegular = true; // every child box finite (|x| < 1e30) with min <= max (see slabPair)
    // every child's box lies inside its parent's (true for boxes built as unions; a hand-made tree may break it).  What the two-levels-per-step records, the
    // conservative layouts' "exact box at the leaf" and the occluder cache's "a leaf whose own box passes is one the reference reaches" all rest on: without it the
    // renderer keeps to the binary records, which repeat the reference's test node by node
    bool                boxesNested = true;
    std::vector<uint32_t> quadIndexOfNode; // per reference node: its quad record, kQuadEmpty for the nodes that have none (leaves, odd levels, unreachable nodes)
};

// Host: 48-byte reference nodes -> wide records.
// binary16 <-> f32 for the half-precision quad records (host side; normal numbers and zero only: the builder never emits a subnormal)
inline float halfBitsToFloat(uint16_t h)
{
    const uint32_t sign = static_cast<uint32_t>(h & 0x8000u) << 16, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu;
    if (e == 0) return bitsFloat(sign); // (+-0; subnormals are not produced)
    return bitsFloat(sign | ((e + 112u) << 23) | (m << 13));
}
// the largest binary16 <= v (down = true) / the smallest binary16 >= v (down = false); |v| must be finite; `ok` = false beyond +-65504
inline uint16_t halfDirected(float v, bool down, bool& ok)
{
    if (!(std::fabs(v) <= 65504.0f)) { ok = false; return 0; }
    if (v == 0.0f) return 0;
    const bool neg = v < 0.0f;
    if (std::fabs(v) < 6.103515625e-05f) // below the smallest normal binary16 (2^-14): 0 or +-2^-14, whichever lies on the requested side
        return (neg == down) ? static_cast<uint16_t>((neg ? 0x8000u : 0u) | 0x0400u) : static_cast<uint16_t>(0);
    const uint32_t b = floatBits(std::fabs(v));
    uint32_t       h = (((b >> 23) - 112u) << 10) | ((b >> 13) & 0x3FFu); // magnitude truncated towards zero
    const bool     inexact = (b & 0x1FFFu) != 0u;
    if (inexact && (neg == down)) ++h; // away from zero on the side that was asked for
    if (h > 0x7BFFu) { ok = false; return 0; }
    return static_cast<uint16_t>((neg ? 0x8000u : 0u) | h);
}

// The EXACT box of every leaf, written into the spare floats of the leaf's FIRST device triangle record (64 bytes: {p0 .} {p1 .} {p2 .}
// {. . . .}):  lo = (record[0].w, record[1].w, record[2].w),  hi = record[3].xyz.  Read by the half-precision quad kernels only.
// Returns false when two leaves claim the same first-triangle slot (a hand-made tree: validateScene only checks that leaf ranges lie inside the
// triangle array, not that they are disjoint): the later leaf's box would overwrite the earlier one's, the half-precision / local-grid kernels
// would cull with a box that is not the leaf's, and the image would depend on the record layout.  The caller keeps those layouts off then.
inline bool leafBoxesIntoTriangles(const BvhNode* nodes, size_t count, float4* triangles /* 4 float4 per triangle */, size_t numTriangles, uint32_t hintLevels = 0,
                                   const std::vector<uint32_t>* quadIndexOfNode = nullptr)
{
    // record[3].w: what the occluder cache (kTraceWide, any-hit) remembers when a ray is stopped in this leaf.  0: the leaf's own child word.  Else the QUAD record
    // index of the record `hintLevels` quad levels above the leaf (1: the record that has the leaf among its entries) -- for scenes whose triangles are small against
    // the sun disc's footprint, where the NEXT ray from the same place is stopped by a neighbour of this triangle rather than by the triangle itself.  Any record is
    // a valid place to start an any-hit ray: the entry only decides what it looks at first.
    // The index comes from buildWide's OWN numbering (WideBuild::quadIndexOfNode: the nodes reachable from the root in two-level steps) and the ancestors from a walk
    // that starts at the root: a hand-made tree may hold interior nodes no parent reaches, or children with two parents (validateScene accepts both), and a
    // numbering recomputed here from node order and depth named the wrong record -- or one past the end -- for them.  A leaf the walk does not reach, or
    // whose ancestor is not a quad record, remembers itself (0).
    std::vector<uint32_t> parent;
    const bool            hints = hintLevels != 0 && quadIndexOfNode != nullptr && quadIndexOfNode->size() == count && count > 0 && nodes[0].triangleCount == 0;
    if (hints)
    {
        parent.assign(count, kQuadEmpty);
        std::vector<uint32_t> todo{0u};
        while (!todo.empty())
        {
            const uint32_t i = todo.back();
            todo.pop_back();
            if (nodes[i].triangleCount != 0) continue;
            for (const size_t c : {static_cast<size_t>(i) + 1, static_cast<size_t>(nodes[i].secondChildOffset)})
                if (c < count && c > i && parent[c] == kQuadEmpty) parent[c] = i, todo.push_back(static_cast<uint32_t>(c));
        }
    }
    std::vector<bool> claimed(numTriangles, false);
    bool              distinct = true;
    for (size_t i = 0; i < count; ++i)
    {
        const BvhNode& n = nodes[i];
        if (n.triangleCount == 0 || n.trianglesOffset >= numTriangles) continue;
        if (claimed[n.trianglesOffset]) distinct = false;
        claimed[n.trianglesOffset] = true;
        float4*  t = triangles + 4 * static_cast<size_t>(n.trianglesOffset);
        uint32_t hint = 0u;
        if (hints && i != 0 && parent[i] != kQuadEmpty)
        {
            const auto& quadIndex = *quadIndexOfNode;
            uint32_t    a = parent[i];
            if (quadIndex[a] == kQuadEmpty && a != 0u) a = parent[a]; // the record that holds the leaf as an entry
            for (uint32_t l = 1; l < hintLevels && a != 0u && a != kQuadEmpty; ++l)
            {
                const uint32_t up = parent[a];
                a = (up == 0u || up == kQuadEmpty) ? 0u : parent[up];
            }
            if (a != 0u && a != kQuadEmpty && quadIndex[a] != kQuadEmpty && quadIndex[a] < (1u << kWideIndexBits)) hint = quadIndex[a]; // (the root is where a ray starts anyway)
        }
        t[0].w = n.aabb.min.x, t[1].w = n.aabb.min.y, t[2].w = n.aabb.min.z;
        t[3] = make_float4(n.aabb.max.x, n.aabb.max.y, n.aabb.max.z, bitsFloat(hint));
    }
    return distinct;
}

inline WideBuild buildWide(const BvhNode* nodes, size_t count)
{
    WideBuild             out;
    std::vector<uint32_t> wideIndex(count, 0);
    uint32_t              numInterior = 0;
    for (size_t i = 0; i < count; ++i)
        if (nodes[i].triangleCount == 0) wideIndex[i] = numInterior++;
    if (numInterior >= (1u << kWideIndexBits)) out.boxesRegular = false; // index field too narrow: not usable
    std::vector<uint32_t> bigLeafOf; // per node: index into bigLeaves + 1 (0: none yet), so that every layout names the same table entry
    auto leafWord = [&](size_t idx) -> uint32_t {
        const BvhNode& n = nodes[idx];
        if (n.triangleCount <= 7 && n.trianglesOffset < (1u << kWideIndexBits)) return kWideLeafBit | ((n.triangleCount - 1) << kWideIndexBits) | n.trianglesOffset;
        if (bigLeafOf.empty()) bigLeafOf.assign(count, 0u);
        if (bigLeafOf[idx] == 0u)
        {
            out.bigLeaves.push_back(make_uint2(n.trianglesOffset, n.triangleCount));
            bigLeafOf[idx] = static_cast<uint32_t>(out.bigLeaves.size());
        }
        return kWideLeafBit | (7u << kWideIndexBits) | (bigLeafOf[idx] - 1u);
    };
    auto childWord = [&](size_t idx) -> uint32_t { return nodes[idx].triangleCount == 0 ? wideIndex[idx] : leafWord(idx); };
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
        // 12 box floats as (x,y) and (z,z) pairs for packed f32 math (see the layout note above)
        w[0] = make_float4(a.aabb.min.x, a.aabb.min.y, a.aabb.max.x, a.aabb.max.y);
        w[1] = make_float4(a.aabb.min.z, a.aabb.max.z, b.aabb.min.z, b.aabb.max.z);
        w[2] = make_float4(b.aabb.min.x, b.aabb.min.y, b.aabb.max.x, b.aabb.max.y);
        // 14 dwords are read per step: 12 box floats + the two child words (the split axis rides in the first word)
        w[3] = make_float4(bitsFloat(childWord(c0) | ((n.splitAxis & 3u) << kWideAxisShift)), bitsFloat(childWord(c1)), 0.0f, 0.0f);
        // Compact-capable layout (kTraceWide<..., COMPACT>).  A node's box is the union of its children's, so each of its planes
        // is attained by (at least) one child; a lane that enters the node straight from its parent's step has the t-values of
        // the node's own planes in hand.  For the two x planes the record therefore names only the OTHER child's value ("inner")
        // and which child that is (bits 30..29 of the second child word), which leaves everything a step needs in 12 dwords:
        //     {innerLoX c0.lo.y innerHiX c0.hi.y | c0.lo.z c0.hi.z c1.lo.z c1.hi.z | word0 c1.lo.y word1 c1.hi.y | outerLoX outerHiX - -}
        // = three dwordx4 instead of 3 + 1 loads per step, i.e. three vector-L1 tag accesses instead of four -- the unit the deep
        // bounces exhaust.  The fourth piece (the shared values themselves) is read only by a lane that arrives from the stack or
        // starts at the root.  Same planes, same (value - o) * inv per plane: bit-identical decisions.
        {
            if (out.compact.empty()) out.compact.resize(out.nodes.size());
            float4*    c = &out.compact[4 * static_cast<size_t>(wideIndex[i])];
            const bool loFromA = a.aabb.min.x == n.aabb.min.x, loFromB = b.aabb.min.x == n.aabb.min.x; // which child attains the node's min.x
            const bool hiFromA = a.aabb.max.x == n.aabb.max.x, hiFromB = b.aabb.max.x == n.aabb.max.x;
            if (!(loFromA || loFromB) || !(hiFromA || hiFromB)) out.compactUsable = false;
            const uint32_t selLo = loFromA ? 1u : 0u, selHi = hiFromA ? 1u : 0u; // 1: child 1 holds the inner value (child 0 shares the node's plane)
            const float    innerLo = selLo ? b.aabb.min.x : a.aabb.min.x, outerLo = selLo ? a.aabb.min.x : b.aabb.min.x;
            const float    innerHi = selHi ? b.aabb.max.x : a.aabb.max.x, outerHi = selHi ? a.aabb.max.x : b.aabb.max.x;
            const uint32_t w0 = floatBits(w[3].x), w1 = floatBits(w[3].y); // the plain record's child words (split axis in word0)
            if ((w1 & (3u << kWideAxisShift)) != 0u) out.compactUsable = false; // (cannot happen: those bits are free in the second word)
            c[0] = make_float4(innerLo, a.aabb.min.y, innerHi, a.aabb.max.y);
            c[1] = make_float4(a.aabb.min.z, a.aabb.max.z, b.aabb.min.z, b.aabb.max.z);
            c[2] = make_float4(bitsFloat(w0), b.aabb.min.y, bitsFloat(w1 | (selLo << kWideAxisShift) | (selHi << (kWideAxisShift + 1))), b.aabb.max.y);
            c[3] = make_float4(outerLo, outerHi, 0.0f, 0.0f);
        }
        // The 32-byte layout (kTraceWide<..., COMPACT = 2>) takes the same idea to all six planes: the lane carries the t-values
        // of the whole box of the node it enters from its parent's step, and a step reads
        //     {innerLoX innerLoY innerHiX innerHiY | innerLoZ innerHiZ word0 word1}
        // = TWO dwordx4 (two vector-L1 tag accesses, one 32-byte piece of a line: two records per 64-byte L2 request, four per
        // 128-byte line) with six selector bits in the child words: bits 25..24 of word0 = {lo.x, lo.y}, bits 25..24 of
        // word1 = {hi.x, hi.y}, bits 30..29 of word1 = {lo.z, hi.z}; selector 1: child 0 shares the node's plane and child 1 holds
        // the inner value.  The index fields shrink to 24 bits for it.  The node's own box sits in a second array that only
        // a lane arriving from the stack reads (the root's box is a kernel argument).
        {
            if (out.hot.empty())
            {
                out.hot.resize(out.nodes.size() / 2);
                out.own.resize(out.nodes.size() / 2);
            }
            float4*        h = &out.hot[2 * static_cast<size_t>(wideIndex[i])];
            float4*        o = &out.own[2 * static_cast<size_t>(wideIndex[i])];
            const float    nv[6] = {n.aabb.min.x, n.aabb.min.y, n.aabb.max.x, n.aabb.max.y, n.aabb.min.z, n.aabb.max.z};
            const float    av[6] = {a.aabb.min.x, a.aabb.min.y, a.aabb.max.x, a.aabb.max.y, a.aabb.min.z, a.aabb.max.z};
            const float    bv[6] = {b.aabb.min.x, b.aabb.min.y, b.aabb.max.x, b.aabb.max.y, b.aabb.min.z, b.aabb.max.z};
            float          inner[6];
            uint32_t       sel[6];
            for (int k = 0; k < 6; ++k)
            {
                const bool fromA = av[k] == nv[k], fromB = bv[k] == nv[k];
                if (!(fromA || fromB)) out.hotUsable = false;
                sel[k] = fromA ? 1u : 0u;
                inner[k] = fromA ? bv[k] : av[k];
            }
            uint32_t w0 = floatBits(w[3].x), w1 = floatBits(w[3].y); // the plain record's child words (split axis in word0)
            if (((w0 | w1) & (3u << 24)) != 0u) out.hotUsable = false; // an index that needs more than 24 bits
            w0 |= (sel[0] << 25) | (sel[1] << 24);
            w1 |= (sel[2] << 25) | (sel[3] << 24) | (sel[4] << 30) | (sel[5] << 29);
            h[0] = make_float4(inner[0], inner[1], inner[2], inner[3]);
            h[1] = make_float4(inner[4], inner[5], bitsFloat(w0), bitsFloat(w1));
            o[0] = make_float4(nv[0], nv[1], nv[2], nv[3]);
            o[1] = make_float4(nv[4], nv[5], 0.0f, 0.0f);
        }
        // the packed slab test assumes ordered, finite boxes (always true for boxes of real triangles)
        const float lo[6] = {a.aabb.min.x, a.aabb.min.y, a.aabb.min.z, b.aabb.min.x, b.aabb.min.y, b.aabb.min.z};
        const float hi[6] = {a.aabb.max.x, a.aabb.max.y, a.aabb.max.z, b.aabb.max.x, b.aabb.max.y, b.aabb.max.z};
        for (int k = 0; k < 6; ++k)
            if (!(std::fabs(lo[k]) < 1e30f && std::fabs(hi[k]) < 1e30f && lo[k] <= hi[k])) out.boxesRegular = false;
        const float nlo[3] = {n.aabb.min.x, n.aabb.min.y, n.aabb.min.z}, nhi[3] = {n.aabb.max.x, n.aabb.max.y, n.aabb.max.z};
        for (int k = 0; k < 6; ++k)
            if (!(lo[k] >= nlo[k % 3] && hi[k] <= nhi[k % 3])) out.boxesNested = false;
    }
    // ---- quad records: the interior nodes reachable from the root in steps of two levels, numbered in node (= depth-first) order
    if (numInterior > 0 && out.boxesRegular && out.boxesNested)
    {
        std::vector<uint32_t>& quadIndex = out.quadIndexOfNode;
        quadIndex.assign(count, kQuadEmpty);
        {
            std::vector<uint8_t>  member(count, 0);
            std::vector<uint32_t> todo{0u};
            while (!todo.empty())
            {
                const uint32_t i = todo.back();
                todo.pop_back();
                member[i] = 1;
                for (const size_t c : {static_cast<size_t>(i) + 1, static_cast<size_t>(nodes[i].secondChildOffset)})
                    if (nodes[c].triangleCount == 0)
                        for (const size_t g : {c + 1, static_cast<size_t>(nodes[c].secondChildOffset)})
                            if (nodes[g].triangleCount == 0) todo.push_back(static_cast<uint32_t>(g));
            }
            uint32_t numQuad = 0;
            for (size_t i = 0; i < count; ++i)
                if (member[i]) quadIndex[i] = numQuad++;
            out.quad.assign(8 * static_cast<size_t>(numQuad), make_float4(0.0f, 0.0f, 0.0f, 0.0f));
        }
        struct Entry
        {
            Aabb     box;
            uint32_t word;
        };
        const auto attained = [&](const BvhNode& n) { // every plane of an interior node's box is a plane of one of its children
            const BvhNode &a = nodes[(&n - nodes) + 1], &b = nodes[n.secondChildOffset];
            return (a.aabb.min.x == n.aabb.min.x || b.aabb.min.x == n.aabb.min.x) && (a.aabb.min.y == n.aabb.min.y || b.aabb.min.y == n.aabb.min.y) &&
                   (a.aabb.min.z == n.aabb.min.z || b.aabb.min.z == n.aabb.min.z) && (a.aabb.max.x == n.aabb.max.x || b.aabb.max.x == n.aabb.max.x) &&
                   (a.aabb.max.y == n.aabb.max.y || b.aabb.max.y == n.aabb.max.y) && (a.aabb.max.z == n.aabb.max.z || b.aabb.max.z == n.aabb.max.z);
        };
        for (size_t i = 0; i < count; ++i)
        {
            if (quadIndex[i] == kQuadEmpty) continue;
            const BvhNode& n = nodes[i];
            Entry          e[4];
            uint32_t       axes[2] = {0u, 0u};
            const size_t   kids[2] = {i + 1, n.secondChildOffset};
            for (int k = 0; k < 2; ++k)
            {
                const BvhNode& c = nodes[kids[k]];
                if (c.triangleCount > 0)
                {
                    e[2 * k] = Entry{c.aabb, leafWord(kids[k])};
                    e[2 * k + 1] = Entry{Aabb{vec3(0.0f, 0.0f, 0.0f), 0.0f, vec3(0.0f, 0.0f, 0.0f), 0.0f}, kQuadEmpty};
                }
                else
                {
                    if (!attained(c)) out.quadUsable = false; // the skipped level must be implied by its children's tests
                    axes[k] = c.splitAxis & 3u;
                    const size_t g[2] = {kids[k] + 1, c.secondChildOffset};
                    for (int j = 0; j < 2; ++j)
                    {
                        const BvhNode& gn = nodes[g[j]];
                        const uint32_t w = gn.triangleCount > 0 ? leafWord(g[j]) : quadIndex[g[j]];
                        if (gn.triangleCount == 0 && w >= (1u << kWideIndexBits)) out.quadUsable = false;
                        e[2 * k + j] = Entry{gn.aabb, w};
                    }
                }
            }
            float4* q = &out.quad[8 * static_cast<size_t>(quadIndex[i])];
            for (int k = 0; k < 2; ++k)
            {
                const Aabb &a = e[2 * k].box, &b = e[2 * k + 1].box;
                q[3 * k] = make_float4(a.min.x, a.min.y, a.max.x, a.max.y);
                q[3 * k + 1] = make_float4(a.min.z, a.max.z, b.min.z, b.max.z);
                q[3 * k + 2] = make_float4(b.min.x, b.min.y, b.max.x, b.max.y);
            }
            const auto tag = [](uint32_t word, uint32_t axis) { return word == kQuadEmpty ? word : (word | (axis << kWideAxisShift)); };
            q[6] = make_float4(bitsFloat(tag(e[0].word, n.splitAxis & 3u)), bitsFloat(tag(e[1].word, axes[0])), bitsFloat(e[2].word), bitsFloat(tag(e[3].word, axes[1])));
        }
    }
    if (out.bigLeaves.empty()) out.bigLeaves.push_back(make_uint2(0, 0));
    if (!out.boxesRegular || !out.boxesNested || numInterior == 0) out.quadUsable = false;
    if (!out.quadUsable) out.quad.clear(), out.quadIndexOfNode.clear();
    if (!out.quad.empty())
    {
        // ---- half-precision quad records (see WideBuild::quadHalf)
        double R = 0.0;
        for (const float c : {out.rootLo.x, out.rootLo.y, out.rootLo.z, out.rootHi.x, out.rootHi.y, out.rootHi.z}) R = std::max(R, static_cast<double>(std::fabs(c)));
        const double bound = 4.0 * R + 1.0;
        out.originBound = static_cast<float>(bound);
        if (static_cast<double>(out.originBound) > bound) out.originBound = std::nextafterf(out.originBound, 0.0f);
        const double margin = (static_cast<double>(out.originBound) + R) * 4.76837158203125e-07; // 2^-21
        const size_t numQuad = out.quad.size() / 8;
        out.quadHalf.assign(4 * numQuad, make_uint4(0u, 0u, 0u, 0u));
        bool ok = true;
        const auto lower = [&](float v) { // the largest f32 <= v - margin, then down to binary16
            const double target = static_cast<double>(v) - margin;
            float        f = static_cast<float>(target);
            if (static_cast<double>(f) > target) f = std::nextafterf(f, -std::numeric_limits<float>::infinity());
            return halfDirected(f, true, ok);
        };
        const auto upper = [&](float v) {
            const double target = static_cast<double>(v) + margin;
            float        f = static_cast<float>(target);
            if (static_cast<double>(f) < target) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
            return halfDirected(f, false, ok);
        };
        double areaExact = 0.0, areaHalf = 0.0;
        const auto area = [](const double lo[3], const double hi[3]) {
            const double ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
            return 2.0 * (ex * ey + ey * ez + ez * ex);
        };
        for (size_t r = 0; r < numQuad && ok; ++r)
        {
            const float4* q = &out.quad[8 * r];
            uint32_t      d[16];
            for (int k = 0; k < 2; ++k) // entries 2k, 2k + 1 (see the quad layout above)
            {
                const float4 a = q[3 * k], z = q[3 * k + 1], b = q[3 * k + 2];
                const float  lo[2][3] = {{a.x, a.y, z.x}, {b.x, b.y, z.z}}, hi[2][3] = {{a.z, a.w, z.y}, {b.z, b.w, z.w}};
                for (int j = 0; j < 2; ++j)
                    for (int ax = 0; ax < 3; ++ax)
                        d[3 * (2 * k + j) + ax] = static_cast<uint32_t>(lower(lo[j][ax])) | (static_cast<uint32_t>(upper(hi[j][ax])) << 16);
            }
            d[12] = floatBits(q[6].x), d[13] = floatBits(q[6].y), d[14] = floatBits(q[6].z), d[15] = floatBits(q[6].w);
            // an EMPTY slot (its child is a leaf and fills one slot only) holds an INVERTED box -- lower planes +65504, upper planes -65504 -- that no ray can pass: its
            // near and far bounds are fma(+-65504, 1/d, b) with the same b, 131 008 |1/d| apart and the wrong way round, so `near <= far` fails -- FOR THE RAYS THAT REACH
            // THESE RECORDS: kTraceWide sends every ray whose origin lies beyond wide.originBound (4 R + 1) or whose 1/d is not of ordinary magnitude (1e-18 ... 1e18) to the
            // scalar traversal before its first step (the refill's `fastRay` / `ordinary` tests).  The invariant depends on that gate: with |o / d| beyond ~2^41 on all three
            // axes both fma results would round to the same b, `near == far` would pass, and -- the step no longer testing the word -- kQuadEmpty & kAxisMask would be followed
            // as a record index.  Under the gate |b| = |o / d| <= (4 R + 1) * 1e18 only where |1/d| itself is huge, and then 131 008 |1/d| >> ulp(b); the fuzz covers a
            // far diagonal camera with these records forced (tests/test_gpu_parity.py: test_far_camera_on_the_half_precision_records).
            // The step's hit test needs no `word != kQuadEmpty` for these records (kTraceWide, COMPACT == 4: two compares and two mask operations per step).
            for (int e = 0; e < 4; ++e)
                if (d[12 + e] == kQuadEmpty)
                    for (int ax = 0; ax < 3; ++ax) d[3 * e + ax] = kHalfEmptyPlanes;
            for (int k = 0; k < 4; ++k) out.quadHalf[4 * r + k] = make_uint4(d[4 * k], d[4 * k + 1], d[4 * k + 2], d[4 * k + 3]);
            for (int e = 0; e < 4 && ok; ++e)
            {
                if (d[12 + e] == kQuadEmpty) continue;
                const int    k = e / 2, j = e % 2;
                const float4 a = q[3 * k], z = q[3 * k + 1], b = q[3 * k + 2];
                const double lo[3] = {j ? b.x : a.x, j ? b.y : a.y, j ? z.z : z.x}, hi[3] = {j ? b.z : a.z, j ? b.w : a.w, j ? z.w : z.y};
                double       hlo[3], hhi[3];
                for (int ax = 0; ax < 3; ++ax)
                {
                    hlo[ax] = halfBitsToFloat(static_cast<uint16_t>(d[3 * e + ax] & 0xFFFFu));
                    hhi[ax] = halfBitsToFloat(static_cast<uint16_t>(d[3 * e + ax] >> 16));
                }
                areaExact += area(lo, hi);
                areaHalf += area(hlo, hhi);
            }
        }
        out.quadHalfAreaRatio = areaExact > 0.0 ? static_cast<float>(areaHalf / areaExact) : 1.0f;
        if (!ok) out.quadHalf.clear(); // a coordinate beyond the binary16 range: the f32 quad records serve
        // ---- local-grid quad records (see WideBuild::quadLocal)
        {
            const double marginL = (static_cast<double>(out.originBound) + R) * 3.814697265625e-06; // 2^-18
            out.quadLocal.assign(4 * numQuad, make_uint4(0u, 0u, 0u, 0u));
            double aExact = 0.0, aLocal = 0.0;
            bool   okL = true;
            for (size_t r = 0; r < numQuad && okL; ++r)
            {
                const float4* q = &out.quad[8 * r];
                const uint32_t words[4] = {floatBits(q[6].x), floatBits(q[6].y), floatBits(q[6].z), floatBits(q[6].w)};
                double         lo[4][3], hi[4][3];
                for (int e = 0; e < 4; ++e)
                {
                    const int    k = e / 2, j = e % 2;
                    const float4 a = q[3 * k], z = q[3 * k + 1], b = q[3 * k + 2];
                    lo[e][0] = j ? b.x : a.x, lo[e][1] = j ? b.y : a.y, lo[e][2] = j ? z.z : z.x;
                    hi[e][0] = j ? b.z : a.z, hi[e][1] = j ? b.w : a.w, hi[e][2] = j ? z.w : z.y;
                }
                float    anchor[3], scale[3];
                uint32_t axisWords[3][2] = {{0u, 0u}, {0u, 0u}, {0u, 0u}};
                double   dlo[4][3], dhi[4][3];
                for (int ax = 0; ax < 3; ++ax)
                {
                    double nlo = std::numeric_limits<double>::infinity(), nhi = -nlo;
                    for (int e = 0; e < 4; ++e)
                        if (words[e] != kQuadEmpty) nlo = std::min(nlo, lo[e][ax]), nhi = std::max(nhi, hi[e][ax]);
                    const double target = nlo - marginL;
                    float        an = static_cast<float>(target);
                    if (static_cast<double>(an) > target) an = std::nextafterf(an, -std::numeric_limits<float>::infinity());
                    const double span = (nhi + marginL) - static_cast<double>(an);
                    int          ex = 0;
                    (void)std::frexp(span / 254.0, &ex); // span / 254 = m * 2^ex, m in [0.5, 1): 2^ex >= span / 254
                    if (!(span > 0.0) || ex < -100 || ex > 100) { okL = false; break; }
                    const double sc = std::ldexp(1.0, ex);
                    anchor[ax] = an, scale[ax] = static_cast<float>(sc);
                    for (int e = 0; e < 4; ++e)
                    {
                        uint32_t bl = 0u, bh = 0u;
                        if (words[e] != kQuadEmpty)
                        {
                            const double fl = std::floor((lo[e][ax] - marginL - static_cast<double>(an)) / sc), ch = std::ceil((hi[e][ax] + marginL - static_cast<double>(an)) / sc);
                            if (!(fl >= 0.0 && ch <= 255.0 && fl <= ch)) { okL = false; break; }
                            bl = static_cast<uint32_t>(fl), bh = static_cast<uint32_t>(ch);
                        }
                        dlo[e][ax] = static_cast<double>(an) + bl * sc, dhi[e][ax] = static_cast<double>(an) + bh * sc;
                        axisWords[ax][e / 2] |= (bl | (bh << 8)) << (16 * (e % 2));
                    }
                }
                if (!okL) break;
                out.quadLocal[4 * r] = make_uint4(floatBits(anchor[0]), floatBits(anchor[1]), floatBits(anchor[2]), floatBits(scale[0]));
                out.quadLocal[4 * r + 1] = make_uint4(floatBits(scale[1]), floatBits(scale[2]), axisWords[0][0], axisWords[0][1]);
                out.quadLocal[4 * r + 2] = make_uint4(axisWords[1][0], axisWords[1][1], axisWords[2][0], axisWords[2][1]);
                out.quadLocal[4 * r + 3] = make_uint4(words[0], words[1], words[2], words[3]);
                for (int e = 0; e < 4; ++e)
                    if (words[e] != kQuadEmpty) aExact += area(lo[e], hi[e]), aLocal += area(dlo[e], dhi[e]);
            }
            out.quadLocalAreaRatio = aExact > 0.0 ? static_cast<float>(aLocal / aExact) : 1.0f;
            if (!okL) out.quadLocal.clear();
        }
    }

    if (!out.quadLocal.empty())
    {
        // ---- oct records (see WideBuild::oct): members = the root and every interior great-grandchild of a member, numbered in node order
        std::vector<uint32_t>& octIndex = out.octIndexOfNode;
        octIndex.assign(count, kQuadEmpty);
        struct Slots
        {
            uint32_t node[8];   // reference node in the slot, kQuadEmpty: empty
            uint32_t axis[7];   // split axes: [0] the member's, [1 + c] child c's, [3 + 2 c + g] grandchild (c, g)'s (0 where that node is a leaf / absent)
        };
        const auto slotsOf = [&](uint32_t i) {
            Slots sl;
            for (uint32_t& v : sl.node) v = kQuadEmpty;
            for (uint32_t& v : sl.axis) v = 0u;
            sl.axis[0] = nodes[i].splitAxis & 3u;
            const size_t kids[2] = {static_cast<size_t>(i) + 1, nodes[i].secondChildOffset};
            for (int c = 0; c < 2; ++c)
            {
                const BvhNode& cn = nodes[kids[c]];
                if (cn.triangleCount > 0) { sl.node[4 * c] = static_cast<uint32_t>(kids[c]); continue; }
                sl.axis[1 + c] = cn.splitAxis & 3u;
                const size_t gk[2] = {kids[c] + 1, cn.secondChildOffset};
                for (int g = 0; g < 2; ++g)
                {
                    const BvhNode& gn = nodes[gk[g]];
                    if (gn.triangleCount > 0) { sl.node[4 * c + 2 * g] = static_cast<uint32_t>(gk[g]); continue; }
                    sl.axis[3 + 2 * c + g] = gn.splitAxis & 3u;
                    sl.node[4 * c + 2 * g] = static_cast<uint32_t>(gk[g] + 1);
                    sl.node[4 * c + 2 * g + 1] = gn.secondChildOffset;
                }
            }
            return sl;
        };
        {
            std::vector<uint8_t>  member(count, 0);
            std::vector<uint32_t> todo{0u};
            while (!todo.empty())
            {
                const uint32_t i = todo.back();
                todo.pop_back();
                member[i] = 1;
                const Slots sl = slotsOf(i);
                for (const uint32_t nd : sl.node)
                    if (nd != kQuadEmpty && nodes[nd].triangleCount == 0) todo.push_back(nd);
            }
            uint32_t numOct = 0;
            for (size_t i = 0; i < count; ++i)
                if (member[i]) octIndex[i] = numOct++;
            out.oct.assign(8 * static_cast<size_t>(numOct), make_uint4(0u, 0u, 0u, 0u));
        }
        double R = 0.0;
        for (const float c : {out.rootLo.x, out.rootLo.y, out.rootLo.z, out.rootHi.x, out.rootHi.y, out.rootHi.z}) R = std::max(R, static_cast<double>(std::fabs(c)));
        const double marginL = (static_cast<double>(out.originBound) + R) * 3.814697265625e-06; // 2^-18, as for the local-grid quad records
        const auto   area = [](const double lo[3], const double hi[3]) {
            const double ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
            return 2.0 * (ex * ey + ey * ez + ez * ex);
        };
        double aExact = 0.0, aLocal = 0.0;
        bool   okO = true;
        for (size_t i = 0; i < count && okO; ++i)
        {
            if (octIndex[i] == kQuadEmpty) continue;
            const Slots sl = slotsOf(static_cast<uint32_t>(i));
            uint32_t    words[8];
            double      lo[8][3], hi[8][3];
            for (int e = 0; e < 8; ++e)
            {
                words[e] = kQuadEmpty;
                if (sl.node[e] == kQuadEmpty) continue;
                const BvhNode& en = nodes[sl.node[e]];
                words[e] = en.triangleCount > 0 ? leafWord(sl.node[e]) : octIndex[sl.node[e]];
                if (en.triangleCount == 0 && words[e] >= (1u << kWideIndexBits)) okO = false;
                lo[e][0] = en.aabb.min.x, lo[e][1] = en.aabb.min.y, lo[e][2] = en.aabb.min.z;
                hi[e][0] = en.aabb.max.x, hi[e][1] = en.aabb.max.y, hi[e][2] = en.aabb.max.z;
            }
            float    anchor[3], scale[3];
            uint32_t axisWords[3][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
            double   dlo[8][3], dhi[8][3];
            for (int ax = 0; ax < 3 && okO; ++ax)
            {
                double nlo = std::numeric_limits<double>::infinity(), nhi = -nlo;
                for (int e = 0; e < 8; ++e)
                    if (words[e] != kQuadEmpty) nlo = std::min(nlo, lo[e][ax]), nhi = std::max(nhi, hi[e][ax]);
                const double target = nlo - marginL;
                float        an = static_cast<float>(target);
                if (static_cast<double>(an) > target) an = std::nextafterf(an, -std::numeric_limits<float>::infinity());
                const double span = (nhi + marginL) - static_cast<double>(an);
                int          ex = 0;
                (void)std::frexp(span / 254.0, &ex); // span / 254 = m * 2^ex, m in [0.5, 1): 2^ex >= span / 254
                if (!(span > 0.0) || ex < -100 || ex > 100) { okO = false; break; }
                const double sc = std::ldexp(1.0, ex);
                anchor[ax] = an, scale[ax] = static_cast<float>(sc);
                for (int e = 0; e < 8; ++e)
                {
                    uint32_t bl = 255u, bh = 0u; // an empty slot: no ray passes lo = 255, hi = 0
                    if (words[e] != kQuadEmpty)
                    {
                        const double fl = std::floor((lo[e][ax] - marginL - static_cast<double>(an)) / sc), ch = std::ceil((hi[e][ax] + marginL - static_cast<double>(an)) / sc);
                        if (!(fl >= 0.0 && ch <= 255.0 && fl <= ch)) { okO = false; break; }
                        bl = static_cast<uint32_t>(fl), bh = static_cast<uint32_t>(ch);
                    }
                    dlo[e][ax] = static_cast<double>(an) + bl * sc, dhi[e][ax] = static_cast<double>(an) + bh * sc;
                    axisWords[ax][e / 2] |= (bl | (bh << 8)) << (16 * (e % 2));
                }
            }
            if (!okO) break;
            // visit positions of the eight slots for the four sign patterns with negZ = 0 (see WideBuild::oct)
            uint32_t order[2] = {0u, 0u};
            for (uint32_t pattern = 0; pattern < 4; ++pattern)
            {
                const auto     neg = [&](uint32_t axis) -> uint32_t { return axis < 2u ? (pattern >> axis) & 1u : 0u; };
                const uint32_t sN = neg(sl.axis[0]);
                uint32_t       field = 0u;
                for (uint32_t c = 0; c < 2; ++c)
                    for (uint32_t g = 0; g < 2; ++g)
                    {
                        const uint32_t position = ((c ^ sN) << 2) | ((g ^ neg(sl.axis[1 + c])) << 1) | (0u ^ neg(sl.axis[3 + 2 * c + g]));
                        field |= position << (4 * (2 * c + g));
                    }
                order[pattern / 2] |= field << (16 * (pattern % 2));
            }
            uint4* o = &out.oct[8 * static_cast<size_t>(octIndex[i])];
            o[0] = make_uint4(floatBits(anchor[0]), floatBits(anchor[1]), floatBits(anchor[2]), floatBits(scale[0]));
            o[1] = make_uint4(floatBits(scale[1]), floatBits(scale[2]), order[0], order[1]);
            o[2] = make_uint4(axisWords[0][0], axisWords[0][1], axisWords[0][2], axisWords[0][3]);
            o[3] = make_uint4(axisWords[1][0], axisWords[1][1], axisWords[1][2], axisWords[1][3]);
            o[4] = make_uint4(axisWords[2][0], axisWords[2][1], axisWords[2][2], axisWords[2][3]);
            o[5] = make_uint4(words[0], words[1], words[2], words[3]);
            o[6] = make_uint4(words[4], words[5], words[6], words[7]);
            for (int e = 0; e < 8; ++e)
                if (words[e] != kQuadEmpty) aExact += area(lo[e], hi[e]), aLocal += area(dlo[e], dhi[e]);
        }
        out.octAreaRatio = aExact > 0.0 ? static_cast<float>(aLocal / aExact) : 1.0f;
        if (!okO) out.oct.clear(), out.octIndexOfNode.clear();
    }
    if (!out.boxesRegular) out.compactUsable = false;
    if (!out.compactUsable || numInterior == 0) out.compact.clear();
    if (!out.boxesRegular) out.hotUsable = false;
    if (!out.hotUsable || numInterior == 0)
    {
        out.hot.clear();
        out.own.clear();
    }
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

// ---------------------------------------------------------------------------------------------
// Both boxes of one record at once.  Precondition: none of the twelve slab products
// (b - o) * inv is NaN.  Then per axis t(lo) <= t(hi) for inv >= 0 and t(hi) <= t(lo) for inv < 0
// (IEEE rounding is monotonic; +-inf included), so the reference's sign-selected near/far planes
// equal min(t(lo), t(hi)) / max(t(lo), t(hi)) bit for bit, its ternary max/min chains equal the
// hardware max3/min3, and its pairwise early-outs
//     !(tmin > tymax) && !(tymin > tmax) && !(max(tmin,tymin) > tzmax) && !(tzmin > min(tmax,tymax))
// are together equivalent to  max3(near) <= min3(far)  (each axis has near <= far; the remaining six
// cross pairs are exactly the four tests).  So P and tmin are those of slabBounds(), with 12 packed
// f32 ops + 16 min/max/compare instead of ~70 VALU and no data-dependent branches.
//
// When can a product be NaN, given ordered finite boxes (WideBuild::boxesRegular)?
//   class A  origin finite (|o| < 1e30), 1/direction finite on all axes: never
//            (b - o is finite, finite * finite is not NaN);
//   class B  origin finite, some 1/direction component = +-inf (direction component +-0 or denormal:
//            axis-aligned walls + a blue-noise value of exactly 0 make these ~0.3 % of all rays):
//            only 0 * inf, i.e. when the origin lies EXACTLY on a plane of the box being tested --
//            checked per step (slabPairHasNaN) for these lanes only;
//   class C  anything else (NaN direction, non-finite origin): always assumed.
// A ray for which a NaN is possible (class C) or occurs (class B) is redone whole by the
// reference-ordered scalar traversal (kTraceWide), so results stay bit-identical in every case.
// Bit-equality with slabBounds() is checked on the GPU by tests/test_gpu_parity.py.
// ---------------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));

struct PackedRay
{
    v2f   oXY, iXY; // origin and 1 / direction: x, y as a register pair
    float oZ, iZ;   // z (splatted into both halves of a packed operand)
};

enum RayClass : uint32_t
{
    kRayPlain = 0,    // class A
    kRayHasInf = 1,   // class B
    kRayIrregular = 2 // class C
};

__device__ __forceinline__ uint32_t classifyRay(const RayPrep& r)
{
    const bool fo = fabsf(r.origin.x) < 1e30f && fabsf(r.origin.y) < 1e30f && fabsf(r.origin.z) < 1e30f; // false for NaN
    const bool nanInv = r.invDir.x != r.invDir.x || r.invDir.y != r.invDir.y || r.invDir.z != r.invDir.z;
    if (!fo || nanInv) return kRayIrregular;
    const bool fi = __builtin_isfinite(r.invDir.x) && __builtin_isfinite(r.invDir.y) && __builtin_isfinite(r.invDir.z);
    return fi ? kRayPlain : kRayHasInf;
}

__device__ __forceinline__ PackedRay packRay(const RayPrep& r)
{
    PackedRay p;
    p.oXY = v2f{r.origin.x, r.origin.y};
    p.iXY = v2f{r.invDir.x, r.invDir.y};
    p.oZ = r.origin.z;
    p.iZ = r.invDir.z;
    return p;
}

// q0 = {c0.lo.xy, c0.hi.xy}  q1 = {c0.lo.z, c0.hi.z, c1.lo.z, c1.hi.z}  q2 = {c1.lo.xy, c1.hi.xy}
// slabPairBounds: max3(near) / min3(far) of both boxes; slabPair: P and tmin from them.
__device__ __forceinline__ void slabPairBounds(const PackedRay& r, float4 q0, float4 q1, float4 q2, float& near0, float& far0, float& near1, float& far1)
{
    const v2f oZZ = v2f{r.oZ, r.oZ}, iZZ = v2f{r.iZ, r.iZ};
    const v2f a = (v2f{q0.x, q0.y} - r.oXY) * r.iXY; // c0: t(lo.x), t(lo.y)
    const v2f b = (v2f{q0.z, q0.w} - r.oXY) * r.iXY; // c0: t(hi.x), t(hi.y)
    const v2f c = (v2f{q1.x, q1.y} - oZZ) * iZZ;     // c0: t(lo.z), t(hi.z)
    const v2f d = (v2f{q1.z, q1.w} - oZZ) * iZZ;     // c1: t(lo.z), t(hi.z)
    const v2f e = (v2f{q2.x, q2.y} - r.oXY) * r.iXY; // c1: t(lo.x), t(lo.y)
    const v2f f = (v2f{q2.z, q2.w} - r.oXY) * r.iXY; // c1: t(hi.x), t(hi.y)
    near0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(a.x, b.x), __builtin_fminf(a.y, b.y)), __builtin_fminf(c.x, c.y));
    far0 = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(a.x, b.x), __builtin_fmaxf(a.y, b.y)), __builtin_fmaxf(c.x, c.y));
    near1 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(e.x, f.x), __builtin_fminf(e.y, f.y)), __builtin_fminf(d.x, d.y));
    far1 = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(e.x, f.x), __builtin_fmaxf(e.y, f.y)), __builtin_fmaxf(d.x, d.y));
}
// slabPairBounds that also hands out the four x-plane products (what a lane carries into the child it enters, compact-capable records)
__device__ __forceinline__ void slabPairBoundsX(const PackedRay& r, float4 q0, float4 q1, float4 q2, float& near0, float& far0, float& near1, float& far1,
                                                float& c0LoX, float& c0HiX, float& c1LoX, float& c1HiX)
{
    const v2f oZZ = v2f{r.oZ, r.oZ}, iZZ = v2f{r.iZ, r.iZ};
    const v2f a = (v2f{q0.x, q0.y} - r.oXY) * r.iXY;
    const v2f b = (v2f{q0.z, q0.w} - r.oXY) * r.iXY;
    const v2f c = (v2f{q1.x, q1.y} - oZZ) * iZZ;
    const v2f d = (v2f{q1.z, q1.w} - oZZ) * iZZ;
    const v2f e = (v2f{q2.x, q2.y} - r.oXY) * r.iXY;
    const v2f f = (v2f{q2.z, q2.w} - r.oXY) * r.iXY;
    c0LoX = a.x, c0HiX = b.x, c1LoX = e.x, c1HiX = f.x;
    near0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(a.x, b.x), __builtin_fminf(a.y, b.y)), __builtin_fminf(c.x, c.y));
    far0 = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(a.x, b.x), __builtin_fmaxf(a.y, b.y)), __builtin_fmaxf(c.x, c.y));
    near1 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(e.x, f.x), __builtin_fminf(e.y, f.y)), __builtin_fminf(d.x, d.y));
    far1 = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(e.x, f.x), __builtin_fmaxf(e.y, f.y)), __builtin_fmaxf(d.x, d.y));
}
__device__ __forceinline__ void slabPair(const PackedRay& r, float4 q0, float4 q1, float4 q2, bool& ok0, float& tmin0, bool& ok1, float& tmin1)
{
    float far0, far1;
    slabPairBounds(r, q0, q1, q2, tmin0, far0, tmin1, far1);
    ok0 = tmin0 <= far0 && far0 > 0.0f;
    ok1 = tmin1 <= far1 && far1 > 0.0f;
}

// (v_bfi_b32 / v_min_f32 / v_max3_f32 ... are written out: the carried t-values reach this block through phi nodes, the compiler
// no longer knows them to be canonical and would spend twelve `v_max x, x` on quieting them in front of its own min / max;
// none of them is a signalling NaN -- they are products -- and a quiet NaN means a class B ray, which is redone anyway.)
__device__ __forceinline__ float isaMin(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float isaMax(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float isaMax3(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float isaMin3(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
// bit `BIT` of `word` set ? a : b, as a bitwise select under the sign-extended bit (v_bfe_i32 + v_bfi_b32 per pair of selects)
template<int BIT>
__device__ __forceinline__ void swapUnderBit(uint32_t word, float shared, float inner, float& child0, float& child1)
{
    uint32_t m;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(word), "n"(BIT));
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(child0) : "v"(m), "v"(shared), "v"(inner));
    asm("v_bfi_b32 %0, %1, %2, %3" : "=v"(child1) : "v"(m), "v"(inner), "v"(shared));
}
// The compact-capable record (buildWide): q0 = {innerLoX c0.lo.y innerHiX c0.hi.y}  q1 = {c0.lo.z c0.hi.z c1.lo.z c1.hi.z}
// q2 = {word0 c1.lo.y word1 c1.hi.y}; tOuterLo / tOuterHi = (value - o.x) * inv.x of the node's own x planes (carried from the
// parent's step or computed from the record's fourth piece); selLo / selHi: child 1 holds the inner value.  Outputs as
// slabPairBounds plus the four x-plane t-values of the children (what the lane carries into the child it enters).
// (The x lanes of the two products on q2 work on the child words' bit patterns: never read.)
__device__ __forceinline__ void slabPairCompactBounds(const PackedRay& r, float4 q0, float4 q1, float4 q2, float tOuterLo, float tOuterHi, uint32_t word1,
                                                      float& near0, float& far0, float& near1, float& far1, float& c0LoX, float& c0HiX, float& c1LoX, float& c1HiX)
{
    const v2f oZZ = v2f{r.oZ, r.oZ}, iZZ = v2f{r.iZ, r.iZ};
    const v2f a = (v2f{q0.x, q0.y} - r.oXY) * r.iXY; // t(innerLoX), c0: t(lo.y)
    const v2f b = (v2f{q0.z, q0.w} - r.oXY) * r.iXY; // t(innerHiX), c0: t(hi.y)
    const v2f c = (v2f{q1.x, q1.y} - oZZ) * iZZ;     // c0: t(lo.z), t(hi.z)
    const v2f d = (v2f{q1.z, q1.w} - oZZ) * iZZ;     // c1: t(lo.z), t(hi.z)
    const v2f e = (v2f{q2.x, q2.y} - r.oXY) * r.iXY; // -, c1: t(lo.y)
    const v2f f = (v2f{q2.z, q2.w} - r.oXY) * r.iXY; // -, c1: t(hi.y)
    swapUnderBit<kWideAxisShift>(word1, tOuterLo, a.x, c0LoX, c1LoX);
    swapUnderBit<kWideAxisShift + 1>(word1, tOuterHi, b.x, c0HiX, c1HiX);
    near0 = isaMax3(isaMin(c0LoX, c0HiX), isaMin(a.y, b.y), isaMin(c.x, c.y));
    far0 = isaMin3(isaMax(c0LoX, c0HiX), isaMax(a.y, b.y), isaMax(c.x, c.y));
    near1 = isaMax3(isaMin(c1LoX, c1HiX), isaMin(e.y, f.y), isaMin(d.x, d.y));
    far1 = isaMin3(isaMax(c1LoX, c1HiX), isaMax(e.y, f.y), isaMax(d.x, d.y));
}

// The 32-byte record (buildWide): h0 = {innerLoX innerLoY innerHiX innerHiY}, h1 = {innerLoZ innerHiZ word0 word1}.  `own` = the
// t-values (value - o) * inv of the six planes of the node's own box, carried from the parent's step (or computed from the
// own-box array after a pop / from the root box at refill).  Per plane the selector bit says which child shares the node's plane
// (1: child 0) -- the other one gets the record's inner value.  Same planes and the same product per plane as slabPairBounds().
struct BoxT
{
    float loX, loY, hiX, hiY, loZ, hiZ;
};
__device__ __forceinline__ BoxT boxPlaneT(const PackedRay& r, float4 q0, float q1x, float q1y) // q0 = {lo.x lo.y hi.x hi.y}, q1 = {lo.z hi.z}
{
    const v2f oZZ = v2f{r.oZ, r.oZ}, iZZ = v2f{r.iZ, r.iZ};
    const v2f a = (v2f{q0.x, q0.y} - r.oXY) * r.iXY;
    const v2f b = (v2f{q0.z, q0.w} - r.oXY) * r.iXY;
    const v2f c = (v2f{q1x, q1y} - oZZ) * iZZ;
    return BoxT{a.x, a.y, b.x, b.y, c.x, c.y};
}
__device__ __forceinline__ void slabPairHotBounds(const PackedRay& r, float4 h0, float h1x, float h1y, uint32_t w0, uint32_t w1, const BoxT& own,
                                                  float& near0, float& far0, float& near1, float& far1, BoxT& c0, BoxT& c1)
{
    const BoxT inner = boxPlaneT(r, h0, h1x, h1y);
    swapUnderBit<25>(w0, own.loX, inner.loX, c0.loX, c1.loX);
    swapUnderBit<24>(w0, own.loY, inner.loY, c0.loY, c1.loY);
    swapUnderBit<25>(w1, own.hiX, inner.hiX, c0.hiX, c1.hiX);
    swapUnderBit<24>(w1, own.hiY, inner.hiY, c0.hiY, c1.hiY);
    swapUnderBit<30>(w1, own.loZ, inner.loZ, c0.loZ, c1.loZ);
    swapUnderBit<29>(w1, own.hiZ, inner.hiZ, c0.hiZ, c1.hiZ);
    near0 = isaMax3(isaMin(c0.loX, c0.hiX), isaMin(c0.loY, c0.hiY), isaMin(c0.loZ, c0.hiZ));
    far0 = isaMin3(isaMax(c0.loX, c0.hiX), isaMax(c0.loY, c0.hiY), isaMax(c0.loZ, c0.hiZ));
    near1 = isaMax3(isaMin(c1.loX, c1.hiX), isaMin(c1.loY, c1.hiY), isaMin(c1.loZ, c1.hiZ));
    far1 = isaMin3(isaMax(c1.loX, c1.hiX), isaMax(c1.loY, c1.hiY), isaMax(c1.loZ, c1.hiZ));
}
// Class B lanes only: is any of the twelve slab products of the step NaN (0 * inf)?
__device__ __forceinline__ bool boxPairHasNaN(const BoxT& c0, const BoxT& c1)
{
    return __builtin_isunordered(c0.loX, c0.hiX) || __builtin_isunordered(c0.loY, c0.hiY) || __builtin_isunordered(c0.loZ, c0.hiZ) ||
           __builtin_isunordered(c1.loX, c1.hiX) || __builtin_isunordered(c1.loY, c1.hiY) || __builtin_isunordered(c1.loZ, c1.hiZ);
}

// Class B lanes only: is any of the twelve slab products of this compact-capable record NaN (0 * inf)?  (The four x-plane values are
// the step's own; the other eight are recomputed here, in the rare path.)
__device__ __forceinline__ bool slabPairCompactHasNaN(const PackedRay& r, float4 q0, float4 q1, float4 q2, float c0LoX, float c0HiX, float c1LoX, float c1HiX)
{
    const float ay = (q0.y - r.oXY.y) * r.iXY.y, by = (q0.w - r.oXY.y) * r.iXY.y, ey = (q2.y - r.oXY.y) * r.iXY.y, fy = (q2.w - r.oXY.y) * r.iXY.y;
    const float cx = (q1.x - r.oZ) * r.iZ, cy = (q1.y - r.oZ) * r.iZ, dx = (q1.z - r.oZ) * r.iZ, dy = (q1.w - r.oZ) * r.iZ;
    return __builtin_isunordered(c0LoX, c0HiX) || __builtin_isunordered(c1LoX, c1HiX) || __builtin_isunordered(ay, by) || __builtin_isunordered(cx, cy) ||
           __builtin_isunordered(dx, dy) || __builtin_isunordered(ey, fy);
}

// Half-precision quad records: t' = fma(plane', inv, b) with plane' the low / high binary16 half of `h` (converted exactly), one
// rounding (v_fma_mix_f32: no decode instruction).  The S variants take the word from an SGPR (wave-uniform steps).
__device__ __forceinline__ float fmixLo(uint32_t h, float inv, float b) { float r; asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(inv), "v"(b)); return r; }
__device__ __forceinline__ float fmixHi(uint32_t h, float inv, float b) { float r; asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(h), "v"(inv), "v"(b)); return r; }
__device__ __forceinline__ float fmixLoS(uint32_t h, float inv, float b) { float r; asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "s"(h), "v"(inv), "v"(b)); return r; }
__device__ __forceinline__ float fmixHiS(uint32_t h, float inv, float b) { float r; asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "s"(h), "v"(inv), "v"(b)); return r; }
// conservative [near, far] of one entry from its three plane words (x, y, z); NaNs drop out of v_min / v_max (see WideBuild::quadHalf)
// `rx ry rz`: 0 or 16 in bits 4..0 -- the plane word rotated by 16 when 1/d < 0 on that axis, so that its low half is the NEAR plane (the
// reference's sign-selected planes: no per-axis min / max; the conservative planes keep near' < far' by the margin even for a flat box).
template<bool SCALAR>
__device__ __forceinline__ void halfEntryBounds(uint32_t wx, uint32_t wy, uint32_t wz, uint32_t rx, uint32_t ry, uint32_t rz, float ix, float iy, float iz, float bx, float by,
                                                float bz, float& near, float& far)
{
    const uint32_t sx = __builtin_amdgcn_alignbit(wx, wx, rx), sy = __builtin_amdgcn_alignbit(wy, wy, ry), sz = __builtin_amdgcn_alignbit(wz, wz, rz);
    (void)SCALAR;
    near = isaMax3(fmixLo(sx, ix, bx), fmixLo(sy, iy, by), fmixLo(sz, iz, bz));
    far = isaMin3(fmixHi(sx, ix, bx), fmixHi(sy, iy, by), fmixHi(sz, iz, bz));
}
// Local-grid quad records: conservative [near, far] of entry `pair` (0 / 1) of the axis words wx wy wz.  sx sy sz: the v_perm_b32 selectors of
// pair 0 -- 0x00050004 when 1/d >= 0 on that axis (low half = 1024 + lo byte, high half = 1024 + hi byte), 0x00040005 when 1/d < 0 (swapped:
// the low half is always the NEAR plane); pair 1 sits two bytes further (+ 0x00020002).  A = scale / d, B = (anchor - o) / d - 1024 A.
template<int PAIR>
__device__ __forceinline__ void localEntryBounds(uint32_t wx, uint32_t wy, uint32_t wz, uint32_t sx, uint32_t sy, uint32_t sz, float ax, float ay, float az, float bx, float by,
                                                 float bz, float& near, float& far)
{
    constexpr uint32_t kStep = PAIR ? 0x00020002u : 0u;
    const uint32_t     px = __builtin_amdgcn_perm(wx, 0x64646464u, sx + kStep), py = __builtin_amdgcn_perm(wy, 0x64646464u, sy + kStep),
                   pz = __builtin_amdgcn_perm(wz, 0x64646464u, sz + kStep);
    near = isaMax3(fmixLo(px, ax, bx), fmixLo(py, ay, by), fmixLo(pz, az, bz));
    far = isaMin3(fmixHi(px, ax, bx), fmixHi(py, ay, by), fmixHi(pz, az, bz));
}
// The EXACT slab bounds of one box (a leaf's, from its triangle record) in the packed arithmetic of slabPairBounds: same planes, same
// (plane - o) * inv per plane, so the same decisions as the reference's whenever no product is NaN; `hasNaN` says whether one is.
__device__ __forceinline__ void slabSingleBounds(const PackedRay& r, float loX, float loY, float loZ, float hiX, float hiY, float hiZ, float& near, float& far, bool& hasNaN)
{
    const v2f oZZ = v2f{r.oZ, r.oZ}, iZZ = v2f{r.iZ, r.iZ};
    const v2f a = (v2f{loX, loY} - r.oXY) * r.iXY; // t(lo.x), t(lo.y)
    const v2f b = (v2f{hiX, hiY} - r.oXY) * r.iXY; // t(hi.x), t(hi.y)
    const v2f c = (v2f{loZ, hiZ} - oZZ) * iZZ;     // t(lo.z), t(hi.z)
    near = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(a.x, b.x), __builtin_fminf(a.y, b.y)), __builtin_fminf(c.x, c.y));
    far = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(a.x, b.x), __builtin_fmaxf(a.y, b.y)), __builtin_fmaxf(c.x, c.y));
    // (pinned here so that the min / max chains stay in the basic block of their products: sunk behind the callers' rare class-B branch, the compiler no longer knows the
    // products to be canonical and spends six v_max x,x on quieting them -- as in slabStep, rf_trace.hip)
    asm volatile("" : "+v"(near), "+v"(far));
    hasNaN = __builtin_isunordered(a.x, a.y) || __builtin_isunordered(b.x, b.y) || __builtin_isunordered(c.x, c.y);
}

// Class B lanes only: is any of the twelve slab products of this record NaN (0 * inf)?
__device__ __forceinline__ bool slabPairHasNaN(const PackedRay& r, float4 q0, float4 q1, float4 q2)
{
    const v2f oZZ = v2f{r.oZ, r.oZ}, iZZ = v2f{r.iZ, r.iZ};
    const v2f a = (v2f{q0.x, q0.y} - r.oXY) * r.iXY;
    const v2f b = (v2f{q0.z, q0.w} - r.oXY) * r.iXY;
    const v2f c = (v2f{q1.x, q1.y} - oZZ) * iZZ;
    const v2f d = (v2f{q1.z, q1.w} - oZZ) * iZZ;
    const v2f e = (v2f{q2.x, q2.y} - r.oXY) * r.iXY;
    const v2f f = (v2f{q2.z, q2.w} - r.oXY) * r.iXY;
    return __builtin_isunordered(a.x, a.y) || __builtin_isunordered(b.x, b.y) || __builtin_isunordered(c.x, c.y) ||
           __builtin_isunordered(d.x, d.y) || __builtin_isunordered(e.x, e.y) || __builtin_isunordered(f.x, f.y);
}

#endif
} // namespace rf
