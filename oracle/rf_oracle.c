/*
 * rf_oracle.c -- CPU ORACLE for the rayfinder hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is a plain-C restatement of the reference's algorithm for the path
 *   ".pt scene -> camera rays -> BVH traversal (Moller-Trumbore) -> Lambertian shade + sun NEE +
 *    Hosek-Wilkie sky -> f32 accumulation".
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; nothing under
 * rayfinder_amd/ (the product) includes, links or calls it.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 *
 * Parity pins (see DESIGN.md "Oracle"):
 *   - sky model: checked bit-for-bit against oracle/_ref/libhwsky_ref.so, which is the reference's
 *     own src/hw-skymodel/hw_skymodel.c compiled unmodified with gcc (no deps).        -> PINNED
 *   - traversal/builder: the reference's own tests (src/tests/bvh.cpp:34-102 BVH == brute force on
 *     Duck, src/tests/aabb.cpp:8-132, src/tests/intersection.cpp:9-28) replayed against this file.
 *     The reference's C++ for these needs glm 0.9.9.8, which is absent from the image and the
 *     mount, so it is unbuildable here and cannot be linked as oracle/_ref.            -> PINNED by
 *     the reference's test scenarios + the glm operation orders stated in SURVEY.md 8(a).
 *   - shading (WGSL rayColor): no reference test exists and WGSL cannot run here       -> PARITY
 *     UNPINNED for the shading stage (no reference-produced vector can exist); what stands in: analytic
 *     known answers and an independent second restatement (tests/analytic_scene.py, float64, no BVH,
 *     no shared code) that this file matches on multi-bounce paths (tests/test_oracle_pins.py).
 *   - deferred-lighting variant (orc_deferred_frame): same status -- PARITY UNPINNED; its G-buffer comes
 *     from primary rays where the reference rasterises, so it is not the reference's pixels by construction.
 *   - where the reference leaves behaviour to its standard library or asserts (triangle order inside
 *     multi-triangle leaves = std::partition; a node without a finite SAH cost), the documented choice is
 *     libstdc++'s partition / "becomes a leaf", identical in the product's host and GPU builders.
 *
 * Floating point policy (the same policy is implemented independently by the HIP product):
 *   - all arithmetic IEEE f32, no FMA contraction (build with -ffp-contract=off), glm 0.9.9.8
 *     operation order: dot=(x+y)+z, normalize = v*(1/sqrt(dot)), min(a,b)=(b<a)?b:a,
 *     max(a,b)=(a<b)?b:a.
 *   - WGSL transcendental builtins (sin cos acos exp pow) have implementation-defined precision;
 *     the documented choice here is the f32 rounding of a SPECIFIED f64 evaluation (fdlibm kernels
 *     written out as IEEE f64 operations: d_exp / d_sin / d_cos / d_acos below; pow(x, 1.5) =
 *     x * sqrt(x) in f64) -- the correctly rounded f32 except about once in 2^29 calls, and the
 *     same bits on the GPU, which runs the same sequences.  sqrt and division are IEEE f32.
 *   - sky_state_new (host C code in the reference) uses libm float functions powf/fmodf exactly
 *     as hw_skymodel.c does.
 */
#include <float.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------ */
/* vec3 with glm 0.9.9.8 semantics (SURVEY.md 8(a) "glm semantics")                            */
/* ------------------------------------------------------------------------------------------ */
typedef struct { float x, y, z; } v3;

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3s(float s) { return V3(s, s, s); }
static inline v3 v3_add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_mul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_scale(float s, v3 a) { return V3(s * a.x, s * a.y, s * a.z); }
static inline float fmin_glm(float a, float b) { return (b < a) ? b : a; }
static inline float fmax_glm(float a, float b) { return (a < b) ? b : a; }
static inline v3 v3_min(v3 a, v3 b) { return V3(fmin_glm(a.x, b.x), fmin_glm(a.y, b.y), fmin_glm(a.z, b.z)); }
static inline v3 v3_max(v3 a, v3 b) { return V3(fmax_glm(a.x, b.x), fmax_glm(a.y, b.y), fmax_glm(a.z, b.z)); }
static inline float v3_dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 v3_cross(v3 x, v3 y)
{
    return V3(x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y);
}
static inline v3 v3_normalize(v3 v) { return v3_scale(1.0f / sqrtf(v3_dot(v, v)), v); }
static inline float v3_get(v3 v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : v.z); }

/* ------------------------------------------------------------------------------------------ */
/* Aabb / BvhNode  (common/aabb.hpp:12-71, common/bvh.hpp:14-21)                               */
/* ------------------------------------------------------------------------------------------ */
typedef struct { float min[3]; float pad0; float max[3]; float pad1; } OrcAabb; /* 32 B */
typedef struct {
    OrcAabb  aabb;
    uint32_t trianglesOffset, secondChildOffset, triangleCount, splitAxis;
} OrcBvhNode; /* 48 B */
typedef struct { v3 v0, v1, v2; } OrcPositions;                     /* 36 B, triangle_attributes.hpp:7-12 */
typedef struct { float p0[3], pad0, p1[3], pad1, p2[3], pad2; } OrcPositionAttribute; /* 48 B */
typedef struct {
    float n0[3], pad0, n1[3], pad1, n2[3], pad2;
    float uv0[2], uv1[2], uv2[2];
    uint32_t textureIdx, pad3;
} OrcVertexAttributes; /* 80 B, pt-format/vertex_attributes.hpp:17-35 */

typedef struct { v3 min, max; } Box;

static inline Box box_default(void)
{ /* aabb.hpp:14-16 */
    Box b = {v3s(FLT_MAX), v3s(-FLT_MAX)};
    return b;
}
static inline Box box_ctor(v3 p1, v3 p2)
{ /* aabb.hpp:19-25: min/max of BOTH arguments */
    Box b = {v3_min(p1, p2), v3_max(p1, p2)};
    return b;
}
static inline v3 box_centroid(Box b) { return v3_scale(0.5f, v3_add(b.min, b.max)); } /* :29 */
static inline v3 box_diagonal(Box b) { return v3_sub(b.max, b.min); }                  /* :31 */
static inline int box_max_dimension(Box b)
{ /* aabb.hpp:33-48 */
    v3 d = box_diagonal(b);
    if (d.x > d.y && d.x > d.z) return 0;
    else if (d.y > d.z) return 1;
    else return 2;
}
static inline Box box_merge_point(Box b, v3 p) { return box_ctor(v3_min(b.min, p), v3_max(b.max, p)); } /* :50-53 */
static inline Box box_merge(Box l, Box r) { return box_ctor(v3_min(l.min, r.min), v3_max(l.max, r.max)); } /* :55-58 */
static inline float box_surface_area(Box b)
{ /* aabb.hpp:60-64 */
    v3 d = box_diagonal(b);
    return 2.0f * ((d.x * d.y + d.x * d.z) + d.y * d.z);
}
static inline Box box_of_triangle(const OrcPositions* t)
{ /* aabb.hpp:66-71 */
    return box_ctor(v3_min(v3_min(t->v0, t->v1), t->v2), v3_max(v3_max(t->v0, t->v1), t->v2));
}

/* exported small helpers so tests can replay src/tests/aabb.cpp */
ORC_API void orc_aabb_merge_point(const float* bmin, const float* bmax, const float* p, float* omin, float* omax)
{
    Box b = {V3(bmin[0], bmin[1], bmin[2]), V3(bmax[0], bmax[1], bmax[2])};
    Box r = box_merge_point(b, V3(p[0], p[1], p[2]));
    omin[0] = r.min.x; omin[1] = r.min.y; omin[2] = r.min.z;
    omax[0] = r.max.x; omax[1] = r.max.y; omax[2] = r.max.z;
}
ORC_API void orc_aabb_merge(const float* amin, const float* amax, const float* bmin, const float* bmax, float* omin, float* omax)
{
    Box a = {V3(amin[0], amin[1], amin[2]), V3(amax[0], amax[1], amax[2])};
    Box b = {V3(bmin[0], bmin[1], bmin[2]), V3(bmax[0], bmax[1], bmax[2])};
    Box r = box_merge(a, b);
    omin[0] = r.min.x; omin[1] = r.min.y; omin[2] = r.min.z;
    omax[0] = r.max.x; omax[1] = r.max.y; omax[2] = r.max.z;
}
ORC_API int orc_aabb_max_dimension(const float* bmin, const float* bmax)
{
    Box b = {V3(bmin[0], bmin[1], bmin[2]), V3(bmax[0], bmax[1], bmax[2])};
    return box_max_dimension(b);
}
ORC_API float orc_aabb_surface_area(const float* bmin, const float* bmax)
{
    Box b = {V3(bmin[0], bmin[1], bmin[2]), V3(bmax[0], bmax[1], bmax[2])};
    return box_surface_area(b);
}

/* ------------------------------------------------------------------------------------------ */
/* buildBvh  (common/bvh.cpp:81-291)                                                           */
/* ------------------------------------------------------------------------------------------ */
typedef struct { Box aabb; v3 centroid; size_t triangleIdx; } Prim; /* bvh.cpp:12-17 */

typedef struct {
    OrcBvhNode* nodes; size_t numNodes, capNodes;
    uint64_t*   triangleIndices;
    int         maxDepth;
} BuildCtx;

static void node_set_box(OrcBvhNode* n, Box b)
{
    n->aabb.min[0] = b.min.x; n->aabb.min[1] = b.min.y; n->aabb.min[2] = b.min.z; n->aabb.pad0 = 0.0f;
    n->aabb.max[0] = b.max.x; n->aabb.max[1] = b.max.y; n->aabb.max[2] = b.max.z; n->aabb.pad1 = 0.0f;
}

static void build_leaf(BuildCtx* c, size_t nodeIdx, Box nodeAabb, const Prim* prims, size_t count, size_t orderedOffset)
{ /* bvh.cpp:57-79 + initLeafNode :31-42 */
    for (size_t i = 0; i < count; ++i) c->triangleIndices[prims[i].triangleIdx] = orderedOffset + i;
    OrcBvhNode* n = &c->nodes[nodeIdx];
    node_set_box(n, nodeAabb);
    n->secondChildOffset = 0;
    n->trianglesOffset = (uint32_t)orderedOffset;
    n->triangleCount = (uint32_t)count;
    n->splitAxis = (uint32_t)-1;
}

static inline size_t bucket_of(const Prim* p, int axis, Box centroidAabb)
{ /* bvh.cpp:152-155 and :211-215 -- 12.0f * (c - min) / (max - min), left to right */
    const size_t numBuckets = 12;
    float  q = (float)numBuckets * (v3_get(p->centroid, axis) - v3_get(centroidAabb.min, axis)) /
              (v3_get(centroidAabb.max, axis) - v3_get(centroidAabb.min, axis));
    size_t b = (size_t)q;
    return b < numBuckets - 1 ? b : numBuckets - 1;
}

static size_t build_recursive(BuildCtx* c, Prim* prims, size_t count, size_t orderedOffset, int depth)
{
    if (depth > c->maxDepth) c->maxDepth = depth;
    const size_t currentNodeIdx = c->numNodes++; /* bvh.cpp:93-94 */
    memset(&c->nodes[currentNodeIdx], 0, sizeof(OrcBvhNode));

    Box nodeAabb = box_default(), centroidAabb = box_default(); /* :98-104 */
    for (size_t i = 0; i < count; ++i) {
        nodeAabb = box_merge(nodeAabb, prims[i].aabb);
        centroidAabb = box_merge_point(centroidAabb, prims[i].centroid);
    }
    const int splitAxis = box_max_dimension(centroidAabb); /* :105 */

    if (box_surface_area(nodeAabb) == 0.0f ||
        v3_get(centroidAabb.min, splitAxis) == v3_get(centroidAabb.max, splitAxis) || count == 1) { /* :110-121 */
        build_leaf(c, currentNodeIdx, nodeAabb, prims, count, orderedOffset);
        return currentNodeIdx;
    }

    size_t splitIdx;
    if (count < 3) {
        /* :126-137 std::nth_element on 2 elements == put the smaller centroid first (libstdc++
         * introselect falls to insertion sort for ranges <= 3). */
        splitIdx = count / 2;
        if (v3_get(prims[1].centroid, splitAxis) < v3_get(prims[0].centroid, splitAxis)) {
            Prim t = prims[0]; prims[0] = prims[1]; prims[1] = t;
        }
    } else {
        enum { numBuckets = 12, numSplits = 11 };
        const size_t maxTrianglesInNode = 255;
        const float  traversalCost = 0.5f, intersectionCost = 1.0f;
        size_t bcount[numBuckets]; Box baabb[numBuckets];
        for (int i = 0; i < numBuckets; ++i) { bcount[i] = 0; baabb[i] = box_default(); }
        for (size_t i = 0; i < count; ++i) { /* :150-158 */
            size_t b = bucket_of(&prims[i], splitAxis, centroidAabb);
            bcount[b]++;
            baabb[b] = box_merge(baabb[b], prims[i].aabb);
        }
        float costs[numSplits];
        for (int i = 0; i < numSplits; ++i) costs[i] = 0.0f;
        { /* :165-172 */
            size_t countBelow = 0; Box aabbBelow = box_default();
            for (int i = 0; i < numSplits; ++i) {
                countBelow += bcount[i];
                aabbBelow = box_merge(aabbBelow, baabb[i]);
                costs[i] += intersectionCost * (float)countBelow * box_surface_area(aabbBelow);
            }
        }
        { /* :174-181 */
            size_t countAbove = 0; Box aabbAbove = box_default();
            for (int i = numSplits; i > 0; --i) {
                countAbove += bcount[i];
                aabbAbove = box_merge(aabbAbove, baabb[i]);
                costs[i - 1] += intersectionCost * (float)countAbove * box_surface_area(aabbAbove);
            }
        }
        float  minCost = FLT_MAX; size_t splitBucketIdx = (size_t)-1; /* :184-193 */
        for (int i = 0; i < numSplits; ++i) {
            if (costs[i] < minCost) { minCost = costs[i]; splitBucketIdx = (size_t)i; }
        }
        const float leafCost = intersectionCost * (float)count;                     /* :203 */
        const float totalCost = traversalCost + minCost / box_surface_area(nodeAabb); /* :204 */
        /* every cost non-finite (areas overflow to inf): the reference asserts 0 < splitIdx < size (:215-216) and would
         * recurse forever in a release build; documented choice (same in the product's two builders): a leaf */
        if (splitBucketIdx != (size_t)-1 && (count > maxTrianglesInNode || totalCost < leafCost)) {
            /* :208-217 std::partition, bidirectional-iterator algorithm of libstdc++ */
            size_t first = 0, last = count;
            for (;;) {
                for (;;) {
                    if (first == last) goto done;
                    else if (bucket_of(&prims[first], splitAxis, centroidAabb) <= splitBucketIdx) ++first;
                    else break;
                }
                --last;
                for (;;) {
                    if (first == last) goto done;
                    else if (!(bucket_of(&prims[last], splitAxis, centroidAabb) <= splitBucketIdx)) --last;
                    else break;
                }
                { Prim t = prims[first]; prims[first] = prims[last]; prims[last] = t; }
                ++first;
            }
        done:
            splitIdx = first;
        } else {
            build_leaf(c, currentNodeIdx, nodeAabb, prims, count, orderedOffset); /* :224-231 */
            return currentNodeIdx;
        }
    }

    build_recursive(c, prims, splitIdx, orderedOffset, depth + 1); /* :238-243 */
    const size_t second = build_recursive(c, prims + splitIdx, count - splitIdx, orderedOffset + splitIdx, depth + 1);

    OrcBvhNode* n = &c->nodes[currentNodeIdx]; /* initInteriorNode :44-55 */
    node_set_box(n, nodeAabb);
    n->secondChildOffset = (uint32_t)second;
    n->trianglesOffset = 0;
    n->triangleCount = 0;
    n->splitAxis = (uint32_t)splitAxis;
    return currentNodeIdx;
}

/* buildBvh, bvh.cpp:263-291.  nodes_out must hold 2*n entries; triangle_indices_out n entries
 * (triangleIndices[src] = dst).  Returns node count; *max_depth (root = 1) if non-NULL. */
ORC_API uint64_t orc_build_bvh(const float* triangles36, uint64_t n, OrcBvhNode* nodes_out,
                               uint64_t* triangle_indices_out, int* max_depth)
{
    const OrcPositions* tris = (const OrcPositions*)triangles36;
    Prim* prims = (Prim*)malloc(sizeof(Prim) * (size_t)n);
    for (size_t i = 0; i < n; ++i) {
        prims[i].aabb = box_of_triangle(&tris[i]);
        prims[i].centroid = box_centroid(prims[i].aabb);
        prims[i].triangleIdx = i;
    }
    BuildCtx c = {nodes_out, 0, (size_t)(2 * n), triangle_indices_out, 0};
    build_recursive(&c, prims, (size_t)n, 0, 1);
    free(prims);
    if (max_depth) *max_depth = c.maxDepth;
    return c.numNodes;
}

/* reorderAttributes, bvh.hpp:36-46, for any fixed-size record */
ORC_API void orc_reorder_attributes(const void* in, void* out, uint64_t n, uint64_t stride, const uint64_t* triangleIndices)
{
    for (uint64_t i = 0; i < n; ++i)
        memcpy((char*)out + triangleIndices[i] * stride, (const char*)in + i * stride, stride);
}

/* ------------------------------------------------------------------------------------------ */
/* Ray / intersection  (common/ray_intersection.cpp)                                            */
/* ------------------------------------------------------------------------------------------ */
static v3 offset_ray(v3 p, v3 n)
{ /* ray_intersection.cpp:17-35 == reference_path_tracer.wgsl:523-544 */
    const float ORIGIN = 1.0f / 32.0f, FLOAT_SCALE = 1.0f / 65536.0f, INT_SCALE = 256.0f;
    const int32_t ox = (int32_t)(INT_SCALE * n.x), oy = (int32_t)(INT_SCALE * n.y), oz = (int32_t)(INT_SCALE * n.z);
    int32_t ix, iy, iz;
    memcpy(&ix, &p.x, 4); memcpy(&iy, &p.y, 4); memcpy(&iz, &p.z, 4);
    ix = (int32_t)((uint32_t)ix + (uint32_t)(p.x < 0 ? -ox : ox));
    iy = (int32_t)((uint32_t)iy + (uint32_t)(p.y < 0 ? -oy : oy));
    iz = (int32_t)((uint32_t)iz + (uint32_t)(p.z < 0 ? -oz : oz));
    v3 po; memcpy(&po.x, &ix, 4); memcpy(&po.y, &iy, 4); memcpy(&po.z, &iz, 4);
    return V3(fabsf(p.x) < ORIGIN ? p.x + FLOAT_SCALE * n.x : po.x,
              fabsf(p.y) < ORIGIN ? p.y + FLOAT_SCALE * n.y : po.y,
              fabsf(p.z) < ORIGIN ? p.z + FLOAT_SCALE * n.z : po.z);
}

typedef struct { v3 p; float t; float u, v; } TriHit;

static int ray_intersect_triangle(v3 ro, v3 rd, v3 p0, v3 p1, v3 p2, float rayTMax, TriHit* hit)
{ /* ray_intersection.cpp:38-90 == wgsl:477-521 */
    const float EPSILON = 0.00001f;
    const v3 e1 = v3_sub(p1, p0), e2 = v3_sub(p2, p0);
    const v3 h = v3_cross(rd, e2);
    const float det = v3_dot(e1, h);
    if (det > -EPSILON && det < EPSILON) return 0;
    const float invDet = 1.0f / det;
    const v3 s = v3_sub(ro, p0);
    const float u = invDet * v3_dot(s, h);
    if (u < 0.0f || u > 1.0f) return 0;
    const v3 q = v3_cross(s, e1);
    const float v = invDet * v3_dot(rd, q);
    if (v < 0.0f || u + v > 1.0f) return 0;
    const float t = invDet * v3_dot(e2, q);
    if (t > EPSILON && t < rayTMax) {
        const v3 p = v3_add(v3_add(p0, v3_scale(u, e1)), v3_scale(v, e2));
        const v3 n = v3_normalize(v3_cross(e1, e2));
        hit->p = offset_ray(p, n);
        hit->t = t; hit->u = u; hit->v = v;
        return 1;
    }
    return 0;
}

typedef struct { v3 origin, invDir; uint32_t dirNeg[3]; } Intersector;

static Intersector make_intersector(v3 ro, v3 rd)
{ /* ray_intersection.cpp:92-99 == wgsl:437-445 */
    Intersector it;
    it.origin = ro;
    it.invDir = V3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
    it.dirNeg[0] = it.invDir.x < 0.0f; it.dirNeg[1] = it.invDir.y < 0.0f; it.dirNeg[2] = it.invDir.z < 0.0f;
    return it;
}

static int ray_intersect_aabb(const Intersector* it, const OrcAabb* aabb, float rayTMax)
{ /* ray_intersection.cpp:101-136 == wgsl:447-475 */
    const float* bounds[2] = {aabb->min, aabb->max};
    float tmin = (bounds[it->dirNeg[0]][0] - it->origin.x) * it->invDir.x;
    float tmax = (bounds[1 - it->dirNeg[0]][0] - it->origin.x) * it->invDir.x;
    const float tymin = (bounds[it->dirNeg[1]][1] - it->origin.y) * it->invDir.y;
    const float tymax = (bounds[1 - it->dirNeg[1]][1] - it->origin.y) * it->invDir.y;
    if ((tmin > tymax) || (tymin > tmax)) return 0;
    tmin = fmax_glm(tymin, tmin);
    tmax = fmin_glm(tymax, tmax);
    const float tzmin = (bounds[it->dirNeg[2]][2] - it->origin.z) * it->invDir.z;
    const float tzmax = (bounds[1 - it->dirNeg[2]][2] - it->origin.z) * it->invDir.z;
    if ((tmin > tzmax) || (tzmin > tmax)) return 0;
    tmin = fmax_glm(tzmin, tmin);
    tmax = fmin_glm(tzmax, tmax);
    return (tmin < rayTMax) && (tmax > 0.0f);
}

ORC_API int orc_ray_intersect_aabb(const float* ro, const float* rd, const float* bmin, const float* bmax, float rayTMax)
{
    Intersector it = make_intersector(V3(ro[0], ro[1], ro[2]), V3(rd[0], rd[1], rd[2]));
    OrcAabb a = {{bmin[0], bmin[1], bmin[2]}, 0.0f, {bmax[0], bmax[1], bmax[2]}, 0.0f};
    return ray_intersect_aabb(&it, &a, rayTMax);
}

ORC_API int orc_ray_intersect_triangle(const float* ro, const float* rd, const float* tri9, float rayTMax, float* p_out, float* t_out)
{
    TriHit h;
    int r = ray_intersect_triangle(V3(ro[0], ro[1], ro[2]), V3(rd[0], rd[1], rd[2]), V3(tri9[0], tri9[1], tri9[2]),
                                   V3(tri9[3], tri9[4], tri9[5]), V3(tri9[6], tri9[7], tri9[8]), rayTMax, &h);
    if (r) { p_out[0] = h.p.x; p_out[1] = h.p.y; p_out[2] = h.p.z; *t_out = h.t; }
    return r;
}

/* The reference's stack is 32 entries with no overflow handling (ray_intersection.cpp:148,194 has
 * only a debug assert; wgsl:327,375 has none).  Documented choice, the same as the product's
 * (rf_device.hpp: kLdsStack + kSpillStack = 96): a 96-entry stack so deeper trees stay well
 * defined; a ray that would need a 97th pending node is ABANDONED with what it has found so far
 * (closest hit: the nearest hit up to there; any hit: unoccluded) and reports stackHigh =
 * ORC_STACK + 1.  The high-water mark is reported so tests can assert it stays < 32 on the scenes
 * used (where all behaviours coincide). */
#define ORC_STACK 96

typedef struct {
    uint32_t nodesVisited, triTests, stackHigh;
    uint32_t triIdx; /* index into the BVH-ordered triangle array of the final hit */
    float    t, u, v;
    v3       p;
} BvhHit;

/* closest hit: ray_intersection.cpp:138-213 == wgsl:370-429.  tri_stride_floats = 9 (CPU
 * Positions, 36 B) or 12 (GPU PositionAttribute, 48 B). */
static int ray_intersect_bvh(v3 ro, v3 rd, const OrcBvhNode* nodes, const float* tris, int tri_stride_floats,
                             float rayTMax, BvhHit* out)
{
    const Intersector it = make_intersector(ro, rd);
    uint32_t nodesVisited = 0, triTests = 0, stackHigh = 0;
    size_t   toVisitOffset = 0, currentNodeIdx = 0;
    size_t   nodesToVisit[ORC_STACK];
    int      didIntersect = 0;
    for (;;) {
        ++nodesVisited;
        const OrcBvhNode* node = &nodes[currentNodeIdx];
        if (ray_intersect_aabb(&it, &node->aabb, rayTMax)) {
            if (node->triangleCount > 0) {
                for (size_t idx = 0; idx < node->triangleCount; ++idx) {
                    const float* tp = tris + (size_t)(node->trianglesOffset + idx) * tri_stride_floats;
                    const int    s = tri_stride_floats / 3;
                    TriHit th;
                    ++triTests;
                    if (ray_intersect_triangle(ro, rd, V3(tp[0], tp[1], tp[2]), V3(tp[s], tp[s + 1], tp[s + 2]),
                                               V3(tp[2 * s], tp[2 * s + 1], tp[2 * s + 2]), rayTMax, &th)) {
                        rayTMax = th.t;
                        didIntersect = 1;
                        out->t = th.t; out->u = th.u; out->v = th.v; out->p = th.p;
                        out->triIdx = (uint32_t)(node->trianglesOffset + idx);
                    }
                }
                if (toVisitOffset == 0) break;
                currentNodeIdx = nodesToVisit[--toVisitOffset];
            } else {
                if (toVisitOffset == ORC_STACK) { stackHigh = ORC_STACK + 1; break; } /* abandoned (see ORC_STACK) */
                if (it.dirNeg[node->splitAxis]) {
                    nodesToVisit[toVisitOffset++] = currentNodeIdx + 1;
                    currentNodeIdx = node->secondChildOffset;
                } else {
                    nodesToVisit[toVisitOffset++] = node->secondChildOffset;
                    currentNodeIdx = currentNodeIdx + 1;
                }
                if (toVisitOffset > stackHigh) stackHigh = (uint32_t)toVisitOffset;
            }
        } else {
            if (toVisitOffset == 0) break;
            currentNodeIdx = nodesToVisit[--toVisitOffset];
        }
    }
    out->nodesVisited = nodesVisited; out->triTests = triTests; out->stackHigh = stackHigh;
    return didIntersect;
}

/* any hit: wgsl:321-368.  Returns 1.0 when nothing is hit. */
static float shadow_ray(v3 ro, v3 rd, const OrcBvhNode* nodes, const float* tris, int tri_stride_floats, float rayTMax,
                        uint32_t* nodesVisited, uint32_t* triTests)
{
    const Intersector it = make_intersector(ro, rd);
    size_t toVisitOffset = 0, currentNodeIdx = 0;
    size_t nodesToVisit[ORC_STACK];
    for (;;) {
        ++*nodesVisited;
        const OrcBvhNode* node = &nodes[currentNodeIdx];
        if (ray_intersect_aabb(&it, &node->aabb, rayTMax)) {
            if (node->triangleCount > 0) {
                for (size_t idx = 0; idx < node->triangleCount; ++idx) {
                    const float* tp = tris + (size_t)(node->trianglesOffset + idx) * tri_stride_floats;
                    const int    s = tri_stride_floats / 3;
                    TriHit th;
                    ++*triTests;
                    if (ray_intersect_triangle(ro, rd, V3(tp[0], tp[1], tp[2]), V3(tp[s], tp[s + 1], tp[s + 2]),
                                               V3(tp[2 * s], tp[2 * s + 1], tp[2 * s + 2]), rayTMax, &th))
                        return 0.0f;
                }
                if (toVisitOffset == 0) break;
                currentNodeIdx = nodesToVisit[--toVisitOffset];
            } else {
                if (toVisitOffset == ORC_STACK) break; /* abandoned (see ORC_STACK): counts as unoccluded */
                if (it.dirNeg[node->splitAxis]) {
                    nodesToVisit[toVisitOffset++] = currentNodeIdx + 1;
                    currentNodeIdx = node->secondChildOffset;
                } else {
                    nodesToVisit[toVisitOffset++] = node->secondChildOffset;
                    currentNodeIdx = currentNodeIdx + 1;
                }
            }
        } else {
            if (toVisitOffset == 0) break;
            currentNodeIdx = nodesToVisit[--toVisitOffset];
        }
    }
    return 1.0f;
}

/* Batch closest-hit for tests: rays = n x (ox,oy,oz,dx,dy,dz). Outputs may be NULL. */
ORC_API void orc_intersect_bvh_batch(const OrcBvhNode* nodes, const float* tris, int tri_stride_floats, const float* rays,
                                     uint64_t n, float rayTMax, uint8_t* hit_out, float* t_out, float* p_out,
                                     uint32_t* tri_out, float* uv_out, uint32_t* nodes_visited_out,
                                     uint32_t* tri_tests_out, uint32_t* stack_high_out)
{
    for (uint64_t i = 0; i < n; ++i) {
        const float* r = rays + 6 * i;
        BvhHit h; memset(&h, 0, sizeof h);
        int did = ray_intersect_bvh(V3(r[0], r[1], r[2]), V3(r[3], r[4], r[5]), nodes, tris, tri_stride_floats, rayTMax, &h);
        if (hit_out) hit_out[i] = (uint8_t)did;
        if (t_out) t_out[i] = did ? h.t : 0.0f;
        if (p_out) { p_out[3 * i] = did ? h.p.x : 0; p_out[3 * i + 1] = did ? h.p.y : 0; p_out[3 * i + 2] = did ? h.p.z : 0; }
        if (tri_out) tri_out[i] = did ? h.triIdx : 0xffffffffu;
        if (uv_out) { uv_out[2 * i] = did ? h.u : 0; uv_out[2 * i + 1] = did ? h.v : 0; }
        if (nodes_visited_out) nodes_visited_out[i] = h.nodesVisited;
        if (tri_tests_out) tri_tests_out[i] = h.triTests;
        if (stack_high_out) stack_high_out[i] = h.stackHigh;
    }
}

/* Batch any-hit for tests. visibility_out[i] = 1.0 / 0.0 */
ORC_API void orc_shadow_batch(const OrcBvhNode* nodes, const float* tris, int tri_stride_floats, const float* rays,
                              uint64_t n, float rayTMax, float* visibility_out)
{
    for (uint64_t i = 0; i < n; ++i) {
        const float* r = rays + 6 * i;
        uint32_t nv = 0, tt = 0;
        visibility_out[i] = shadow_ray(V3(r[0], r[1], r[2]), V3(r[3], r[4], r[5]), nodes, tris, tri_stride_floats, rayTMax, &nv, &tt);
    }
}

/* brute force over all triangles, src/tests/bvh.cpp:15-32 */
ORC_API int orc_brute_force_intersect(const float* tris, int tri_stride_floats, uint64_t ntris, const float* ray6,
                                      float rayTMax, float* t_out)
{
    int did = 0;
    const int s = tri_stride_floats / 3;
    for (uint64_t i = 0; i < ntris; ++i) {
        const float* tp = tris + i * tri_stride_floats;
        TriHit th;
        if (ray_intersect_triangle(V3(ray6[0], ray6[1], ray6[2]), V3(ray6[3], ray6[4], ray6[5]), V3(tp[0], tp[1], tp[2]),
                                   V3(tp[s], tp[s + 1], tp[s + 2]), V3(tp[2 * s], tp[2 * s + 1], tp[2 * s + 2]), rayTMax, &th)) {
            rayTMax = th.t; did = 1; *t_out = th.t;
        }
    }
    return did;
}

/* ------------------------------------------------------------------------------------------ */
/* Camera  (common/camera.cpp:7-52)                                                            */
/* ------------------------------------------------------------------------------------------ */
typedef struct { v3 origin, lowerLeftCorner, horizontal, vertical, up, right; float lensRadius; } OrcCamera; /* 19 floats */

/* Angle::degrees, common/units/angle.hpp:12-15: degrees * pi_f / 180.0f */
ORC_API float orc_degrees_to_radians(float degrees) { return degrees * 3.14159265358979323846f / 180.0f; }

ORC_API void orc_create_camera(const float* origin3, const float* lookAt3, float aperture, float focusDistance,
                               float vfovRadians, float aspectRatio, float* cam19)
{ /* camera.cpp:7-42 */
    const v3 origin = V3(origin3[0], origin3[1], origin3[2]), lookAt = V3(lookAt3[0], lookAt3[1], lookAt3[2]);
    const float theta = vfovRadians;
    const float halfHeight = focusDistance * tanf(0.5f * theta);
    const float halfWidth = aspectRatio * halfHeight;
    const v3 worldUp = V3(0.0f, 1.0f, 0.0f);
    const v3 forward = v3_normalize(v3_sub(lookAt, origin));
    const v3 right = v3_normalize(v3_cross(forward, worldUp));
    const v3 up = v3_cross(right, forward);
    const v3 llc = v3_add(v3_sub(v3_sub(origin, v3_scale(halfWidth, right)), v3_scale(halfHeight, up)),
                          v3_scale(focusDistance, forward));
    const v3 horizontal = v3_scale(2.0f * halfWidth, right);
    const v3 vertical = v3_scale(2.0f * halfHeight, up);
    OrcCamera c = {origin, llc, horizontal, vertical, up, right, 0.5f * aperture};
    memcpy(cam19, &c, sizeof c);
}

static void generate_camera_ray_pinhole(const OrcCamera* c, float u, float v, v3* ro, v3* rd)
{ /* camera.cpp:44-52 */
    *ro = c->origin;
    const v3 d = v3_sub(v3_add(v3_add(c->lowerLeftCorner, v3_scale(u, c->horizontal)), v3_scale(v, c->vertical)), c->origin);
    *rd = v3_normalize(d);
}

ORC_API void orc_generate_camera_ray(const float* cam19, float u, float v, float* ray6)
{
    OrcCamera c; memcpy(&c, cam19, sizeof c);
    v3 ro, rd; generate_camera_ray_pinhole(&c, u, v, &ro, &rd);
    ray6[0] = ro.x; ray6[1] = ro.y; ray6[2] = ro.z; ray6[3] = rd.x; ray6[4] = rd.y; ray6[5] = rd.z;
}

/* The path tracer's default camera: pose fly_camera_controller.hpp:47-52, orientation
 * fly_camera_controller.cpp:138-148 (std::cos/std::sin on float = cosf/sinf), getCamera
 * .cpp:12-22; vfov is passed in (UI default 70 deg, pt/main.cpp:49,314). */
ORC_API void orc_fly_camera(const float* position3, float yawDegrees, float pitchDegrees, float vfovDegrees,
                            float aperture, float focusDistance, float aspectRatio, float* cam19)
{
    const float yaw = orc_degrees_to_radians(yawDegrees), pitch = orc_degrees_to_radians(pitchDegrees);
    const v3 forward = v3_normalize(V3(cosf(yaw) * cosf(pitch), sinf(pitch), sinf(yaw) * cosf(pitch)));
    const v3 pos = V3(position3[0], position3[1], position3[2]);
    const v3 look = v3_add(pos, v3_scale(focusDistance, forward));
    float o[3] = {pos.x, pos.y, pos.z}, l[3] = {look.x, look.y, look.z};
    orc_create_camera(o, l, aperture, focusDistance, orc_degrees_to_radians(vfovDegrees), aspectRatio, cam19);
}

/* bvh-visualizer camera, src/bvh-visualizer/main.cpp:36-55.  NB the x offset is a DOUBLE product
 * (-0.8 * d) cast to f32, the z offset is an f32 product (0.8f * d). */
ORC_API void orc_bvh_visualizer_camera(const OrcBvhNode* nodes, float aspectRatio, float* cam19)
{
    const OrcBvhNode* root = &nodes[0];
    Box rootAabb = box_ctor(V3(root->aabb.min[0], root->aabb.min[1], root->aabb.min[2]),
                            V3(root->aabb.max[0], root->aabb.max[1], root->aabb.max[2]));
    const v3 diag = box_diagonal(rootAabb), cen = box_centroid(rootAabb);
    const int maxDim = box_max_dimension(rootAabb);
    const float d = v3_get(diag, maxDim);
    const v3 off = V3((float)(-0.8 * (double)d), 0.0f, 0.8f * d);
    const v3 origin = v3_sub(cen, off);
    float o[3] = {origin.x, origin.y, origin.z}, l[3] = {cen.x, cen.y, cen.z};
    orc_create_camera(o, l, 0.0f, 1.0f, orc_degrees_to_radians(70.0f), aspectRatio, cam19);
}

/* src/tests/bvh.cpp:46-74 camera: model AABB from triangle vertices, f32 -0.8f product, aspect 1 */
ORC_API void orc_bvh_test_camera(const float* tris, int tri_stride_floats, uint64_t ntris, float* cam19)
{
    Box b = box_default();
    const int s = tri_stride_floats / 3;
    for (uint64_t i = 0; i < ntris; ++i) {
        const float* tp = tris + i * tri_stride_floats;
        b = box_merge_point(b, V3(tp[0], tp[1], tp[2]));
        b = box_merge_point(b, V3(tp[s], tp[s + 1], tp[s + 2]));
        b = box_merge_point(b, V3(tp[2 * s], tp[2 * s + 1], tp[2 * s + 2]));
    }
    const v3 diag = box_diagonal(b), cen = box_centroid(b);
    const int maxDim = box_max_dimension(b);
    const float d = v3_get(diag, maxDim);
    const v3 origin = v3_sub(cen, V3(-0.8f * d, 0.0f, 0.8f * d));
    float o[3] = {origin.x, origin.y, origin.z}, l[3] = {cen.x, cen.y, cen.z};
    orc_create_camera(o, l, 0.0f, 1.0f, orc_degrees_to_radians(70.0f), 1.0f, cam19);
}

/* The bvh-visualizer pixel loop, src/bvh-visualizer/main.cpp:60-78, at any W x H.
 * tris are CPU Positions (stride 9 floats) or GPU PositionAttribute (12). */
ORC_API void orc_bvh_visualize(const OrcBvhNode* nodes, const float* tris, int tri_stride_floats, const float* cam19,
                               int W, int H, int row0, int row1, uint32_t* nodes_visited_out, uint8_t* hit_out,
                               float* t_out, uint32_t* tri_tests_out, uint32_t* stack_high_out)
{
    OrcCamera c; memcpy(&c, cam19, sizeof c);
    for (int i = row0; i < row1; ++i) {
        for (int j = 0; j < W; ++j) {
            const float u = (float)j / (float)W;
            const float v = 1.0f - (float)(i + 1) / (float)H;
            v3 ro, rd; generate_camera_ray_pinhole(&c, u, v, &ro, &rd);
            BvhHit h; memset(&h, 0, sizeof h);
            int did = ray_intersect_bvh(ro, rd, nodes, tris, tri_stride_floats, FLT_MAX, &h);
            const size_t k = (size_t)i * W + j;
            nodes_visited_out[k] = h.nodesVisited;
            if (hit_out) hit_out[k] = (uint8_t)did;
            if (t_out) t_out[k] = did ? h.t : 0.0f;
            if (tri_tests_out) tri_tests_out[k] = h.triTests;
            if (stack_high_out) stack_high_out[k] = h.stackHigh;
        }
    }
}

/* grey value written to bvh-visualizer.png, main.cpp:73-76 */
ORC_API uint32_t orc_bvh_visualizer_pixel(uint32_t nodesVisited)
{
    const float x = 0.01f * (float)nodesVisited;
    const uint32_t p = (uint32_t)((x < 1.0f ? x : 1.0f) * 255.0f);
    return (255u << 24) | (p << 16) | (p << 8) | p;
}

/* ------------------------------------------------------------------------------------------ */
/* Hosek-Wilkie sky  (hw-skymodel/hw_skymodel.c, pt/aligned_sky_state.hpp)                      */
/* ------------------------------------------------------------------------------------------ */
static const float* g_sky_tables = NULL; /* layout: tools/extract_reference_tables.py */
ORC_API void orc_set_sky_tables(const float* tables3630) { g_sky_tables = tables3630; }

static float quintic(const float* data, size_t stride, float t)
{ /* hw_skymodel.c:18-62 (quintic_9: stride 9, quintic_1: stride 1) */
    const float t2 = t * t, t3 = t2 * t, t4 = t2 * t2, t5 = t4 * t;
    const float it = 1.0f - t, it2 = it * it, it3 = it2 * it, it4 = it2 * it2, it5 = it4 * it;
    const float m0 = data[0] * it5;
    const float m1 = data[stride] * 5.0f * it4 * t;
    const float m2 = data[2 * stride] * 10.0f * it3 * t2;
    const float m3 = data[3 * stride] * 10.0f * it2 * t3;
    const float m4 = data[4 * stride] * 5.0f * it * t4;
    const float m5 = data[5 * stride] * t5;
    return m0 + m1 + m2 + m3 + m4 + m5;
}

ORC_API int orc_sky_state_new(float elevation, float turbidity, const float* albedo3, float* state33)
{ /* hw_skymodel.c:141-180 */
    const float PI_F = (float)M_PI;
    if (elevation < 0.0f || elevation > PI_F) return 1;
    if (turbidity < 1.0f || turbidity > 10.0f) return 2;
    for (int i = 0; i < 3; ++i) if (albedo3[i] < 0.0f || albedo3[i] > 1.0f) return 3;
    const float t = powf((elevation / (0.5f * PI_F)), (1.0f / 3.0f));
    const size_t ti = (size_t)turbidity;
    const float  trem = fmodf(turbidity, 1.0f);
    const size_t tmin = ti - 1, tmax = ti < 9 ? ti : 9;
    for (int ch = 0; ch < 3; ++ch) {
        const float albedo = albedo3[ch];
        const float s0 = (1.0f - albedo) * (1.0f - trem), s1 = (1.0f - albedo) * trem;
        const float s2 = albedo * (1.0f - trem), s3 = albedo * trem;
        { /* init_params :64-95 */
            const float* data = g_sky_tables + 1080 * ch;
            const float *p0 = data + 9 * 6 * tmin, *p1 = data + 9 * 6 * tmax;
            const float *p2 = data + (9 * 6 * 10 + 9 * 6 * tmin), *p3 = data + (9 * 6 * 10 + 9 * 6 * tmax);
            float* out = state33 + 9 * ch;
            for (size_t i = 0; i < 9; ++i) {
                out[i] = 0.0f;
                out[i] += s0 * quintic(p0 + i, 9, t);
                out[i] += s1 * quintic(p1 + i, 9, t);
                out[i] += s2 * quintic(p2 + i, 9, t);
                out[i] += s3 * quintic(p3 + i, 9, t);
            }
        }
        { /* init_sky_radiance :97-125 */
            const float* data = g_sky_tables + 3240 + 120 * ch;
            const float *p0 = data + 6 * tmin, *p1 = data + 6 * tmax;
            const float *p2 = data + (6 * 10 + 6 * tmin), *p3 = data + (6 * 10 + 6 * tmax);
            float* out = state33 + 27 + ch;
            *out = 0.0f;
            *out += s0 * quintic(p0, 1, t);
            *out += s1 * quintic(p1, 1, t);
            *out += s2 * quintic(p2, 1, t);
            *out += s3 * quintic(p3, 1, t);
        }
        { /* init_solar_radiance :127-139 */
            const float* data = g_sky_tables + 3600 + 10 * ch;
            state33[30 + ch] = data[tmin] * (1.0f - trem) + data[tmax] * trem;
        }
    }
    return 0;
}

/* sky_state_radiance, hw_skymodel.c:182-222, libm float functions as in the C reference */
ORC_API float orc_sky_state_radiance(const float* state33, float theta, float gamma, int channel)
{
    const float r = state33[27 + channel];
    const float* p = state33 + 9 * channel;
    const float cos_gamma = cosf(gamma), cos_gamma_2 = cos_gamma * cos_gamma, cos_theta = fabsf(cosf(theta));
    const float exp_m = expf(p[4] * gamma), ray_m = cos_gamma_2, mie_m_lhs = 1.0f + cos_gamma_2;
    const float mie_m_rhs = powf(1.0f + p[8] * p[8] - 2.0f * p[8] * cos_gamma, 1.5f);
    const float mie_m = mie_m_lhs / mie_m_rhs, zenith = sqrtf(cos_theta);
    const float radiance_lhs = 1.0f + p[0] * expf(p[1] / (cos_theta + 0.01f));
    const float radiance_rhs = p[2] + p[3] * exp_m + p[5] * ray_m + p[6] * mie_m + p[7] * zenith;
    const float radiance_dist = radiance_lhs * radiance_rhs;
    const float solar_disk_radius = gamma / 0.004450589f;
    const float solar_radiance = solar_disk_radius <= 1.f ? state33[30 + channel] : 0.f;
    return r * radiance_dist + solar_radiance;
}

/* AlignedSkyState ctor, pt/aligned_sky_state.hpp:44-70.  out40: params[27] sky[3] solar[3]
 * pad[3] sunDirection[3] pad */
ORC_API int orc_aligned_sky_state(float turbidity, const float* albedo3, float sunZenithDegrees, float sunAzimuthDegrees, float* out40)
{
    memset(out40, 0, 40 * sizeof(float));
    const float sunZenith = orc_degrees_to_radians(sunZenithDegrees);
    const float sunAzimuth = orc_degrees_to_radians(sunAzimuthDegrees);
    const v3 sd = v3_normalize(V3(sinf(sunZenith) * cosf(sunAzimuth), cosf(sunZenith), -sinf(sunZenith) * sinf(sunAzimuth)));
    out40[36] = sd.x; out40[37] = sd.y; out40[38] = sd.z;
    float st[33];
    int rc = orc_sky_state_new(0.5f * 3.14159265358979323846f - sunZenith, turbidity, albedo3, st);
    if (rc) return rc;
    memcpy(out40, st, 33 * sizeof(float));
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* WGSL path tracer  (pt/reference_path_tracer.wgsl)                                            */
/* ------------------------------------------------------------------------------------------ */
/* WGSL builtins with implementation-defined precision (sin cos acos exp, and the one pow(x, 1.5) of the sky model).  Documented choice
 * since round 3: the f32 rounding of a SPECIFIED f64 evaluation -- the classic fdlibm kernels (Sun Microsystems, freely distributable),
 * written out below as plain sequences of IEEE f64 operations (+ - * / sqrt fma rint ldexp), each within about one ulp(f64) of the true
 * value, i.e. the correctly rounded f32 except about once in 2^29 calls.  The product evaluates the SAME sequences on the GPU
 * (rf_device.hpp), so the two sides agree bit for bit by construction; with the C library's functions on one side and ocml's on the
 * other they differed in the last f64 bit now and then, and once in ~10^9 calls that bit decided an f32 rounding (2 of 20 000 fuzz scenes).
 * pow() proper is kept for the two display-side uses (sRGB -> linear table, built on the host on both sides; tonemap, quantised to 8 bits). */
static double d_exp(double x)
{
    if (x != x) return x;
    if (x > 709.0) return INFINITY;
    if (x < -745.0) return 0.0;
    const double k = rint(x * 1.44269504088896338700e+00);
    double r = fma(-k, 6.93147180369123816490e-01, x);
    r = fma(-k, 1.90821492927058770002e-10, r);
    /* exp(r), |r| <= 0.3466: Taylor polynomial of degree 13, Horner with fma */
    double p = 1.6059043836821613e-10;           /* 1/13! */
    p = fma(p, r, 2.08767569878681e-09);        /* 1/12! */
    p = fma(p, r, 2.505210838544172e-08);       /* 1/11! */
    p = fma(p, r, 2.755731922398589e-07);       /* 1/10! */
    p = fma(p, r, 2.7557319223985893e-06);      /* 1/9! */
    p = fma(p, r, 2.48015873015873e-05);        /* 1/8! */
    p = fma(p, r, 0.0001984126984126984);       /* 1/7! */
    p = fma(p, r, 0.001388888888888889);        /* 1/6! */
    p = fma(p, r, 0.008333333333333333);        /* 1/5! */
    p = fma(p, r, 0.041666666666666664);        /* 1/4! */
    p = fma(p, r, 0.16666666666666666);         /* 1/3! */
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)k);
}
/* sin and cos of r in [-pi/4, pi/4] (fdlibm __kernel_sin / __kernel_cos without the tail argument) */
static double d_ksin(double r)
{
    const double z = r * r;
    const double t = -1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
    return r + (r * z) * t;
}
static double d_kcos(double r)
{
    const double z = r * r;
    const double t = 4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 + z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
    return (1.0 - 0.5 * z) + (z * z) * t;
}
/* argument reduction by pi/2 in three parts (exact products through fma); meant for the |x| <= 2 pi of this renderer, defined for all x */
static double d_reduce(double x, long long* quadrant)
{
    const double k = rint(x * 6.36619772367581382433e-01);
    double r = fma(-k, 1.57079632673412561417e+00, x);
    r = fma(-k, 6.07710050650619224932e-11, r);
    r = fma(-k, 2.02226624879595063154e-21, r);
    *quadrant = (long long)k;
    return r;
}
static double d_sin(double x)
{
    if (!(fabs(x) < 1.0e15)) return x - x; /* inf, NaN (and arguments no f32 angle of this renderer reaches): NaN */
    long long q;
    const double r = d_reduce(x, &q);
    switch (q & 3) { case 0: return d_ksin(r); case 1: return d_kcos(r); case 2: return -d_ksin(r); default: return -d_kcos(r); }
}
static double d_cos(double x)
{
    if (!(fabs(x) < 1.0e15)) return x - x;
    long long q;
    const double r = d_reduce(x, &q);
    switch (q & 3) { case 0: return d_kcos(r); case 1: return -d_ksin(r); case 2: return -d_kcos(r); default: return d_ksin(r); }
}
/* fdlibm __ieee754_acos */
static double d_acos_ratio(double z)
{
    const double p = z * (1.66666666666666657415e-01 + z * (-3.25565818622400915405e-01 + z * (2.01212532134862925881e-01 + z * (-4.00555345006794114027e-02 + z * (7.91534994289814532176e-04 + z * 3.47933107596021167570e-05)))));
    const double q = 1.0 + z * (-2.40339491173441421878e+00 + z * (2.02094576023350569471e+00 + z * (-6.88283971605453293030e-01 + z * 7.70381505559019352791e-02)));
    return p / q;
}
static double d_acos(double x)
{
    const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17, pi = 3.14159265358979311600e+00;
    if (x != x) return x;
    if (fabs(x) >= 1.0)
    {
        if (x == 1.0) return 0.0;
        if (x == -1.0) return pi + 2.0 * pio2_lo;
        return (x - x) / (x - x); /* |x| > 1: NaN */
    }
    if (fabs(x) < 0.5)
    {
        if (fabs(x) <= 6.938893903907228e-18) return pio2_hi + pio2_lo; /* 2^-57 */
        const double r = d_acos_ratio(x * x);
        return pio2_hi - (x - (pio2_lo - x * r));
    }
    if (x < 0.0)
    {
        const double z = (1.0 + x) * 0.5, s = sqrt(z), r = d_acos_ratio(z), w = r * s - pio2_lo;
        return pi - 2.0 * (s + w);
    }
    const double z = (1.0 - x) * 0.5, s = sqrt(z);
    uint64_t bits; memcpy(&bits, &s, 8); bits &= 0xFFFFFFFF00000000ull;
    double df; memcpy(&df, &bits, 8);
    const double c = (z - df * df) / (s + df), r = d_acos_ratio(z), w = r * s + c;
    return 2.0 * (df + w);
}
ORC_API double orc_d_exp(double x) { return d_exp(x); }
ORC_API double orc_d_sin(double x) { return d_sin(x); }
ORC_API double orc_d_cos(double x) { return d_cos(x); }
ORC_API double orc_d_acos(double x) { return d_acos(x); }
static inline float W_SIN(float x) { return (float)d_sin((double)x); }
static inline float W_COS(float x) { return (float)d_cos((double)x); }
static inline float W_ACOS(float x) { return (float)d_acos((double)x); }
static inline float W_EXP(float x) { return (float)d_exp((double)x); }
static inline float W_POW(float x, float y) { return (float)pow((double)x, (double)y); }
static inline float W_POW15(float x) { const double xd = (double)x; return (float)(xd * sqrt(xd)); } /* pow(x, 1.5f): within 1.5 ulp(f64) */
static inline float W_FRACT(float x) { return x - floorf(x); } /* WGSL fract = e - floor(e) */

static const float W_PI = 3.1415927f;       /* wgsl:68 */
static const float W_FRAC_1_PI = 0.31830987f; /* wgsl:69 */
static const float W_T_MAX = 10000.0f;      /* wgsl:73 */

static inline float f32_from_bits(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
/* wgsl:79-83 const-evaluated in f32: radius 0x3B91D640, cos 0x3F7FFF5A, invPdf 0x38826048 */
#define SOLAR_COS_THETA_MAX f32_from_bits(0x3F7FFF5Au)
#define SOLAR_INV_PDF f32_from_bits(0x38826048u)

typedef struct { uint32_t width, height, offset; } OrcTextureDescriptor; /* reference_path_tracer.cpp:211-214 */

typedef struct {
    const OrcBvhNode*           nodes;
    const OrcPositionAttribute* positions;
    const OrcVertexAttributes*  attrs;
    const OrcTextureDescriptor* texDescs;
    const uint32_t*             texels;
    uint64_t                    numTexels;
    const uint8_t*              blueNoise; /* 128*128*2 bytes */
} OrcScene;

typedef struct {
    uint32_t width, height;
    float    camera[19];
    uint32_t numSamplesPerPixel, numBounces;
    float    exposure;
    float    sky[40]; /* AlignedSkyState */
} OrcRenderParams;

typedef struct {
    uint64_t closestRays, shadowRays, closestNodeVisits, shadowNodeVisits, closestTriTests, shadowTriTests;
    uint64_t texelOobClamps, nanPixels;
    uint32_t stackHigh;
} OrcStats;

static void pixar_onb(v3 n, v3* u, v3* v)
{ /* wgsl:309-319 */
    const float s = (n.z >= 0.0f) ? 1.0f : -1.0f;
    const float a = -1.0f / (s + n.z);
    const float b = n.x * n.y * a;
    *u = V3(1.0f + s * n.x * n.x * a, s * b, -s * n.x);
    *v = V3(b, s + n.y * n.y * a, -n.y);
}
static inline v3 mat3_mul(v3 c0, v3 c1, v3 c2, v3 v)
{ /* mat3x3(c0,c1,c2) * v = (c0*v.x + c1*v.y) + c2*v.z */
    return v3_add(v3_add(v3_scale(v.x, c0), v3_scale(v.y, c1)), v3_scale(v.z, c2));
}
static v3 direction_in_cone(float ux, float uy, float cosThetaMax)
{ /* wgsl:568-579 */
    const float cosTheta = 1.0f - ux * (1.0f - cosThetaMax);
    const float sinTheta = sqrtf(1.0f - cosTheta * cosTheta);
    const float phi = 2.0f * W_PI * uy;
    return V3(W_COS(phi) * sinTheta, W_SIN(phi) * sinTheta, cosTheta);
}
static v3 direction_in_cosine_weighted_hemisphere(float ux, float uy)
{ /* wgsl:582-592 */
    const float phi = 2.0f * W_PI * uy;
    const float sinTheta = sqrtf(1.0f - ux);
    return V3(W_COS(phi) * sinTheta, W_SIN(phi) * sinTheta, sqrtf(ux));
}

static float sky_radiance(const float* sky, float theta, float gamma, uint32_t channel)
{ /* wgsl:247-275 (no solar disk term) */
    const float r = sky[27 + channel];
    const float* p = sky + 9 * channel;
    const float cosGamma = W_COS(gamma);
    const float cosGamma2 = cosGamma * cosGamma;
    const float cosTheta = fabsf(W_COS(theta));
    const float expM = W_EXP(p[4] * gamma);
    const float rayM = cosGamma2;
    const float mieMLhs = 1.0f + cosGamma2;
    const float mieMRhs = W_POW15(1.0f + p[8] * p[8] - 2.0f * p[8] * cosGamma);
    const float mieM = mieMLhs / mieMRhs;
    const float zenith = sqrtf(cosTheta);
    const float radianceLhs = 1.0f + p[0] * W_EXP(p[1] / (cosTheta + 0.01f));
    const float radianceRhs = p[2] + p[3] * expM + p[5] * rayM + p[6] * mieM + p[7] * zenith;
    const float radianceDist = radianceLhs * radianceRhs;
    return r * radianceDist;
}

/* exported for tests: wgsl skyRadiance */
ORC_API float orc_wgsl_sky_radiance(const float* sky40, float theta, float gamma, uint32_t channel)
{
    return sky_radiance(sky40, theta, gamma, channel);
}

static v3 texture_lookup(const OrcScene* sc, uint32_t descIdx, float uvx, float uvy, OrcStats* st)
{ /* wgsl:303-307, 552-565.  Out-of-range texel index (fract()*w rounding up to w on the last
   * row): WGSL robust buffer access clamps the index into the array -- documented choice. */
    const OrcTextureDescriptor d = sc->texDescs[descIdx];
    const float u = W_FRACT(uvx), v = W_FRACT(uvy);
    const uint32_t j = (uint32_t)(u * (float)d.width);
    const uint32_t i = (uint32_t)(v * (float)d.height);
    uint64_t idx = (uint64_t)d.offset + (uint64_t)(i * d.width + j);
    if (idx >= sc->numTexels) { idx = sc->numTexels - 1; if (st) st->texelOobClamps++; }
    const uint32_t bgra = sc->texels[idx];
    const v3 srgb = V3((float)((bgra >> 16) & 0xffu) / 255.0f, (float)((bgra >> 8) & 0xffu) / 255.0f, (float)(bgra & 0xffu) / 255.0f);
    return V3(W_POW(srgb.x, 2.2f), W_POW(srgb.y, 2.2f), W_POW(srgb.z, 2.2f));
}

ORC_API void orc_texture_lookup(const OrcScene* sc, uint32_t descIdx, float uvx, float uvy, float* rgb)
{
    v3 c = texture_lookup(sc, descIdx, uvx, uvy, NULL);
    rgb[0] = c.x; rgb[1] = c.y; rgb[2] = c.z;
}

static void animated_blue_noise(const OrcScene* sc, uint32_t x, uint32_t y, uint32_t frameIdx, uint32_t totalSampleCount, float* ox, float* oy)
{ /* wgsl:602-616; table conversion u8/255.0f reference_path_tracer.cpp:174-178 */
    const uint32_t idx = (y % 128u) * 128u + (x % 128u);
    const float bx = (float)sc->blueNoise[2 * idx] / 255.0f, by = (float)sc->blueNoise[2 * idx + 1] / 255.0f;
    const uint32_t n = frameIdx % totalSampleCount;
    const float a1 = 0.7548776662466927f, a2 = 0.5698402909980532f;
    const float r2x = W_FRACT(a1 * (float)n), r2y = W_FRACT(a2 * (float)n);
    *ox = W_FRACT(bx + r2x); *oy = W_FRACT(by + r2y);
}

ORC_API void orc_animated_blue_noise(const uint8_t* table, uint32_t x, uint32_t y, uint32_t frameIdx, uint32_t spp, float* out2)
{
    OrcScene sc; memset(&sc, 0, sizeof sc); sc.blueNoise = table;
    animated_blue_noise(&sc, x, y, frameIdx, spp, &out2[0], &out2[1]);
}

static void generate_camera_ray_lens(const OrcCamera* c, float nx, float ny, float u, float v, v3* ro, v3* rd)
{ /* wgsl:236-245, pointInUnitDisk :594-600 */
    const float r = sqrtf(nx);
    const float theta = 2.0f * W_PI * ny;
    const float px = c->lensRadius * (r * W_COS(theta)), py = c->lensRadius * (r * W_SIN(theta));
    const v3 lensOffset = v3_add(v3_scale(px, c->right), v3_scale(py, c->up));
    const v3 origin = v3_add(c->origin, lensOffset);
    const v3 d = v3_sub(v3_add(v3_add(c->lowerLeftCorner, v3_scale(u, c->horizontal)), v3_scale(v, c->vertical)), origin);
    *ro = origin; *rd = v3_normalize(d);
}

ORC_API void orc_wgsl_camera_ray(const OrcRenderParams* rp, const uint8_t* blueNoise, uint32_t x, uint32_t y, uint32_t frameIdx, float* ray6)
{ /* wgsl:36-54 for pixel (x,y): texCoord at the fragment centre */
    OrcScene sc; memset(&sc, 0, sizeof sc); sc.blueNoise = blueNoise;
    OrcCamera c; memcpy(&c, rp->camera, sizeof c);
    const float u = ((float)x + 0.5f) / (float)rp->width, v = ((float)y + 0.5f) / (float)rp->height;
    float nx, ny; animated_blue_noise(&sc, x, y, frameIdx, rp->numSamplesPerPixel, &nx, &ny);
    const float jx = nx / (float)rp->width, jy = ny / (float)rp->height;
    v3 ro, rd; generate_camera_ray_lens(&c, nx, ny, u + jx, (1.0f - v) + jy, &ro, &rd);
    ray6[0] = ro.x; ray6[1] = ro.y; ray6[2] = ro.z; ray6[3] = rd.x; ray6[4] = rd.y; ray6[5] = rd.z;
}

static v3 ray_color(const OrcScene* sc, const OrcRenderParams* rp, float nx, float ny, v3 ro, v3 rd, OrcStats* st)
{ /* wgsl:180-234 */
    v3 radiance = v3s(0.0f), throughput = v3s(1.0f);
    uint32_t bounce = 1;
    const uint32_t numBounces = rp->numBounces;
    const float* sky = rp->sky;
    const v3 sunDirection = V3(sky[36], sky[37], sky[38]);
    for (;;) {
        BvhHit h; memset(&h, 0, sizeof h);
        st->closestRays++;
        const int did = ray_intersect_bvh(ro, rd, sc->nodes, (const float*)sc->positions, 12, W_T_MAX, &h);
        st->closestNodeVisits += h.nodesVisited; st->closestTriTests += h.triTests;
        if (h.stackHigh > st->stackHigh) st->stackHigh = h.stackHigh;
        if (did) {
            /* wgsl:391-400: attributes of the final (closest) accepted triangle */
            const OrcVertexAttributes* va = &sc->attrs[h.triIdx];
            const float b0 = 1.0f - h.u - h.v, b1 = h.u, b2 = h.v; /* wgsl:515 */
            const v3 n = v3_add(v3_add(v3_scale(b0, V3(va->n0[0], va->n0[1], va->n0[2])), v3_scale(b1, V3(va->n1[0], va->n1[1], va->n1[2]))),
                                v3_scale(b2, V3(va->n2[0], va->n2[1], va->n2[2])));
            const float uvx = (b0 * va->uv0[0] + b1 * va->uv1[0]) + b2 * va->uv2[0];
            const float uvy = (b0 * va->uv0[1] + b1 * va->uv1[1]) + b2 * va->uv2[1];
            const v3 albedo = texture_lookup(sc, va->textureIdx, uvx, uvy, st); /* :191 */
            const v3 p = h.p;
            /* :194 sampleSolarDiskDirection :287-292 */
            v3 ou, ov; pixar_onb(sunDirection, &ou, &ov);
            const v3 lightDirection = mat3_mul(ou, ov, sunDirection, direction_in_cone(nx, ny, SOLAR_COS_THETA_MAX));
            const v3 lightIntensity = V3(sky[30], sky[31], sky[32]);
            const v3 brdf = v3_scale(W_FRAC_1_PI, albedo);
            const v3 reflectance = v3_scale(v3_dot(n, lightDirection), brdf);
            st->shadowRays++;
            uint32_t snv = 0, stt = 0;
            const float lightVisibility = shadow_ray(p, lightDirection, sc->nodes, (const float*)sc->positions, 12, W_T_MAX, &snv, &stt);
            st->shadowNodeVisits += snv; st->shadowTriTests += stt;
            /* :203 radiance += throughput * lightIntensity * reflectance * lightVisibility * SOLAR_INV_PDF */
            radiance = v3_add(radiance, v3_scale(SOLAR_INV_PDF, v3_scale(lightVisibility, v3_mul(v3_mul(throughput, lightIntensity), reflectance))));
            if (bounce == numBounces) break;
            /* :209 evalImplicitLambertian :294-301 */
            v3 nu, nv; pixar_onb(n, &nu, &nv);
            const v3 wi = mat3_mul(nu, nv, n, direction_in_cosine_weighted_hemisphere(nx, ny));
            ro = p; rd = wi;
            throughput = v3_mul(throughput, albedo);
        } else {
            const v3 v = rd, s = sunDirection;
            const float theta = W_ACOS(v.y);
            float dvs = v3_dot(v, s);
            dvs = fmin_glm(fmax_glm(dvs, -1.0f), 1.0f); /* clamp(e,low,high) = min(max(e,low),high) */
            const float gamma = W_ACOS(dvs);
            const v3 skyRad = V3(sky_radiance(sky, theta, gamma, 0), sky_radiance(sky, theta, gamma, 1), sky_radiance(sky, theta, gamma, 2));
            radiance = v3_add(radiance, v3_mul(throughput, skyRad));
            break;
        }
        bounce += 1;
    }
    return radiance;
}

/* One sample of one pixel (wgsl:42-55 without the accumulate). */
ORC_API void orc_pixel_sample(const OrcScene* sc, const OrcRenderParams* rp, uint32_t x, uint32_t y, uint32_t frameCount,
                              float* rgb_out, OrcStats* st)
{
    OrcStats local; memset(&local, 0, sizeof local);
    if (!st) st = &local;
    OrcCamera c; memcpy(&c, rp->camera, sizeof c);
    const float u = ((float)x + 0.5f) / (float)rp->width, v = ((float)y + 0.5f) / (float)rp->height;
    float nx, ny; animated_blue_noise(sc, x, y, frameCount, rp->numSamplesPerPixel, &nx, &ny);
    const float jx = nx / (float)rp->width, jy = ny / (float)rp->height;
    v3 ro, rd; generate_camera_ray_lens(&c, nx, ny, u + jx, (1.0f - v) + jy, &ro, &rd);
    const v3 col = ray_color(sc, rp, nx, ny, ro, rd, st);
    rgb_out[0] = col.x; rgb_out[1] = col.y; rgb_out[2] = col.z;
}

/* Host stepping of reference_path_tracer.cpp:565-595 + fsMain accumulate wgsl:45-57 for a fresh
 * renderer: `numFrames` calls of render(); frame f uses frameCount = firstFrame + f and
 * accumulatedSampleCount = min(firstFrame + f, spp).  image = W*H*4 floats (array<vec3f>, 16-B
 * stride, wgsl:32); only pixels in [x0,x1) x [y0,y1) are touched. */
/* Host-side state stepping of reference_path_tracer.cpp:556-595: frameCount only ever grows, the accumulated sample
 * count restarts at 0 whenever setRenderParameters() sees a change.  orc_render_from() steps numFrames render() calls
 * from (frameCount = firstFrame, accumulatedSampleCount = accumulatedStart); a fresh renderer has both equal. */
ORC_API void orc_render_from(const OrcScene* sc, const OrcRenderParams* rp, uint32_t firstFrame, uint32_t accumulatedStart, uint32_t numFrames,
                             uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, float* image, OrcStats* st)
{
    OrcStats local; memset(&local, 0, sizeof local);
    if (!st) st = &local;
    for (uint32_t f = 0; f < numFrames; ++f) {
        const uint32_t frameCount = firstFrame + f;
        const uint32_t accNow = accumulatedStart + f;
        const uint32_t acc = accNow < rp->numSamplesPerPixel ? accNow : rp->numSamplesPerPixel;
        for (uint32_t y = y0; y < y1; ++y) {
            for (uint32_t x = x0; x < x1; ++x) {
                float* px = image + 4 * ((size_t)y * rp->width + x);
                if (acc == 0) { px[0] = px[1] = px[2] = 0.0f; }
                if (acc < rp->numSamplesPerPixel) {
                    float rgb[3];
                    orc_pixel_sample(sc, rp, x, y, frameCount, rgb, st);
                    px[0] += rgb[0]; px[1] += rgb[1]; px[2] += rgb[2];
                }
            }
        }
    }
    for (uint32_t y = y0; y < y1; ++y)
        for (uint32_t x = x0; x < x1; ++x) {
            const float* px = image + 4 * ((size_t)y * rp->width + x);
            if (px[0] != px[0] || px[1] != px[1] || px[2] != px[2]) st->nanPixels++;
        }
}

ORC_API void orc_render(const OrcScene* sc, const OrcRenderParams* rp, uint32_t firstFrame, uint32_t numFrames,
                        uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, float* image, OrcStats* st)
{
    orc_render_from(sc, rp, firstFrame, firstFrame, numFrames, x0, y0, x1, y1, image, st);
}

/* ------------------------------------------------------------------------------------------ */
/* All-cores drivers for bench.py's cpu_baseline ("row-parallel threads, static   */
/* scanline blocks").  The reference has no threading anywhere (src/common, bvh-visualizer): these */
/* only run the single-threaded loops above on disjoint rows.  Block b of `rowsPerBlock` rows      */
/* belongs to thread b % numThreads (a static deal; interleaved so that sky rows and interior rows */
/* spread over the threads).  Pixels are independent, so the image is the serial one bit for bit.  */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    const OrcScene* sc; const OrcRenderParams* rp;
    uint32_t firstFrame, accumulatedStart, numFrames, x0, y0, x1, y1, rowsPerBlock, thread, numThreads;
    float* image; OrcStats st;
    /* bvh-visualizer variant */
    const OrcBvhNode* nodes; const float* tris; int triStride; const float* cam19; int W, H;
    uint32_t* nodesVisited;
} __attribute__((aligned(128))) OrcThreadJob; /* one thread's counters never share a cache line with another's */

static void* render_thread_main(void* arg)
{
    OrcThreadJob* j = (OrcThreadJob*)arg;
    uint32_t b = j->thread;
    for (;; b += j->numThreads) {
        const uint64_t r0 = (uint64_t)j->y0 + (uint64_t)b * j->rowsPerBlock;
        if (r0 >= j->y1) break;
        const uint32_t r1 = (uint32_t)(r0 + j->rowsPerBlock < j->y1 ? r0 + j->rowsPerBlock : j->y1);
        orc_render_from(j->sc, j->rp, j->firstFrame, j->accumulatedStart, j->numFrames, j->x0, (uint32_t)r0, j->x1, r1, j->image, &j->st);
    }
    return NULL;
}

static void* visualize_thread_main(void* arg)
{
    OrcThreadJob* j = (OrcThreadJob*)arg;
    uint32_t b = j->thread;
    for (;; b += j->numThreads) {
        const uint64_t r0 = (uint64_t)j->y0 + (uint64_t)b * j->rowsPerBlock;
        if (r0 >= j->y1) break;
        const uint32_t r1 = (uint32_t)(r0 + j->rowsPerBlock < j->y1 ? r0 + j->rowsPerBlock : j->y1);
        orc_bvh_visualize(j->nodes, j->tris, j->triStride, j->cam19, j->W, j->H, (int)r0, (int)r1, j->nodesVisited, NULL, NULL, NULL, NULL);
    }
    return NULL;
}

static int run_jobs(OrcThreadJob* jobs, uint32_t n, void* (*fn)(void*))
{
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * n);
    uint32_t started = 0;
    if (!th) return -1;
    for (; started < n; ++started)
        if (pthread_create(&th[started], NULL, fn, &jobs[started]) != 0) break;
    for (uint32_t i = started; i < n; ++i) fn(&jobs[i]); /* could not start that many threads: run the rest here */
    for (uint32_t i = 0; i < started; ++i) pthread_join(th[i], NULL);
    free(th);
    return (int)started;
}

/* -> number of threads actually started (the calling thread runs whatever could not be started) */
ORC_API int orc_render_threads(const OrcScene* sc, const OrcRenderParams* rp, uint32_t firstFrame, uint32_t accumulatedStart, uint32_t numFrames,
                               uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1, float* image, OrcStats* st, uint32_t numThreads, uint32_t rowsPerBlock)
{
    if (numThreads == 0) numThreads = 1;
    if (rowsPerBlock == 0) rowsPerBlock = 1;
    OrcThreadJob* jobs = (OrcThreadJob*)aligned_alloc(128, (size_t)numThreads * sizeof(OrcThreadJob));
    if (jobs) memset(jobs, 0, (size_t)numThreads * sizeof(OrcThreadJob));
    if (!jobs) return -1;
    for (uint32_t t = 0; t < numThreads; ++t) {
        OrcThreadJob* j = &jobs[t];
        j->sc = sc; j->rp = rp; j->firstFrame = firstFrame; j->accumulatedStart = accumulatedStart; j->numFrames = numFrames;
        j->x0 = x0; j->y0 = y0; j->x1 = x1; j->y1 = y1; j->rowsPerBlock = rowsPerBlock; j->thread = t; j->numThreads = numThreads; j->image = image;
    }
    const int started = run_jobs(jobs, numThreads, render_thread_main);
    if (st) {
        for (uint32_t t = 0; t < numThreads; ++t) {
            const OrcStats* a = &jobs[t].st;
            st->closestRays += a->closestRays; st->shadowRays += a->shadowRays;
            st->closestNodeVisits += a->closestNodeVisits; st->shadowNodeVisits += a->shadowNodeVisits;
            st->closestTriTests += a->closestTriTests; st->shadowTriTests += a->shadowTriTests;
            st->texelOobClamps += a->texelOobClamps; st->nanPixels += a->nanPixels;
            if (a->stackHigh > st->stackHigh) st->stackHigh = a->stackHigh;
        }
    }
    free(jobs);
    return started;
}

ORC_API int orc_bvh_visualize_threads(const OrcBvhNode* nodes, const float* tris, int tri_stride_floats, const float* cam19, int W, int H,
                                      uint32_t* nodes_visited_out, uint32_t numThreads, uint32_t rowsPerBlock)
{
    if (numThreads == 0) numThreads = 1;
    if (rowsPerBlock == 0) rowsPerBlock = 1;
    OrcThreadJob* jobs = (OrcThreadJob*)aligned_alloc(128, (size_t)numThreads * sizeof(OrcThreadJob));
    if (jobs) memset(jobs, 0, (size_t)numThreads * sizeof(OrcThreadJob));
    if (!jobs) return -1;
    for (uint32_t t = 0; t < numThreads; ++t) {
        OrcThreadJob* j = &jobs[t];
        j->nodes = nodes; j->tris = tris; j->triStride = tri_stride_floats; j->cam19 = cam19; j->W = W; j->H = H; j->nodesVisited = nodes_visited_out;
        j->y0 = 0; j->y1 = (uint32_t)H; j->rowsPerBlock = rowsPerBlock; j->thread = t; j->numThreads = numThreads;
    }
    const int started = run_jobs(jobs, numThreads, visualize_thread_main);
    free(jobs);
    return started;
}

/* ------------------------------------------------------------------------------------------ */
/* Deferred-lighting variant  (pt/deferred_renderer_lighting_pass.wgsl + _resolve_pass.wgsl)    */
/* SURVEY.md 8(f) row 4.  The reference fills its G-buffer with a hardware rasteriser            */
/* (deferred_renderer_gbuffer_pass.wgsl); here the G-buffer comes from one primary ray per pixel  */
/* through the jittered pixel centre -- everything downstream restates the WGSL line by line.    */
/* ------------------------------------------------------------------------------------------ */
static v3 offset_position_deferred(v3 p, v3 n)
{ /* lighting_pass.wgsl:498-518: the same construction as offset_ray with FLOAT_SCALE 1/16384 and INT_SCALE 1024 */
    const float ORIGIN = 1.0f / 32.0f, FLOAT_SCALE = 1.0f / 16384.0f, INT_SCALE = 1024.0f;
    const int32_t ox = (int32_t)(INT_SCALE * n.x), oy = (int32_t)(INT_SCALE * n.y), oz = (int32_t)(INT_SCALE * n.z);
    int32_t ix, iy, iz;
    memcpy(&ix, &p.x, 4); memcpy(&iy, &p.y, 4); memcpy(&iz, &p.z, 4);
    ix = (int32_t)((uint32_t)ix + (uint32_t)(p.x < 0 ? -ox : ox));
    iy = (int32_t)((uint32_t)iy + (uint32_t)(p.y < 0 ? -oy : oy));
    iz = (int32_t)((uint32_t)iz + (uint32_t)(p.z < 0 ? -oz : oz));
    v3 po; memcpy(&po.x, &ix, 4); memcpy(&po.y, &iy, 4); memcpy(&po.z, &iz, 4);
    return V3(fabsf(p.x) < ORIGIN ? p.x + FLOAT_SCALE * n.x : po.x,
              fabsf(p.y) < ORIGIN ? p.y + FLOAT_SCALE * n.y : po.y,
              fabsf(p.z) < ORIGIN ? p.z + FLOAT_SCALE * n.z : po.z);
}

static float sky_radiance_with_sun(const float* sky, float theta, float gamma, uint32_t channel)
{ /* lighting_pass.wgsl:200-238: the dome of wgsl:247-275 plus the solar disk (:231-235); TERRESTRIAL_SOLAR_RADIUS
   * = 0.255f * (PI / 180f) const-evaluated in f32 = 0x3B91D640 */
    const float radius = gamma / f32_from_bits(0x3B91D640u);
    const float solar = radius <= 1.0f ? sky[30 + channel] : 0.0f;
    return sky_radiance(sky, theta, gamma, channel) + solar;
}

static void r2_sequence_host(uint32_t n, uint32_t sequenceLength, float* x, float* y)
{ /* common/r_sequence.hpp:9-21 with common/math.hpp:7-17's sign-preserving fract (arguments are >= 0 here) */
    const float G = 1.32471795f;
    const float A1 = 1.0f / G, A2 = 1.0f / (G * G);
    const float i = (float)(n % sequenceLength);
    const float a = 0.5f + A1 * i, b = 0.5f + A2 * i;
    *x = a - floorf(a); *y = b - floorf(b);
}

typedef struct { int hit; v3 position, normal, albedo, direction; } GTexel;

/* Interpolated attributes of a hit (lighting_pass.wgsl:312-321) + the hit point pushed off the surface along the
 * geometric normal with the deferred constants (:447-450). */
static void deferred_hit_attributes(const OrcScene* sc, const BvhHit* h, v3* pOffset, v3* pPlain, v3* n, v3* albedo, OrcStats* st)
{
    const OrcPositionAttribute* t = &sc->positions[h->triIdx];
    const v3 p0 = V3(t->p0[0], t->p0[1], t->p0[2]), p1 = V3(t->p1[0], t->p1[1], t->p1[2]), p2 = V3(t->p2[0], t->p2[1], t->p2[2]);
    const v3 e1 = v3_sub(p1, p0), e2 = v3_sub(p2, p0);
    const v3 p = v3_add(v3_add(p0, v3_scale(h->u, e1)), v3_scale(h->v, e2));
    *pPlain = p;
    *pOffset = offset_position_deferred(p, v3_normalize(v3_cross(e1, e2)));
    const OrcVertexAttributes* va = &sc->attrs[h->triIdx];
    const float b0 = 1.0f - h->u - h->v, b1 = h->u, b2 = h->v;
    *n = v3_add(v3_add(v3_scale(b0, V3(va->n0[0], va->n0[1], va->n0[2])), v3_scale(b1, V3(va->n1[0], va->n1[1], va->n1[2]))),
                v3_scale(b2, V3(va->n2[0], va->n2[1], va->n2[2])));
    const float uvx = (b0 * va->uv0[0] + b1 * va->uv1[0]) + b2 * va->uv2[0];
    const float uvy = (b0 * va->uv0[1] + b1 * va->uv1[1]) + b2 * va->uv2[1];
    *albedo = texture_lookup(sc, va->textureIdx, uvx, uvy, st);
}

static v3 deferred_light_sample(const OrcScene* sc, const float* sky, float ux, float uy, v3 position, v3 normal, v3 albedo, OrcStats* st)
{ /* lighting_pass.wgsl:188-198 */
    const v3 sunDirection = V3(sky[36], sky[37], sky[38]);
    v3 ou, ov; pixar_onb(sunDirection, &ou, &ov);
    const v3 lightDirection = mat3_mul(ou, ov, sunDirection, direction_in_cone(ux, uy, SOLAR_COS_THETA_MAX));
    const v3 lightIntensity = V3(sky[30], sky[31], sky[32]);
    const v3 brdf = v3_scale(W_FRAC_1_PI, albedo);
    const v3 reflectance = v3_scale(v3_dot(normal, lightDirection), brdf);
    st->shadowRays++;
    uint32_t snv = 0, stt = 0;
    const float vis = shadow_ray(position, lightDirection, sc->nodes, (const float*)sc->positions, 12, W_T_MAX, &snv, &stt);
    st->shadowNodeVisits += snv; st->shadowTriTests += stt;
    return v3_scale(SOLAR_INV_PDF, v3_scale(vis, v3_mul(lightIntensity, reflectance)));
}

/* One frame of the deferred renderer for the pixels [x0,x1) x [y0,y1): G-buffer from a primary ray through the pixel
 * centre displaced by the frame's projection jitter (deferred_renderer.cpp:309-315: NDC offset (r2 - 0.5) / size =
 * (r2 - 0.5) / 2 pixels), lighting pass (lighting_pass.wgsl:96-186), resolve (resolve_pass.wgsl:33-54).
 * sampleBuffer / accumulationBuffer: W*H*3 floats (array<array<f32, 3>>).  srgb_out3 (NULL ok): the resolve pass's
 * return value, acesFilmic(exposure * colour) ^ (1/2.2). */
ORC_API void orc_deferred_frame(const OrcScene* sc, const OrcRenderParams* rp, uint32_t frameCount, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                                float* sampleBuffer, float* accumulationBuffer, float* srgb_out3, OrcStats* st)
{
    OrcStats local; memset(&local, 0, sizeof local);
    if (!st) st = &local;
    OrcCamera c; memcpy(&c, rp->camera, sizeof c);
    const float* sky = rp->sky;
    const v3 sunDirection = V3(sky[36], sky[37], sky[38]);
    const float W = (float)rp->width, H = (float)rp->height;
    float jx, jy; r2_sequence_host(frameCount, 1u << 20, &jx, &jy);
    for (uint32_t y = y0; y < y1; ++y) {
        for (uint32_t x = x0; x < x1; ++x) {
            const float su = ((float)x + 0.5f) / W - (jx - 0.5f) / (2.0f * W);
            const float tv = (1.0f - ((float)y + 0.5f) / H) - (jy - 0.5f) / (2.0f * H);
            v3 ro, rd; generate_camera_ray_pinhole(&c, su, tv, &ro, &rd);
            BvhHit h; memset(&h, 0, sizeof h);
            st->closestRays++;
            v3 color;
            if (!ray_intersect_bvh(ro, rd, sc->nodes, (const float*)sc->positions, 12, W_T_MAX, &h)) {
                /* :106-117 (depth == 0: nothing rasterised) */
                float dvs = v3_dot(rd, sunDirection);
                dvs = fmin_glm(fmax_glm(dvs, -1.0f), 1.0f);
                const float theta = W_ACOS(rd.y), gamma = W_ACOS(dvs);
                color = V3(sky_radiance_with_sun(sky, theta, gamma, 0), sky_radiance_with_sun(sky, theta, gamma, 1), sky_radiance_with_sun(sky, theta, gamma, 2));
            } else {
                v3 pOff, pPlain, normal, albedo;
                deferred_hit_attributes(sc, &h, &pOff, &pPlain, &normal, &albedo, st);
                /* :118-125: the G-buffer position is the surface point; it is offset along the SHADING normal */
                v3 position = offset_position_deferred(pPlain, normal);
                /* surfaceColor :142-186, NUM_BOUNCES = 2 */
                v3 radiance = v3s(0.0f), throughput = v3s(1.0f);
                float ux, uy; animated_blue_noise(sc, x, y, frameCount, 1u << 20, &ux, &uy);
                radiance = v3_add(radiance, v3_mul(throughput, deferred_light_sample(sc, sky, ux, uy, position, normal, albedo, st)));
                for (int bounce = 1; bounce < 2; ++bounce) {
                    v3 nu, nv; pixar_onb(normal, &nu, &nv);
                    const v3 wi = mat3_mul(nu, nv, normal, direction_in_cosine_weighted_hemisphere(ux, uy));
                    throughput = v3_mul(throughput, albedo);
                    BvhHit h2; memset(&h2, 0, sizeof h2);
                    st->closestRays++;
                    if (ray_intersect_bvh(position, wi, sc->nodes, (const float*)sc->positions, 12, W_T_MAX, &h2)) {
                        v3 unused;
                        deferred_hit_attributes(sc, &h2, &position, &unused, &normal, &albedo, st);
                    } else {
                        float dvs = v3_dot(wi, sunDirection);
                        dvs = fmin_glm(fmax_glm(dvs, -1.0f), 1.0f);
                        const float theta = W_ACOS(wi.y), gamma = W_ACOS(dvs);
                        const v3 skyRad = V3(sky_radiance_with_sun(sky, theta, gamma, 0), sky_radiance_with_sun(sky, theta, gamma, 1), sky_radiance_with_sun(sky, theta, gamma, 2));
                        radiance = v3_add(radiance, v3_mul(throughput, skyRad));
                        break;
                    }
                    radiance = v3_add(radiance, v3_mul(throughput, deferred_light_sample(sc, sky, ux, uy, position, normal, albedo, st)));
                }
                color = radiance;
            }
            const size_t idx = (size_t)y * rp->width + x;
            sampleBuffer[3 * idx] = color.x; sampleBuffer[3 * idx + 1] = color.y; sampleBuffer[3 * idx + 2] = color.z;
            /* resolve_pass.wgsl:38-52 */
            v3 outc = color;
            if (frameCount != 0) {
                const v3 prev = V3(accumulationBuffer[3 * idx], accumulationBuffer[3 * idx + 1], accumulationBuffer[3 * idx + 2]);
                outc = v3_add(v3_scale(0.1f, color), v3_scale(0.9f, prev));
            }
            accumulationBuffer[3 * idx] = outc.x; accumulationBuffer[3 * idx + 1] = outc.y; accumulationBuffer[3 * idx + 2] = outc.z;
            if (srgb_out3) {
                const float in[3] = {outc.x, outc.y, outc.z};
                for (int ch = 0; ch < 3; ++ch) {
                    const float xx = rp->exposure * in[ch];
                    const float a = 2.51f, b = 0.03f, cc = 2.43f, d = 0.59f, e = 0.14f;
                    float yy = (xx * (a * xx + b)) / (xx * (cc * xx + d) + e);
                    yy = fmin_glm(fmax_glm(yy, 0.0f), 1.0f);
                    srgb_out3[3 * idx + ch] = W_POW(yy, 1.0f / 2.2f);
                }
            }
        }
    }
}

/* fsMain tonemap tail, wgsl:59-63 + acesFilmic :277-285.  Returns BGRA8 packed like the swap
 * chain (BGRA8Unorm): round-to-nearest of srgb*255. */
ORC_API void orc_tonemap(const float* image, uint32_t numPixels, uint32_t accumulatedSampleCount, float exposure, float* srgb_out3)
{
    for (uint32_t i = 0; i < numPixels; ++i) {
        for (int ch = 0; ch < 3; ++ch) {
            const float est = image[4 * i + ch] / (float)accumulatedSampleCount;
            const float x = exposure * est;
            const float a = 2.51f, b = 0.03f, c = 2.43f, d = 0.59f, e = 0.14f;
            float y = (x * (a * x + b)) / (x * (c * x + d) + e);
            y = fmin_glm(fmax_glm(y, 0.0f), 1.0f);
            srgb_out3[3 * i + ch] = W_POW(y, 1.0f / 2.2f);
        }
    }
}

/* exported for tests: the sampling helpers of the integrator, one call each */
ORC_API void orc_pixar_onb(const float* n3, float* u3, float* v3out)
{
    v3 u, v;
    pixar_onb(V3(n3[0], n3[1], n3[2]), &u, &v);
    u3[0] = u.x; u3[1] = u.y; u3[2] = u.z;
    v3out[0] = v.x; v3out[1] = v.y; v3out[2] = v.z;
}
ORC_API void orc_direction_in_cone(float ux, float uy, float cosThetaMax, float* d3)
{
    const v3 d = direction_in_cone(ux, uy, cosThetaMax);
    d3[0] = d.x; d3[1] = d.y; d3[2] = d.z;
}
ORC_API void orc_direction_in_cosine_weighted_hemisphere(float ux, float uy, float* d3)
{
    const v3 d = direction_in_cosine_weighted_hemisphere(ux, uy);
    d3[0] = d.x; d3[1] = d.y; d3[2] = d.z;
}
ORC_API void orc_offset_ray(const float* p3, const float* n3, float* o3)
{
    const v3 o = offset_ray(V3(p3[0], p3[1], p3[2]), V3(n3[0], n3[1], n3[2]));
    o3[0] = o.x; o3[1] = o.y; o3[2] = o.z;
}

ORC_API uint32_t orc_sizeof_stats(void) { return (uint32_t)sizeof(OrcStats); }
ORC_API uint32_t orc_sizeof_scene(void) { return (uint32_t)sizeof(OrcScene); }
ORC_API uint32_t orc_sizeof_render_params(void) { return (uint32_t)sizeof(OrcRenderParams); }
