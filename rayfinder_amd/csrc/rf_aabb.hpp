// rf_aabb.hpp -- axis-aligned box helpers with the reference's exact semantics
// (src/common/aabb.hpp:12-71).  Note the two-point constructor takes the component-wise min AND
// max of both arguments, so merging two empty boxes yields [lowest, max] rather than an empty
// box; the builder relies on never doing that (bucket 0 and bucket 11 are never empty).
#pragma once

#include "rf_types.hpp"

#include <limits>

namespace rf
{
struct Box
{
    Vec3 lo = splat(std::numeric_limits<float>::max());
    Vec3 hi = splat(std::numeric_limits<float>::lowest());

    Box() = default;
    RF_HD Box(Vec3 p1, Vec3 p2) : lo(vmin(p1, p2)), hi(vmax(p1, p2)) {}
};

RF_HD Vec3  centroid(const Box& b) { return 0.5f * (b.lo + b.hi); }
RF_HD Vec3  diagonal(const Box& b) { return b.hi - b.lo; }
RF_HD int   maxDimension(const Box& b)
{
    const Vec3 d = diagonal(b);
    if (d.x > d.y && d.x > d.z) return 0;
    if (d.y > d.z) return 1;
    return 2;
}
RF_HD Box   merge(const Box& b, Vec3 p) { return Box(vmin(b.lo, p), vmax(b.hi, p)); }
RF_HD Box   merge(const Box& a, const Box& b) { return Box(vmin(a.lo, b.lo), vmax(a.hi, b.hi)); }
RF_HD float surfaceArea(const Box& b)
{
    const Vec3 d = diagonal(b);
    return 2.0f * ((d.x * d.y + d.x * d.z) + d.y * d.z);
}
RF_HD Box   boundsOf(const Positions& t)
{
    return Box(vmin(vmin(t.v0, t.v1), t.v2), vmax(vmax(t.v0, t.v1), t.v2));
}
RF_HD Aabb  toAabb(const Box& b) { return Aabb{b.lo, 0.0f, b.hi, 0.0f}; }
} // namespace rf
