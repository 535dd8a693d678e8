// rf_math.hpp -- f32 vector arithmetic shared by host C++ and gfx950 device code.
//
// Parity contract: every expression is evaluated exactly as glm 0.9.9.8 (the reference's maths
// library, external/CMakeLists.txt:30-31) associates it, in IEEE f32 without FMA contraction.
// All translation units that include this header are compiled with -ffp-contract=off.
//   dot(a,b)      = (a.x*b.x + a.y*b.y) + a.z*b.z
//   cross(a,b)    = (a.y*b.z - b.y*a.z, a.z*b.x - b.z*a.x, a.x*b.y - b.x*a.y)
//   normalize(v)  = v * (1.0f / sqrt(dot(v,v)))
//   min(a,b)      = (b < a) ? b : a          max(a,b) = (a < b) ? b : a      (NaN-propagation
//                   of the ternaries matters for axis-parallel rays: 0*inf slabs)
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RF_HD __host__ __device__ __forceinline__
#else
#define RF_HD inline
#endif

namespace rf
{
struct Vec3
{
    float x, y, z;

    RF_HD float  operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};

RF_HD Vec3 vec3(float x, float y, float z) { return Vec3{x, y, z}; }
RF_HD Vec3 splat(float s) { return Vec3{s, s, s}; }

RF_HD Vec3 operator+(Vec3 a, Vec3 b) { return Vec3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RF_HD Vec3 operator-(Vec3 a, Vec3 b) { return Vec3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RF_HD Vec3 operator*(Vec3 a, Vec3 b) { return Vec3{a.x * b.x, a.y * b.y, a.z * b.z}; }
RF_HD Vec3 operator*(float s, Vec3 a) { return Vec3{s * a.x, s * a.y, s * a.z}; }
RF_HD Vec3 operator*(Vec3 a, float s) { return Vec3{a.x * s, a.y * s, a.z * s}; }

RF_HD float minf(float a, float b) { return (b < a) ? b : a; }
RF_HD float maxf(float a, float b) { return (a < b) ? b : a; }
RF_HD Vec3  vmin(Vec3 a, Vec3 b) { return Vec3{minf(a.x, b.x), minf(a.y, b.y), minf(a.z, b.z)}; }
RF_HD Vec3  vmax(Vec3 a, Vec3 b) { return Vec3{maxf(a.x, b.x), maxf(a.y, b.y), maxf(a.z, b.z)}; }

RF_HD float dot(Vec3 a, Vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
RF_HD Vec3  cross(Vec3 a, Vec3 b)
{
    return Vec3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
// IEEE correctly rounded on host (sqrtss) and device (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt); checked on the GPU by tests/test_gpu_math.py.
RF_HD float rf_sqrt(float x) { return __builtin_sqrtf(x); }
RF_HD Vec3 normalize(Vec3 v) { return v * (1.0f / rf_sqrt(dot(v, v))); }

RF_HD uint32_t floatBits(float f)
{
    uint32_t u;
    memcpy(&u, &f, sizeof u);
    return u;
}
RF_HD float bitsFloat(uint32_t u)
{
    float f;
    memcpy(&f, &u, sizeof f);
    return f;
}
} // namespace rf
