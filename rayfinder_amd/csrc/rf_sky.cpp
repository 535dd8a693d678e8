#include "rf_sky.hpp"

#include "rf_camera.hpp"
#include "rf_data.hpp"

#include <cmath>
#include <cstddef>
#include <cstring>

namespace rf
{
namespace
{
constexpr float kPi = 3.14159265358979323846f;

// Quintic Bernstein interpolation over the 6 control values data[0], data[stride], ...
float bezier5(const float* data, std::size_t stride, float t)
{
    const float t2 = t * t;
    const float t3 = t2 * t;
    const float t4 = t2 * t2;
    const float t5 = t4 * t;
    const float s = 1.0f - t;
    const float s2 = s * s;
    const float s3 = s2 * s;
    const float s4 = s2 * s2;
    const float s5 = s4 * s;

    const float m0 = data[0] * s5;
    const float m1 = data[stride] * 5.0f * s4 * t;
    const float m2 = data[2 * stride] * 10.0f * s3 * t2;
    const float m3 = data[3 * stride] * 10.0f * s2 * t3;
    const float m4 = data[4 * stride] * 5.0f * s * t4;
    const float m5 = data[5 * stride] * t5;
    return m0 + m1 + m2 + m3 + m4 + m5;
}
} // namespace

SkyResult skyStateNew(float elevation, float turbidity, const float albedo[3], float state[33])
{
    if (elevation < 0.0f || elevation > kPi) return SkyResult::ElevationOutOfRange;
    if (turbidity < 1.0f || turbidity > 10.0f) return SkyResult::TurbidityOutOfRange;
    for (int c = 0; c < 3; ++c)
        if (albedo[c] < 0.0f || albedo[c] > 1.0f) return SkyResult::AlbedoOutOfRange;

    const float       t = std::pow((elevation / (0.5f * kPi)), (1.0f / 3.0f));
    const std::size_t whole = static_cast<std::size_t>(turbidity);
    const float       frac = std::fmod(turbidity, 1.0f);
    const std::size_t lo = whole - 1;
    const std::size_t hi = whole < 9 ? whole : 9;

    const float* const paramTables = hwSkyTables();
    const float* const radianceTables = hwSkyTables() + 3 * 1080;
    const float* const solarTables = hwSkyTables() + 3 * 1080 + 3 * 120;

    for (int c = 0; c < 3; ++c)
    {
        const float a = albedo[c];
        const float w0 = (1.0f - a) * (1.0f - frac);
        const float w1 = (1.0f - a) * frac;
        const float w2 = a * (1.0f - frac);
        const float w3 = a * frac;

        const float* pt = paramTables + 1080 * c;
        const float* q0 = pt + 54 * lo;
        const float* q1 = pt + 54 * hi;
        const float* q2 = pt + (540 + 54 * lo);
        const float* q3 = pt + (540 + 54 * hi);
        for (std::size_t i = 0; i < 9; ++i)
        {
            float v = 0.0f;
            v += w0 * bezier5(q0 + i, 9, t);
            v += w1 * bezier5(q1 + i, 9, t);
            v += w2 * bezier5(q2 + i, 9, t);
            v += w3 * bezier5(q3 + i, 9, t);
            state[9 * c + i] = v;
        }

        const float* rt = radianceTables + 120 * c;
        float        r = 0.0f;
        r += w0 * bezier5(rt + 6 * lo, 1, t);
        r += w1 * bezier5(rt + 6 * hi, 1, t);
        r += w2 * bezier5(rt + (60 + 6 * lo), 1, t);
        r += w3 * bezier5(rt + (60 + 6 * hi), 1, t);
        state[27 + c] = r;

        const float* st = solarTables + 10 * c;
        state[30 + c] = st[lo] * (1.0f - frac) + st[hi] * frac;
    }
    return SkyResult::Success;
}

float skyStateRadiance(const float state[33], float theta, float gamma, int channel)
{
    const float  r = state[27 + channel];
    const float* p = state + 9 * channel;

    const float cosGamma = std::cos(gamma);
    const float cosGamma2 = cosGamma * cosGamma;
    const float cosTheta = std::fabs(std::cos(theta));

    const float expM = std::exp(p[4] * gamma);
    const float mieLhs = 1.0f + cosGamma2;
    const float mieRhs = std::pow(1.0f + p[8] * p[8] - 2.0f * p[8] * cosGamma, 1.5f);
    const float mie = mieLhs / mieRhs;
    const float zenith = std::sqrt(cosTheta);
    const float lhs = 1.0f + p[0] * std::exp(p[1] / (cosTheta + 0.01f));
    const float rhs = p[2] + p[3] * expM + p[5] * cosGamma2 + p[6] * mie + p[7] * zenith;
    const float dist = lhs * rhs;

    const float diskRadius = gamma / 0.004450589f;
    const float solar = diskRadius <= 1.f ? state[30 + channel] : 0.f;
    return r * dist + solar;
}

SkyResult alignedSkyState(const Sky& sky, SkyStateGpu& out)
{
    std::memset(&out, 0, sizeof out);
    const float zenith = degreesToRadians(sky.sunZenithDegrees);
    const float azimuth = degreesToRadians(sky.sunAzimuthDegrees);
    const Vec3  sun = normalize(vec3(std::sin(zenith) * std::cos(azimuth), std::cos(zenith), -std::sin(zenith) * std::sin(azimuth)));
    out.sunDirection[0] = sun.x;
    out.sunDirection[1] = sun.y;
    out.sunDirection[2] = sun.z;

    float           state[33];
    const SkyResult rc = skyStateNew(0.5f * kPi - zenith, sky.turbidity, sky.albedo, state);
    if (rc != SkyResult::Success) return rc;
    std::memcpy(out.params, state, 27 * sizeof(float));
    std::memcpy(out.skyRadiances, state + 27, 3 * sizeof(float));
    std::memcpy(out.solarRadiances, state + 30, 3 * sizeof(float));
    return SkyResult::Success;
}
} // namespace rf
