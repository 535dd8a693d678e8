#include "rf_camera.hpp"

#include "rf_aabb.hpp"

#include <cmath>

namespace rf
{
Camera createCamera(Vec3 origin, Vec3 lookAt, float aperture, float focusDistance, float vfovRadians, float aspectRatio)
{
    const float halfHeight = focusDistance * std::tan(0.5f * vfovRadians);
    const float halfWidth = aspectRatio * halfHeight;

    const Vec3 worldUp = vec3(0.0f, 1.0f, 0.0f);
    const Vec3 forward = normalize(lookAt - origin);
    const Vec3 right = normalize(cross(forward, worldUp));
    const Vec3 up = cross(right, forward);

    Camera c;
    c.origin = origin;
    c.lowerLeftCorner = origin - halfWidth * right - halfHeight * up + focusDistance * forward;
    c.horizontal = 2.0f * halfWidth * right;
    c.vertical = 2.0f * halfHeight * up;
    c.up = up;
    c.right = right;
    c.lensRadius = 0.5f * aperture;
    return c;
}

Camera flyCamera(Vec3 position, float yawDegrees, float pitchDegrees, float vfovDegrees, float aperture,
                 float focusDistance, float aspectRatio)
{
    const float yaw = degreesToRadians(yawDegrees);
    const float pitch = degreesToRadians(pitchDegrees);
    const Vec3  forward = normalize(vec3(std::cos(yaw) * std::cos(pitch), std::sin(pitch), std::sin(yaw) * std::cos(pitch)));
    return createCamera(position, position + focusDistance * forward, aperture, focusDistance, degreesToRadians(vfovDegrees), aspectRatio);
}

Camera bvhVisualizerCamera(const Aabb& rootAabb, float aspectRatio)
{
    const Box   box(rootAabb.min, rootAabb.max);
    const Vec3  diag = diagonal(box);
    const Vec3  center = centroid(box);
    const float extent = diag[maxDimension(box)];
    // The tool writes vec3(-0.8 * d, 0.0f, 0.8f * d): the x offset is a double product rounded to
    // f32, the z offset an f32 product (bvh-visualizer/main.cpp:49).
    const Vec3 offset = vec3(static_cast<float>(-0.8 * static_cast<double>(extent)), 0.0f, 0.8f * extent);
    return createCamera(center - offset, center, 0.0f, 1.0f, degreesToRadians(70.0f), aspectRatio);
}
} // namespace rf
