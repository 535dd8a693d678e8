// rf_camera.hpp -- thin-lens camera construction (behaviour: src/common/camera.cpp:7-52,
// src/common/units/angle.hpp:12-15, src/pt/fly_camera_controller.{hpp:47-52,cpp:12-22,138-148}).
#pragma once

#include "rf_types.hpp"

namespace rf
{
// Angle::degrees: degrees * pi_f / 180.0f
inline float degreesToRadians(float degrees) { return degrees * 3.14159265358979323846f / 180.0f; }

// aspectRatio = width / height
Camera createCamera(Vec3 origin, Vec3 lookAt, float aperture, float focusDistance, float vfovRadians, float aspectRatio);

// Camera of the interactive app for a fly-camera pose; the reference's defaults are position
// (1.22, 1.25, -1.25), yaw 129.64 deg, pitch -13.73 deg, aperture 0, focus 10, vfov 70 deg (UI).
Camera flyCamera(Vec3 position, float yawDegrees, float pitchDegrees, float vfovDegrees, float aperture,
                 float focusDistance, float aspectRatio);

// Camera used by the bvh-visualizer tool (src/bvh-visualizer/main.cpp:36-55) for a BVH root box.
Camera bvhVisualizerCamera(const Aabb& rootAabb, float aspectRatio);
} // namespace rf
