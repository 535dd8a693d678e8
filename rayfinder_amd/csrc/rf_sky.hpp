// rf_sky.hpp -- Hosek-Wilkie RGB sky model state (behaviour: src/hw-skymodel/hw_skymodel.c:64-180,
// packing: src/pt/aligned_sky_state.hpp:34-71).  Coefficient tables are data files under
// rayfinder_amd/data, embedded in the library.
#pragma once

#include "rf_types.hpp"

namespace rf
{
struct Sky
{
    float turbidity = 1.0f;
    float albedo[3] = {1.0f, 1.0f, 1.0f};
    float sunZenithDegrees = 30.0f;
    float sunAzimuthDegrees = 0.0f;

    bool operator==(const Sky&) const = default;
};

enum class SkyResult : int
{
    Success = 0,
    ElevationOutOfRange = 1,
    TurbidityOutOfRange = 2,
    AlbedoOutOfRange = 3,
};

// state33 = params[27], sky radiances[3], solar radiances[3]
SkyResult skyStateNew(float elevation, float turbidity, const float albedo[3], float state33[33]);

// CPU evaluation including the solar-disk term (sky_state_radiance, hw_skymodel.c:182-222).
float skyStateRadiance(const float state33[33], float theta, float gamma, int channel);

SkyResult alignedSkyState(const Sky& sky, SkyStateGpu& out);
} // namespace rf
