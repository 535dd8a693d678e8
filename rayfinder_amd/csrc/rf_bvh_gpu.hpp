// rf_bvh_gpu.hpp -- GPU build of the reference's flattened BVH (see rf_bvh_gpu.hip).
#pragma once

#include "rf_bvh.hpp"

namespace rf
{
// Same contract as buildBvh() (src/common/bvh.hpp:33): identical node bytes AND identical triangleIndices
// (the libstdc++ std::partition permutation is reproduced, rf_bvh_gpu.hip).  Throws std::runtime_error without a GPU.
// buildMsOut (optional): device time of the build proper (triangles already uploaded), HIP events.
Bvh buildBvhGpu(std::span<const Positions> triangles, int deviceOrdinal = 0, float* buildMsOut = nullptr);
} // namespace rf
