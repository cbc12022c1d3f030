// Device arithmetic shared by the F(3x3,3x3) input transforms (winograd.hip, roipool_wino.hip): B^T d for one column / row of a
// 5 x 5 patch, interpolation points {0, 1, -1, 2, inf}.  One definition, so that every kernel that forms V = B^T d B rounds
// exactly alike (-ffp-contract=off: the expressions below are the operation order).
#pragma once
#include <hip/hip_runtime.h>

namespace mscnn {

__device__ __forceinline__ void bt5(const float d[5], float r[5]) {
  r[0] = 2.f * d[0] - d[1] - 2.f * d[2] + d[3];
  r[1] = -2.f * d[1] - d[2] + d[3];
  r[2] = 2.f * d[1] - 3.f * d[2] + d[3];
  r[3] = d[3] - d[1];
  r[4] = 2.f * d[1] - d[2] - 2.f * d[3] + d[4];
}

}  // namespace mscnn
