// The depthwise 4x4 / stride 2 / pad 1 Deconvolution of the "-2x" nets (bilinear 2x up-sampling of conv4_3,
// examples/kitti_ped_cyc/mscnn-7s-576-2x/mscnn_deploy.prototxt:464-479) as a per-output-pixel formula, shared by the stand-alone
// kernel (elementwise.hip) and the ROI pooling that reads the up-sampled map without materialising it (roipool.hip), so that both
// produce bit-identical values.  Output (oy, ox) = sum over the <= 2 x 2 contributing taps in the reference's order (ky, then kx,
// ascending: deconv_layer.cpp:25-46 -> col2im):
//     py = oy & 1:  ky in {1, 3} (rows a, a - 1) for oy = 2a;   {0, 2} (rows a + 1, a) for oy = 2a + 1
//     px = ox & 1:  kx in {1, 3} (cols b, b - 1) for ox = 2b;   {0, 2} (cols b + 1, b) for ox = 2b + 1
// Inputs outside the map enter as 0: acc + w * 0 == acc exactly for finite data, so this equals skipping the tap.
#pragma once
#include <hip/hip_runtime.h>

namespace mscnn {

// xAA = x[rowA][colA], xAB = x[rowA][colB], xBA = x[rowB][colA], xBB = x[rowB][colB]; A = the smaller ky / kx of the pair
__device__ __forceinline__ float up2_value(const float* __restrict__ w16, int py, int px, float xAA, float xAB, float xBA, float xBB) {
  const int kyA = py ? 0 : 1, kxA = px ? 0 : 1;
  float acc = 0.f;
  acc += w16[kyA * 4 + kxA] * xAA;
  acc += w16[kyA * 4 + kxA + 2] * xAB;
  acc += w16[(kyA + 2) * 4 + kxA] * xBA;
  acc += w16[(kyA + 2) * 4 + kxA + 2] * xBB;
  return acc;
}
// input rows / columns of output coordinate o:  A (first tap) and B (second tap)
__device__ __forceinline__ int up2_srcA(int o) { return (o + 1) >> 1; }      // o = 2a -> a;  2a + 1 -> a + 1
__device__ __forceinline__ int up2_srcB(int o) { return ((o + 1) >> 1) - 1; }  // o = 2a -> a - 1;  2a + 1 -> a

}  // namespace mscnn
