// Internal interface of roipool_wino.hip: the two ROI poolings of the detection sub-net (ROI window + context window over the same
// map and ROIs) fused with the F(3x3,3x3) input transform of the 3x3 convolution that consumes their concatenation.
#pragma once
#include "common.h"

namespace mscnn {

// 7 x 7 bins feeding a 3x3 / pad 0 convolution (2 x 2 tiles per ROI), C a multiple of 64
bool roipool_wino33_supported(int C, int pooled_h, int pooled_w, int conv_pad_h, int conv_pad_w);
// bytes of the scratch maps the kernel reads (`maps` below: the channel-last copy of the feature map + its sliding maxima over 2 x 2,
// 4 x 4 and 8 x 8 squares)
size_t roipool_wino33_scratch_bytes(int N, int C, int H, int W);
// feat [N][C][H][W] -> maps [4][N][H][W][C] (roipool_wino33_scratch_bytes): depends on the feature map only -- a caller may build the
// maps early, on another stream, while the proposals are still being selected
int roipool_wino33_build_maps(const float* feat, float* maps, int N, int C, int H, int W, hipStream_t st);
// maps -> V[25][2C][T_pad]: rows [0, C) from the window with pad_a, rows [C, 2C) from the window with pad_b; column 4 r + (2 ty + tx)
// = tile (ty, tx) of ROI r.  Columns >= 4 R are not written.
int roipool_wino33_forward(const float* maps, const float* rois, float* V, int R, int N, int C, int H, int W, int T_pad,
                           float spatial_scale, float pad_a, float pad_b, hipStream_t st);

}  // namespace mscnn
