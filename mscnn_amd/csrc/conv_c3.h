// Internal interface of conv_c3.hip: the first convolution of the trunk (conv1_1: 3 -> 64 channels, 3x3, pad 1, stride 1).
#pragma once
#include "common.h"

namespace mscnn {

// true when the shape is one the kernel takes (Cin = 3, 3x3 / stride 1 / pad 1, Cout a multiple of 4 and <= 128, W % 4 == 0)
bool c3_plan(const mscnn_conv_desc& d, int Ho, int Wo);
const char* c3_kernel_name();
// x [N][3][H][W], w [Cout][3][3][3] (the Caffe layout, read as it is), y [N][Cout][H][W]
int c3_forward(const mscnn_conv_desc& d, const float* x, const float* w, const float* bias, float* y, unsigned* amax_out, hipStream_t st);

}  // namespace mscnn
