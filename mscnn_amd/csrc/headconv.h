// Internal interface of headconv.hip (small-Cout proposal-head convolutions); used by the conv plan in conv.hip.
#pragma once
#include "common.h"

namespace mscnn {

struct HeadPlan {
  int entry = -1;          // index into headconv.hip's kernel table; -1: shape not covered
  int NTH = 0, NTW = 0, KI = 0, G = 0, tiles = 0;
  long total_iters = 0;
  size_t packed_bytes = 0, ws_bytes = 0;
  // witness build only (tools/micro/headvalu.hip, `make witness`): valu >= 0 = index into its table; the fields above are then its own
  int valu = -1;
  int LPR = 0, RPW = 0, TR = 0, PR = 0, RL = 0;
  size_t lds_bytes = 0;
};

bool headv_plan(const mscnn_conv_desc& d, int Ho, int Wo, HeadPlan* hp);
const char* headv_kernel_name(const HeadPlan& hp);
int headv_pack(const mscnn_conv_desc& d, const HeadPlan& hp, const float* w, float* packed, hipStream_t st);
int headv_forward(const mscnn_conv_desc& d, const HeadPlan& hp, int Ho, int Wo, const float* x, const float* packed, const float* bias,
                  float* y, void* workspace, size_t workspace_bytes, hipStream_t st);

bool head_plan(const mscnn_conv_desc& d, int Ho, int Wo, HeadPlan* hp);
const char* head_kernel_name(const HeadPlan& hp);
int head_pack(const mscnn_conv_desc& d, const HeadPlan& hp, const float* w, float* packed, hipStream_t st);
int head_forward(const mscnn_conv_desc& d, const HeadPlan& hp, int Ho, int Wo, const float* x, const float* packed,
                 const float* bias, float* y, void* workspace, size_t workspace_bytes, hipStream_t st);

}  // namespace mscnn
