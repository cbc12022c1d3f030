// Internal interface of wconv.hip: the direct 3x3 / pad 1 / stride 1 convolution with 64 output channels (conv1_2 of the VGG trunk:
// 64 -> 64 channels on the full-resolution map) on the ring / one-barrier structure of wgemm.hip.  conv.hip's plan selects it; every
// other shape stays on the igemm kernel (tune_flags bit 15 keeps it there for this one too).
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

namespace mscnn {

struct WconvPlan {
  int N, Cin, H, W;                 // Cout = 64
  int NTH, NTW, KI, tiles;          // tiles of 4 rows x 128 columns, K chunks of 8 channels (x 9 taps)
  int G;                            // persistent grid: slot s takes the whole tiles s, s + G, ...
  size_t packed_bytes;              // weights in the igemm packing wp[kc][tap][ck][64]
  size_t ws_bytes;                  // 0 (no partial sums)
};

// force_small: also plan maps with fewer than two tiles per CU (tests)
bool wconv_plan(int N, int Cin, int H, int W, int Cout, bool force_small, WconvPlan* out);
const char* wconv_kernel_name();

// y[N][64][H][W] = relu?(conv3x3(x, w) + bias); y_pool != nullptr: also the MAX 2x2 / stride 2 pooled map [N][64][H/2][W/2];
// y == nullptr (with y_pool): only the pooled map.  packed: pack_weights_kernel's layout with BM = 64, CK = 8 (conv.hip).
int wconv_launch(const WconvPlan& p, const float* x, const float* packed, const float* bias, float* y, float* y_pool, int relu, hipStream_t st,
                 unsigned long long* dbg = nullptr);

}  // namespace mscnn
