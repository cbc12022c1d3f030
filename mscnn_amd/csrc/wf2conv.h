// Internal interface of wf2conv.hip: the 3x3 / pad 1 / stride 1 convolution with 64 output channels on full-resolution maps
// (conv1_2 of the VGG trunk) as ONE-launch Winograd F(2x2,3x3): input transform, the 16 plane products and the output transform
// inside one workgroup, nothing but x, the transformed filters and y / the pooled map in HBM.  conv.hip's plan selects it (AUTO);
// tune_flags bit 16 keeps the direct ring kernel of wconv.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

namespace mscnn {

struct Wf2Plan {
  int N, Cin, H, W;                 // Cout = 64
  int NTH, NTW, KI, tiles;          // workgroup tiles of 8 rows x 32 columns (4 x 16 Winograd tiles), K chunks of 8 channels
  int cus;                          // CUs of the device current at plan time: the persistent grid is min(tiles, cus) workgroups
  size_t packed_bytes;              // U = G g G^T in the layout Up[cin][xi][cout][nu]: 16 * 64 * Cin floats
};

bool wf2_plan(int N, int Cin, int H, int W, int Cout, Wf2Plan* out);
const char* wf2_kernel_name();
int wf2_pack(const Wf2Plan& p, const float* w, float* packed, hipStream_t st);
// y[N][64][H][W] = relu?(conv3x3(x, w) + bias); y_pool != nullptr: also the MAX 2x2 / stride 2 pooled map; y == nullptr (with y_pool):
// only the pooled map.  The same arithmetic in every mode: the pooled map IS the pooling of y, bit for bit.
int wf2_launch(const Wf2Plan& p, const float* x, const float* packed, const float* bias, float* y, float* y_pool, int relu, hipStream_t st);

}  // namespace mscnn
