// Internal interface of winograd.hip: the three data transforms of the Winograd convolution paths.
// m = output tile edge: 2 -> F(2x2, 3x3) (4x4 input tiles, 16 transform planes), 3 -> F(3x3, 3x3) (5x5 tiles, 25 planes),
// 4 -> F(4x4, 3x3) (6x6 tiles, 36 planes; wino_f4_math.h).
#pragma once
#include "common.h"

namespace mscnn {

// U[xi*4+nu] = (G g G^T)[xi][nu] for every (co, ci), written in the igemm packed layout of a 1x1 convolution with
// per-"image" weights:  wp[xinu][mt][kc][ck][BM]  (zero padded in Cout and Cin).
int wino_pack_weights(int m, const float* w, float* packed, int Cout, int Cin, int BM, int CK, int MT, int KI, hipStream_t st);

// V[xinu][ci][t] = (B^T d B)[xi][nu],  d = the 4x4 input patch of output tile t = (n, ty, tx) (zero outside the image);
// t < T real tiles, row stride T_pad (columns T..T_pad are written as zeros).
int wino_input_transform(int m, const float* x, float* V, int N, int Cin, int H, int W, int pad_h, int pad_w, int tiles_h,
                         int tiles_w, int T_pad, hipStream_t st, bool scalar_f4 = false, bool one_tile_per_lane = false);
// one_tile_per_lane: F(4x4,3x3) only -- the one-tile-per-lane vector kernel instead of the two-tile one (tune_flags bit 12, A/B runs)
// scalar_f4: F(4x4,3x3) only -- run the scalar kernels (the bodies of wino_f4_math.h, checked on the host) instead of the vectorised
// ones; mscnn_conv_desc::tune_flags bit 8, used by the bit-identity test and for A/B runs

// y[n][co][2ty + i][2tx + j] = (A^T m A)[i][j] + bias[co], optional ReLU;  m[xi][nu] = M[xinu][co][t].
// y_pool != nullptr: also write max over the tile's (in-plane) outputs to y_pool[n][co][ty][tx] (fused 2x2/2 max pooling).
int wino_output_transform(int m, const float* M, const float* bias, float* y, float* y_pool, int N, int Cout, int Ho, int Wo,
                          int tiles_h, int tiles_w, int T_pad, int relu, hipStream_t st, unsigned* amax = nullptr, bool scalar_f4 = false);
// amax != nullptr (m >= 3 only): max |y| is published as bit patterns into amax[0 .. kAmaxSlots) (atomicMax, one slot per
// workgroup; the caller zeroes the slots before the forward and takes the maximum over them)

// F(4x4,3x3) only: the output transform of layer L and the input transform of the same-resolution 3x3 / pad 1 layer L + 1 in one
// launch -- M (planes of L, row stride T_pad_m) -> V (planes of L + 1, row stride T_pad_v), bias (+ ReLU) of L in between; y of L is
// written only if y != nullptr.  Bit-identical to wino_output_transform + wino_input_transform.  strip_w / chunk_rows <= 0: chosen here.
bool wino44_outin_supported(int H, int W, int tiles_h, int tiles_w);
int wino44_output_into_input(const float* M, const float* bias, float* y, float* V, int N, int C, int H, int W, int tiles_h, int tiles_w,
                             int T_pad_m, int T_pad_v, int relu, hipStream_t st, unsigned* amax = nullptr, int strip_w = 0, int chunk_rows = 0);

}  // namespace mscnn
