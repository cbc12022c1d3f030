// Winograd data transforms for gfx950 (the batched GEMM between them is the 1x1 igemm kernel of conv.hip).
//
// Used for the 3x3 / stride 1 layers whose channel counts make the transforms cheap next to the multiply (conv2_2 ... conv6_1
// and roi_c1 of the mscnn-7s nets): y = A^T [ (G g G^T) . (B^T d B) ] A per m x m output tile.
//   * F(3x3,3x3) (second half of this file) is the path the plan selects: 25 multiplies per 3x3 tile and channel pair instead
//     of 81 -- 3.24x fewer MFMA FLOPs than the direct form that the reference computes (conv_layer.cu:8-23 im2col + SGEMM);
//   * F(2x2,3x3) (first half: 16 multiplies instead of 36, 2.25x) was the first path and stays selectable
//     (MSCNN_WINOGRAD_PLANE_M=2) for A/B runs and as a second witness in the tests.
// Results differ from the direct form only by fp32 rounding (different but equally short summation trees); the parity tests
// hold both to the same 1e-4 bound (measured: DESIGN.md 3.1b).
//
// F(2x2,3x3):
//   B^T = | 1  0 -1  0 |     G = | 1    0    0  |     A^T = | 1  1  1  0 |
//         | 0  1  1  0 |         | 1/2  1/2  1/2|           | 0  1 -1 -1 |
//         | 0 -1  1  0 |         | 1/2 -1/2  1/2|
//         | 0  1  0 -1 |         | 0    0    1  |
//
// All transform kernels are HBM-bound streaming kernels: one thread per (channel, tile), tiles fastest, so every transform
// plane is read / written as contiguous runs.
#include "winograd.h"
#include "x3_device.h"
#include "wino33_device.h"
#include "wino_f4_math.h"
#include <cstdint>

namespace {

// The transform kernels' one-touch streams as nontemporal accesses (bit-identical results).  WINO_NT bits (development A/B, make
// wino_nt_<bits>): 1 = M loads of the plain F(4x4,3x3) output transform, 2 = M loads of the chained output -> input kernel whatever M's
// size (the product decides by size at launch), 4 = its V stores.  Measured (tools/sessions/r05_s28.sh / r05_s29.sh,
// profiles/r05_ab_nontemporal.txt): bit 1 helps everywhere (conv2_2 63 -> 56 us), bit 4 hurts everywhere.
#ifndef WINO_NT
#define WINO_NT 1
#endif
template <int BIT>
__device__ __forceinline__ float ld_stream(const float* p) {
  if constexpr ((WINO_NT & BIT) != 0) return __builtin_nontemporal_load(p);
  else return *p;
}
template <int BIT>
__device__ __forceinline__ void st_stream(float* p, float v) {
  if constexpr ((WINO_NT & BIT) != 0) __builtin_nontemporal_store(v, p);
  else *p = v;
}

__global__ __launch_bounds__(256) void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin,
                                                          int BM, int CK, int MT, int KI) {
  const long img_stride = (long)MT * KI * CK * BM;
  const long total = (long)MT * BM * KI * CK;           // padded (co, ci) pairs
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int m = (int)(i % BM);
    long r = i / BM;
    const int ck = (int)(r % CK); r /= CK;
    const int kc = (int)(r % KI);
    const int mt = (int)(r / KI);
    const int co = mt * BM + m, ci = kc * CK + ck;
    float g[3][3];
    const bool live = co < Cout && ci < Cin;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) g[a][b] = live ? w[((long)co * Cin + ci) * 9 + a * 3 + b] : 0.f;
    float t[4][3];   // G g
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      t[0][b] = g[0][b];
      t[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
      t[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
      t[3][b] = g[2][b];
    }
    float* dst = wp + (((long)mt * KI + kc) * CK + ck) * BM + m;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const float u0 = t[a][0], u1 = 0.5f * (t[a][0] + t[a][1] + t[a][2]), u2 = 0.5f * (t[a][0] - t[a][1] + t[a][2]), u3 = t[a][2];
      dst[(a * 4 + 0) * img_stride] = u0;
      dst[(a * 4 + 1) * img_stride] = u1;
      dst[(a * 4 + 2) * img_stride] = u2;
      dst[(a * 4 + 3) * img_stride] = u3;
    }
  }
}

__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, float* __restrict__ V, int N, int Cin, int H,
                                                         int W, int pad_h, int pad_w, int tiles_h, int tiles_w, int T, int T_pad) {
  const int t = blockIdx.x * 256 + threadIdx.x;          // tile index (n, ty, tx), tx fastest
  const int ci = blockIdx.y;
  if (t >= T_pad) return;
  const long plane_stride = (long)Cin * T_pad;            // between the 16 transform planes
  float* dst = V + (long)ci * T_pad + t;
  float d[4][4];
  if (t < T) {
    const int tx = t % tiles_w, ty = (t / tiles_w) % tiles_h, n = t / (tiles_w * tiles_h);
    const float* src = x + ((long)n * Cin + ci) * H * W;
    const int h0 = 2 * ty - pad_h, w0 = 2 * tx - pad_w;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int h = h0 + i;
      const bool hok = h >= 0 && h < H;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int wv = w0 + j;
        d[i][j] = (hok && wv >= 0 && wv < W) ? src[h * W + wv] : 0.f;
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) d[i][j] = 0.f;
  }
  float r[4][4];   // B^T d
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    r[0][j] = d[0][j] - d[2][j];
    r[1][j] = d[1][j] + d[2][j];
    r[2][j] = d[2][j] - d[1][j];
    r[3][j] = d[1][j] - d[3][j];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    dst[(i * 4 + 0) * plane_stride] = r[i][0] - r[i][2];
    dst[(i * 4 + 1) * plane_stride] = r[i][1] + r[i][2];
    dst[(i * 4 + 2) * plane_stride] = r[i][2] - r[i][1];
    dst[(i * 4 + 3) * plane_stride] = r[i][1] - r[i][3];
  }
}

__global__ __launch_bounds__(256) void wino_output_kernel(const float* __restrict__ M, const float* __restrict__ bias,
                                                          float* __restrict__ y, float* __restrict__ yp, int N, int Cout, int Ho,
                                                          int Wo, int tiles_h, int tiles_w, int T, int T_pad, int relu) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int co = blockIdx.y;
  if (t >= T) return;
  const long plane_stride = (long)Cout * T_pad;
  const float* src = M + (long)co * T_pad + t;
  float m[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) m[i][j] = src[(i * 4 + j) * plane_stride];
  float r[2][4];   // A^T m
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    r[0][j] = m[0][j] + m[1][j] + m[2][j];
    r[1][j] = m[1][j] - m[2][j] - m[3][j];
  }
  const float b = bias ? bias[co] : 0.f;
  const int tx = t % tiles_w, ty = (t / tiles_w) % tiles_h, n = t / (tiles_w * tiles_h);
  float* dst = y + ((long)n * Cout + co) * Ho * Wo;
  float pooled = -3.402823466e+38f;      // fused PoolingLayer (MAX 2x2 stride 2): the tile IS the pooling window
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int oh = 2 * ty + i;
    if (oh >= Ho) continue;
    float v0 = r[i][0] + r[i][1] + r[i][2] + b;
    float v1 = r[i][1] - r[i][2] - r[i][3] + b;
    if (relu) { v0 = v0 < 0.f ? 0.f : v0; v1 = v1 < 0.f ? 0.f : v1; }
    const int ow = 2 * tx;
    if (ow < Wo && v0 > pooled) pooled = v0;
    if (ow + 1 < Wo && v1 > pooled) pooled = v1;
    if (ow + 1 < Wo && (Wo % 2 == 0)) {
      *reinterpret_cast<float2*>(dst + oh * Wo + ow) = make_float2(v0, v1);     // 8-byte aligned when Wo is even
    } else {
      if (ow < Wo) dst[oh * Wo + ow] = v0;
      if (ow + 1 < Wo) dst[oh * Wo + ow + 1] = v1;
    }
  }
  if (yp) yp[(((long)n * Cout + co) * tiles_h + ty) * tiles_w + tx] = pooled;
}


// ---- F(3x3, 3x3) -------------------------------------------------------------------------------------------------------
// For the detection sub-net's roi_c1 (3x3 over R x 1024 x 7 x 7 ROI-pooled maps, 5x5 outputs): a 2x2 grid of 3x3-output
// tiles covers the 5x5 (6x6) output with 4 x 25 multiplies per channel pair instead of 225 -- 2.25x fewer MFMA FLOPs;
// F(2x2,3x3) would need 9 tiles x 16 = 144.  Interpolation points {0, 1, -1, 2, inf} (Lavin & Gray 2016):
//
//   B^T = | 2 -1 -2  1  0 |    G = | 1/2    0     0  |    A^T = | 1  1  1  1  0 |
//         | 0 -2 -1  1  0 |        |-1/2  -1/2  -1/2 |          | 0  1 -1  2  0 |
//         | 0  2 -3  1  0 |        |-1/6   1/6  -1/6 |          | 0  1  1  4  1 |
//         | 0 -1  0  1  0 |        | 1/6   1/3   2/3 |
//         | 0  2 -1 -2  1 |        | 0     0     1   |
//
// fp32 error of this form is ~10x that of the direct sum (measured 1.2e-5 x layer scale at Cin = 1024, against 1.3e-6
// direct and the 1e-4 parity bound); MSCNN_WINOGRAD=0 selects the direct ROI-mode kernel instead.
using mscnn::bt5;      // (wino33_device.h: shared with the fused ROI-pooling input stage)

__global__ __launch_bounds__(256) void wino33_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin,
                                                            int BM, int CK, int MT, int KI) {
  const long img_stride = (long)MT * KI * CK * BM;
  const long total = (long)MT * BM * KI * CK;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int m = (int)(i % BM);
    long r = i / BM;
    const int ck = (int)(r % CK); r /= CK;
    const int kc = (int)(r % KI);
    const int mt = (int)(r / KI);
    const int co = mt * BM + m, ci = kc * CK + ck;
    const bool live = co < Cout && ci < Cin;
    float g[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int b = 0; b < 3; ++b) g[a][b] = live ? w[((long)co * Cin + ci) * 9 + a * 3 + b] : 0.f;
    float t[5][3];   // G g
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      t[0][b] = 0.5f * g[0][b];
      t[1][b] = -0.5f * (g[0][b] + g[1][b] + g[2][b]);
      t[2][b] = (-g[0][b] + g[1][b] - g[2][b]) * (1.f / 6.f);
      t[3][b] = g[0][b] * (1.f / 6.f) + g[1][b] * (1.f / 3.f) + g[2][b] * (2.f / 3.f);
      t[4][b] = g[2][b];
    }
    float* dst = wp + (((long)mt * KI + kc) * CK + ck) * BM + m;
#pragma unroll
    for (int a = 0; a < 5; ++a) {
      const float x0 = t[a][0], x1 = t[a][1], x2 = t[a][2];
      dst[(a * 5 + 0) * img_stride] = 0.5f * x0;
      dst[(a * 5 + 1) * img_stride] = -0.5f * (x0 + x1 + x2);
      dst[(a * 5 + 2) * img_stride] = (-x0 + x1 - x2) * (1.f / 6.f);
      dst[(a * 5 + 3) * img_stride] = x0 * (1.f / 6.f) + x1 * (1.f / 3.f) + x2 * (2.f / 3.f);
      dst[(a * 5 + 4) * img_stride] = x2;
    }
  }
}

// Input transform of the ROI maps.  x is [roi][channel][H*W] with H*W <= 64 floats per map: for a fixed ROI, 16 consecutive
// channels are one contiguous run (3136 B for 7x7), while one channel of consecutive ROIs is 200 KB apart.  A workgroup
// therefore stages (8 ROIs x 16 channels) through LDS with coalesced float4 loads and then lets thread = (channel, tile)
// with tiles fastest do the 5x5 transform, so every one of the 25 planes V[xinu][ci][t] is written as 128-byte runs
// (8 ROIs x tiles-per-ROI consecutive t).  157 -> see DESIGN.md (the per-(channel, tile) gather form read 64 cache lines per
// load instruction).
constexpr int kW33Rois = 8, kW33Ch = 16, kW33MaxHW = 64;

__global__ __launch_bounds__(256) void wino33_input_kernel(const float* __restrict__ x, float* __restrict__ V, int N, int Cin, int H,
                                                           int W, int pad_h, int pad_w, int tiles_h, int tiles_w, int T, int T_pad) {
  __shared__ __attribute__((aligned(16))) float sm[kW33Rois * kW33Ch * kW33MaxHW];
  const int tid = threadIdx.x;
  const int r0 = blockIdx.x * kW33Rois, c0 = blockIdx.y * kW33Ch;
  const int HW = H * W, tpr = tiles_h * tiles_w;
  const int nch = min(kW33Ch, Cin - c0);
  const int run = nch * HW;                              // contiguous floats per ROI
  // ---- stage: float4 where the run allows it (16 channels x 49 floats = 196 float4), scalars otherwise
  for (int rl = 0; rl < kW33Rois; ++rl) {
    const int r = r0 + rl;
    if (r >= N) break;
    const float* src = x + ((long)r * Cin + c0) * HW;
    float* dst = sm + rl * (kW33Ch * HW);
    if ((run & 3) == 0 && ((((long)r * Cin + c0) * HW) & 3) == 0) {
      for (int i = tid; i < run / 4; i += 256) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
    } else {
      for (int i = tid; i < run; i += 256) dst[i] = src[i];
    }
  }
  __syncthreads();
  // ---- transform: item = (channel, roi, tile) with (roi, tile) fastest
  const int per_ch = kW33Rois * tpr;
  const long plane_stride = (long)Cin * T_pad;
  for (int it = tid; it < nch * per_ch; it += 256) {
    const int c = it / per_ch, q = it % per_ch;
    const int rl = q / tpr, tl = q % tpr;
    const int r = r0 + rl;
    if (r >= N) continue;
    const int ty = tl / tiles_w, tx = tl % tiles_w;
    const float* map = sm + (rl * kW33Ch + c) * HW;
    const int h0 = 3 * ty - pad_h, w0 = 3 * tx - pad_w;
    float d[5][5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int h = h0 + i;
      const bool hok = h >= 0 && h < H;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int wv = w0 + j;
        d[i][j] = (hok && wv >= 0 && wv < W) ? map[h * W + wv] : 0.f;
      }
    }
    float rr[5][5];   // B^T d (columns of d)
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const float col[5] = {d[0][j], d[1][j], d[2][j], d[3][j], d[4][j]};
      float o[5];
      bt5(col, o);
#pragma unroll
      for (int i = 0; i < 5; ++i) rr[i][j] = o[i];
    }
    float* dst = V + (long)(c0 + c) * T_pad + (long)r * tpr + tl;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      float o[5];
      bt5(rr[i], o);     // (B^T d) B: the same combination along the row
#pragma unroll
      for (int j = 0; j < 5; ++j) dst[(i * 5 + j) * plane_stride] = o[j];
    }
  }
  // columns T .. T_pad of every plane (GEMM padding) must be zero: the last ROI block writes them
  if (blockIdx.x == gridDim.x - 1) {
    for (int i = tid; i < nch * (T_pad - T); i += 256) {
      const int c = i / (T_pad - T), t = T + i % (T_pad - T);
#pragma unroll 1
      for (int pl = 0; pl < 25; ++pl) V[pl * plane_stride + (long)(c0 + c) * T_pad + t] = 0.f;
    }
  }
}

// F(3x3,3x3) input transform for whole image planes (one thread per (channel, tile), tiles fastest), as wino_input_kernel.
__global__ __launch_bounds__(256) void wino33_input_plane_kernel(const float* __restrict__ x, float* __restrict__ V, int N, int Cin,
                                                                 int H, int W, int pad_h, int pad_w, int tiles_h, int tiles_w,
                                                                 int T, int T_pad) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int ci = blockIdx.y;
  if (t >= T_pad) return;
  const long plane_stride = (long)Cin * T_pad;
  float* dst = V + (long)ci * T_pad + t;
  float d[5][5];
  if (t < T) {
    const int tx = t % tiles_w, ty = (t / tiles_w) % tiles_h, n = t / (tiles_w * tiles_h);
    const float* src = x + ((long)n * Cin + ci) * H * W;
    const int h0 = 3 * ty - pad_h, w0 = 3 * tx - pad_w;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int h = h0 + i;
      const bool hok = h >= 0 && h < H;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const int wv = w0 + j;
        d[i][j] = (hok && wv >= 0 && wv < W) ? src[h * W + wv] : 0.f;
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j) d[i][j] = 0.f;
  }
  float r[5][5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const float col[5] = {d[0][j], d[1][j], d[2][j], d[3][j], d[4][j]};
    float o[5];
    bt5(col, o);
#pragma unroll
    for (int i = 0; i < 5; ++i) r[i][j] = o[i];
  }
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    float o[5];
    bt5(r[i], o);
#pragma unroll
    for (int j = 0; j < 5; ++j) dst[(i * 5 + j) * plane_stride] = o[j];
  }
}

__global__ __launch_bounds__(256) void wino33_output_kernel(const float* __restrict__ M, const float* __restrict__ bias,
                                                            float* __restrict__ y, int N, int Cout, int Ho, int Wo, int tiles_h,
                                                            int tiles_w, int T, int T_pad, int relu, unsigned* __restrict__ amax) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int co = blockIdx.y;
  unsigned am = 0;
  if (t < T) {
  const long plane_stride = (long)Cout * T_pad;
  const float* src = M + (long)co * T_pad + t;
  float r[3][5];   // A^T m
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const float m0 = src[(0 * 5 + j) * plane_stride], m1 = src[(1 * 5 + j) * plane_stride], m2 = src[(2 * 5 + j) * plane_stride];
    const float m3 = src[(3 * 5 + j) * plane_stride], m4 = src[(4 * 5 + j) * plane_stride];
    r[0][j] = m0 + m1 + m2 + m3;
    r[1][j] = m1 - m2 + 2.f * m3;
    r[2][j] = m1 + m2 + 4.f * m3 + m4;
  }
  const float b = bias ? bias[co] : 0.f;
  const int tx = t % tiles_w, ty = (t / tiles_w) % tiles_h, n = t / (tiles_w * tiles_h);
  float* dst = y + ((long)n * Cout + co) * Ho * Wo;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int oh = 3 * ty + i;
    if (oh >= Ho) continue;
    float v[3];
    v[0] = r[i][0] + r[i][1] + r[i][2] + r[i][3] + b;
    v[1] = r[i][1] - r[i][2] + 2.f * r[i][3] + b;
    v[2] = r[i][1] + r[i][2] + 4.f * r[i][3] + r[i][4] + b;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int ow = 3 * tx + j;
      if (ow >= Wo) continue;
      float u = v[j];
      if (relu) u = u < 0.f ? 0.f : u;
      dst[oh * Wo + ow] = u;
      am = max(am, __float_as_uint(u) & 0x7fffffffu);
    }
  }
  }
  if (amax) mscnn::publish_amax(am, amax, blockIdx.y * gridDim.x + blockIdx.x);
}


// Output transform of the ROI maps (roi_c1: y is [roi][channel][Ho*Wo] with Ho*Wo <= 64, 5x5 for the KITTI nets).  The generic
// kernel above lets consecutive lanes (consecutive tiles) store 4-byte pieces 12 .. 60 bytes apart and, from one ROI to the next,
// Cout * Ho * Wo * 4 bytes apart: 27 MB written as 100-byte fragments, 65 us on the 7s-576 frame (2.1 TB/s over M + y).  Here a
// workgroup owns 16 ROIs x 8 channels, the mirror image of wino33_input_kernel: thread = (channel, roi, tile) with (roi, tile)
// fastest, so each of the 25 plane reads of a wave covers 256-byte runs of M; the 3x3 outputs go to LDS in y's own layout and
// leave as float4 runs of 8 channels x Ho*Wo floats per ROI.  Same expressions in the same order: bit-identical results.
constexpr int kW33ORois = 16, kW33OCh = 8;

__global__ __launch_bounds__(256) void wino33_output_roi_kernel(const float* __restrict__ M, const float* __restrict__ bias,
                                                                float* __restrict__ y, int N, int Cout, int Ho, int Wo, int tiles_h,
                                                                int tiles_w, int T_pad, int relu, unsigned* __restrict__ amax) {
  __shared__ __attribute__((aligned(16))) float sm[kW33ORois * kW33OCh * kW33MaxHW];
  const int tid = threadIdx.x;
  const int r0 = blockIdx.x * kW33ORois, c0 = blockIdx.y * kW33OCh;
  const int HW = Ho * Wo, tpr = tiles_h * tiles_w;
  const int nch = min(kW33OCh, Cout - c0), nroi = min(kW33ORois, N - r0);
  const long plane_stride = (long)Cout * T_pad;
  const int per_ch = nroi * tpr;
  unsigned am = 0;
  for (int it = tid; it < nch * per_ch; it += 256) {
    const int c = it / per_ch, q = it % per_ch;
    const int rl = q / tpr, tl = q % tpr;
    const int ty = tl / tiles_w, tx = tl % tiles_w;
    const float* src = M + (long)(c0 + c) * T_pad + (long)(r0 + rl) * tpr + tl;
    float r[3][5];   // A^T m
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const float m0 = src[(0 * 5 + j) * plane_stride], m1 = src[(1 * 5 + j) * plane_stride], m2 = src[(2 * 5 + j) * plane_stride];
      const float m3 = src[(3 * 5 + j) * plane_stride], m4 = src[(4 * 5 + j) * plane_stride];
      r[0][j] = m0 + m1 + m2 + m3;
      r[1][j] = m1 - m2 + 2.f * m3;
      r[2][j] = m1 + m2 + 4.f * m3 + m4;
    }
    const float b = bias ? bias[c0 + c] : 0.f;
    float* dst = sm + (rl * nch + c) * HW;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int oh = 3 * ty + i;
      if (oh >= Ho) continue;
      float v[3];
      v[0] = r[i][0] + r[i][1] + r[i][2] + r[i][3] + b;
      v[1] = r[i][1] - r[i][2] + 2.f * r[i][3] + b;
      v[2] = r[i][1] + r[i][2] + 4.f * r[i][3] + r[i][4] + b;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int ow = 3 * tx + j;
        if (ow >= Wo) continue;
        float u = v[j];
        if (relu) u = u < 0.f ? 0.f : u;
        dst[oh * Wo + ow] = u;
        am = max(am, __float_as_uint(u) & 0x7fffffffu);
      }
    }
  }
  __syncthreads();
  const int run = nch * HW;                               // contiguous floats of y per ROI
  for (int rl = 0; rl < nroi; ++rl) {
    float* dst = y + ((long)(r0 + rl) * Cout + c0) * HW;
    const float* src = sm + rl * run;
    if ((run & 3) == 0 && ((((long)(r0 + rl) * Cout + c0) * HW) & 3) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0) {
      for (int i = tid; i < run / 4; i += 256) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
    } else {
      for (int i = tid; i < run; i += 256) dst[i] = src[i];
    }
  }
  if (amax) mscnn::publish_amax(am, amax, blockIdx.y * gridDim.x + blockIdx.x);
}


// F(3x3,3x3) output transform with the fused MAX 2x2 / stride 2 pooling: 3x3 tiles and 2x2 windows meet every 6 pixels, so a
// thread owns a 2x2 group of tiles (= 6x6 outputs = 3x3 pooling windows).  Tiles t and t+1 are neighbours in a plane, so the
// 25 x 2 M reads per tile row are float2 loads; the six output rows are written as three float2 each.  Needs an even number
// of tile rows and columns (true for every pooled layer of the deploy nets at their native input sizes).
__device__ __forceinline__ void at5(const float m[5], float o[3]) {
  o[0] = m[0] + m[1] + m[2] + m[3];
  o[1] = m[1] - m[2] + 2.f * m[3];
  o[2] = m[1] + m[2] + 4.f * m[3] + m[4];
}

__global__ __launch_bounds__(256) void wino33_output_pool_kernel(const float* __restrict__ M, const float* __restrict__ bias,
                                                                 float* __restrict__ y, float* __restrict__ yp, int N, int Cout,
                                                                 int Ho, int Wo, int tiles_h, int tiles_w, int T_pad, int relu,
                                                                 unsigned* __restrict__ amax) {
  const int sw = tiles_w / 2, sh = tiles_h / 2;
  const int S = N * sh * sw;
  const int sidx = blockIdx.x * 256 + threadIdx.x;
  const int co = blockIdx.y;
  unsigned am = 0;
  if (sidx < S) {
  const int sx = sidx % sw, sy = (sidx / sw) % sh, n = sidx / (sw * sh);
  const long plane_stride = (long)Cout * T_pad;
  const float b = bias ? bias[co] : 0.f;
  float out[6][6];
#pragma unroll
  for (int half = 0; half < 2; ++half) {                  // upper / lower pair of tiles
    const long t0 = ((long)n * tiles_h + 2 * sy + half) * tiles_w + 2 * sx;
    const float* src = M + (long)co * T_pad + t0;
    float ra[3][5], rb[3][5];                             // A^T m of the left / right tile
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      float ca[5], cb[5];
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const float2 v = *reinterpret_cast<const float2*>(src + (i * 5 + j) * plane_stride);
        ca[i] = v.x; cb[i] = v.y;
      }
      float oa[3], ob[3];
      at5(ca, oa); at5(cb, ob);
#pragma unroll
      for (int i = 0; i < 3; ++i) { ra[i][j] = oa[i]; rb[i][j] = ob[i]; }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      float oa[3], ob[3];
      at5(ra[i], oa); at5(rb[i], ob);
#pragma unroll
      for (int j = 0; j < 3; ++j) { out[half * 3 + i][j] = oa[j] + b; out[half * 3 + i][3 + j] = ob[j] + b; }
    }
  }
  float* dst = y + ((long)n * Cout + co) * Ho * Wo;
  const int oh0 = 6 * sy, ow0 = 6 * sx;
#pragma unroll
  for (int i = 0; i < 6; ++i) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float v = out[i][j];
      if (relu) v = v < 0.f ? 0.f : v;
      const bool in = oh0 + i < Ho && ow0 + j < Wo;
      if (in) am = max(am, __float_as_uint(v) & 0x7fffffffu);
      out[i][j] = in ? v : -3.402823466e+38f;
    }
    const int oh = oh0 + i;
    if (oh >= Ho) continue;
    if ((Wo & 1) == 0 && ow0 + 5 < Wo) {
#pragma unroll
      for (int j = 0; j < 6; j += 2) *reinterpret_cast<float2*>(dst + oh * Wo + ow0 + j) = make_float2(out[i][j], out[i][j + 1]);
    } else {
#pragma unroll
      for (int j = 0; j < 6; ++j) if (ow0 + j < Wo) dst[oh * Wo + ow0 + j] = out[i][j];
    }
  }
  const int Hp = (Ho + 1) / 2, Wp = (Wo + 1) / 2;
  float* pd = yp + ((long)n * Cout + co) * Hp * Wp;
#pragma unroll
  for (int pi = 0; pi < 3; ++pi)
#pragma unroll
    for (int pj = 0; pj < 3; ++pj) {
      const int ph = 3 * sy + pi, pw = 3 * sx + pj;
      if (ph >= Hp || pw >= Wp) continue;
      float m = out[2 * pi][2 * pj];
      if (out[2 * pi][2 * pj + 1] > m) m = out[2 * pi][2 * pj + 1];
      if (out[2 * pi + 1][2 * pj] > m) m = out[2 * pi + 1][2 * pj];
      if (out[2 * pi + 1][2 * pj + 1] > m) m = out[2 * pi + 1][2 * pj + 1];
      pd[ph * Wp + pw] = m;
    }
  }
  if (amax) mscnn::publish_amax(am, amax, blockIdx.y * gridDim.x + blockIdx.x);
}


// ---- F(4x4, 3x3) -------------------------------------------------------------------------------------------------------
// 36 multiplies per 4x4 output tile and channel pair instead of 144: 4x fewer MFMA FLOPs than the direct form, 19 % fewer than
// F(3x3,3x3), and 36 / 16 = 2.25 plane elements per output instead of 25 / 9 = 2.78.  Interpolation points {0, 1, -1, 2, -1/2, inf}:
// with them the fp32 error is that of the F(3x3,3x3) form above (profiles/r02_study_winograd_f4_numerics.txt; the textbook points
// {0, +-1, +-2} treble it).  The per-thread bodies live in wino_f4_math.h as host + device functions, so the host can run exactly what a
// GPU thread runs (tests/test_wino_f4_model.py: 1-D transforms against the exact matrices, whole layers against the direct convolution).
// Same kernel shape as the other forms: one thread per (channel, tile), tiles fastest.  MSCNN_CONV_ALGO_WINO_F4 selects it.
// STATUS (end of round 2): written and CPU-checked (transform arithmetic, tests/test_wino_f4_model.py), NOT yet run on the GPU --
// no default path selects it; tests/test_gpu_ops.py::test_conv_winograd_f4x4 is skipped unless MSCNN_TEST_WINO_F4=1.
__global__ __launch_bounds__(256) void wino44_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin,
                                                            int BM, int CK, int MT, int KI) {
  const long total = (long)MT * BM * KI * CK;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256) wino_f4::weight_pair(w, wp, i, Cout, Cin, BM, CK, MT, KI);
}

__global__ __launch_bounds__(256) void wino44_input_plane_kernel(const float* __restrict__ x, float* __restrict__ V, int N, int Cin,
                                                                 int H, int W, int pad_h, int pad_w, int tiles_h, int tiles_w,
                                                                 int T, int T_pad) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T_pad) return;
  wino_f4::input_tile(x, V, t, blockIdx.y, Cin, H, W, pad_h, pad_w, tiles_h, tiles_w, T, T_pad);
}

__global__ __launch_bounds__(256) void wino44_output_kernel(const float* __restrict__ M, const float* __restrict__ bias,
                                                            float* __restrict__ y, float* __restrict__ yp, int N, int Cout, int Ho,
                                                            int Wo, int tiles_h, int tiles_w, int T, int T_pad, int relu,
                                                            unsigned* __restrict__ amax) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  unsigned am = 0;
  if (t < T) am = wino_f4::output_tile(M, bias, y, yp, t, blockIdx.y, Cout, Ho, Wo, tiles_h, tiles_w, T_pad, relu);
  if (amax) mscnn::publish_amax(am, amax, blockIdx.y * gridDim.x + blockIdx.x);
}


// Vectorised forms of the two F(4x4,3x3) data transforms for the common geometry (pad 1, W a multiple of 4, 16-byte aligned planes):
// a 4x4 tile's columns are one aligned float4 per input row, so a lane loads SIX float4 (its 4 new columns of the 6 patch rows)
// instead of 36 scalars at a 16-byte lane stride (every cache line fetched six times through the L1), and takes the patch's
// left / right column from its neighbour lanes (tiles of one tile row sit in consecutive lanes).  One wave = up to 64 consecutive
// tiles of ONE tile row; the arithmetic is wino_f4::bt6 / at6 in the scalar kernels' order, so the planes are bit-identical
// (tests/test_gpu_ops.py::test_wino_f4_vector_transforms_bit_identical).  Measured: profiles/r03_ab_f4_transforms.txt.
__global__ __launch_bounds__(256) void wino44_input_vec_kernel(const float* __restrict__ x, float* __restrict__ V, int N, int Cin, int H, int W,
                                                               int pad_h, int tiles_h, int tiles_w, int segs, int T_pad) {
  const int lane = threadIdx.x & 63;
  const int wv = blockIdx.x * 4 + (threadIdx.x >> 6);            // wave -> (n, ty, segment of the tile row)
  const int ci = blockIdx.y;
  const int seg = wv % segs, row = wv / segs;                    // row = n * tiles_h + ty
  if (row >= N * tiles_h) return;
  const int ty = row % tiles_h, n = row / tiles_h;
  const int tx = seg * 64 + lane;
  const bool live = tx < tiles_w;
  const float* src = x + ((long)n * Cin + ci) * H * W;
  float d[6][6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int h = 4 * ty - pad_h + i;
    const bool ok = live && h >= 0 && h < H;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) v = *reinterpret_cast<const float4*>(src + (long)h * W + 4 * tx);
    float left = __shfl_up(v.w, 1), right = __shfl_down(v.x, 1);
    // wave borders: the neighbour tile lives in another wave (or outside the image: zero padding)
    if (lane == 0) left = (ok && tx > 0) ? src[(long)h * W + 4 * tx - 1] : 0.f;
    if (lane == 63 || tx == tiles_w - 1) right = (ok && tx + 1 < tiles_w) ? src[(long)h * W + 4 * tx + 4] : 0.f;
    d[i][0] = left; d[i][1] = v.x; d[i][2] = v.y; d[i][3] = v.z; d[i][4] = v.w; d[i][5] = right;
  }
  if (!live) return;
  const long plane_stride = (long)Cin * T_pad;
  float* dst = V + (long)ci * T_pad + (long)row * tiles_w + tx;
  float r[6][6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const float col[6] = {d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]};
    float o[6];
    wino_f4::bt6(col, o);
#pragma unroll
    for (int i = 0; i < 6; ++i) r[i][j] = o[i];
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    float o[6];
    wino_f4::bt6(r[i], o);
#pragma unroll
    for (int j = 0; j < 6; ++j) dst[(i * 6 + j) * plane_stride] = o[j];
  }
}

// Two horizontally adjacent tiles per lane (tiles_w even): a lane loads 32 contiguous bytes per patch row, the halo between its two
// tiles is its own data, and every plane store is a float2 -- a wave writes 512-byte runs of V instead of 256 (the input transform
// sat at 4.6 TB/s against the output transform's 6.1, which reads M in the same 256-byte pattern but writes whole image rows).
// Same bt6 chains per tile => bit-identical planes.
__global__ __launch_bounds__(256) void wino44_input_vec2_kernel(const float* __restrict__ x, float* __restrict__ V, int N, int Cin, int H, int W,
                                                                int pad_h, int tiles_h, int tiles_w, int segs, int T_pad) {
  const int lane = threadIdx.x & 63;
  const int wv = blockIdx.x * 4 + (threadIdx.x >> 6);            // wave -> (n, ty, segment of 128 tiles of the tile row)
  const int ci = blockIdx.y;
  const int seg = wv % segs, row = wv / segs;                    // row = n * tiles_h + ty
  if (row >= N * tiles_h) return;
  const int ty = row % tiles_h, n = row / tiles_h;
  const int tx = (seg * 64 + lane) * 2;                          // this lane's tiles: tx, tx + 1
  const bool live = tx < tiles_w;
  const float* src = x + ((long)n * Cin + ci) * H * W;
  float da[6][6], db[6][6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int h = 4 * ty - pad_h + i;
    const bool ok = live && h >= 0 && h < H;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (ok) {
      const float4* p = reinterpret_cast<const float4*>(src + (long)h * W + 4 * tx);
      a = p[0]; b = p[1];
    }
    float left = __shfl_up(b.w, 1), right = __shfl_down(a.x, 1);
    // wave borders: the neighbour tile lives in another wave (or outside the image: zero padding)
    if (lane == 0) left = (ok && tx > 0) ? src[(long)h * W + 4 * tx - 1] : 0.f;
    if (lane == 63 || tx + 2 >= tiles_w) right = (ok && tx + 2 < tiles_w) ? src[(long)h * W + 4 * tx + 8] : 0.f;
    da[i][0] = left; da[i][1] = a.x; da[i][2] = a.y; da[i][3] = a.z; da[i][4] = a.w; da[i][5] = b.x;
    db[i][0] = a.w;  db[i][1] = b.x; db[i][2] = b.y; db[i][3] = b.z; db[i][4] = b.w; db[i][5] = right;
  }
  if (!live) return;
  const long plane_stride = (long)Cin * T_pad;
  float* dst = V + (long)ci * T_pad + (long)row * tiles_w + tx;
  float ra[6][6], rb[6][6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const float ca[6] = {da[0][j], da[1][j], da[2][j], da[3][j], da[4][j], da[5][j]};
    const float cb[6] = {db[0][j], db[1][j], db[2][j], db[3][j], db[4][j], db[5][j]};
    float oa[6], ob[6];
    wino_f4::bt6(ca, oa);
    wino_f4::bt6(cb, ob);
#pragma unroll
    for (int i = 0; i < 6; ++i) { ra[i][j] = oa[i]; rb[i][j] = ob[i]; }
  }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    float oa[6], ob[6];
    wino_f4::bt6(ra[i], oa);
    wino_f4::bt6(rb[i], ob);
#pragma unroll
    for (int j = 0; j < 6; ++j) *reinterpret_cast<float2*>(dst + (i * 6 + j) * plane_stride) = make_float2(oa[j], ob[j]);
  }
}

__global__ __launch_bounds__(256) void wino44_output_vec_kernel(const float* __restrict__ M, const float* __restrict__ bias, float* __restrict__ y,
                                                                float* __restrict__ yp, int N, int Cout, int Ho, int Wo, int tiles_h,
                                                                int tiles_w, int T, int T_pad, int relu, unsigned* __restrict__ amax) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  const int co = blockIdx.y;
  unsigned am = 0;
  if (t < T) {
    const long plane_stride = (long)Cout * T_pad;
    const float* src = M + (long)co * T_pad + t;
    float r[4][6];   // A^T m (columns of m)
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      float col[6], o[4];
#pragma unroll
      for (int i = 0; i < 6; ++i) col[i] = ld_stream<1>(src + (i * 6 + j) * plane_stride);
      wino_f4::at6(col, o);
#pragma unroll
      for (int i = 0; i < 4; ++i) r[i][j] = o[i];
    }
    const float b = bias ? bias[co] : 0.f;
    const int tx = t % tiles_w, ty = (t / tiles_w) % tiles_h, n = t / (tiles_w * tiles_h);
    float* dst = y + ((long)n * Cout + co) * Ho * Wo;
    float out[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float o[4];
      wino_f4::at6(r[i], o);     // (A^T m) A
      const int oh = 4 * ty + i;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float u = o[j] + b;
        if (relu) u = u < 0.f ? 0.f : u;
        out[i][j] = u;
      }
      if (oh < Ho) {             // (Wo is a multiple of 4 here: every column of the tile is inside the plane)
        if (y) *reinterpret_cast<float4*>(dst + (long)oh * Wo + 4 * tx) = make_float4(out[i][0], out[i][1], out[i][2], out[i][3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) { const unsigned a = wino_f4::abs_bits(out[i][j]); am = a > am ? a : am; }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) out[i][j] = -3.402823466e+38f;
      }
    }
    if (yp) {
      const int Hp = (Ho + 1) / 2, Wp = Wo / 2;
      float* pd = yp + ((long)n * Cout + co) * Hp * Wp;
#pragma unroll
      for (int pi = 0; pi < 2; ++pi) {
        const int ph = 2 * ty + pi;
        if (ph >= Hp) continue;
        float m2[2];
#pragma unroll
        for (int pj = 0; pj < 2; ++pj) {
          float m = out[2 * pi][2 * pj];
          if (out[2 * pi][2 * pj + 1] > m) m = out[2 * pi][2 * pj + 1];
          if (out[2 * pi + 1][2 * pj] > m) m = out[2 * pi + 1][2 * pj];
          if (out[2 * pi + 1][2 * pj + 1] > m) m = out[2 * pi + 1][2 * pj + 1];
          m2[pj] = m;
        }
        *reinterpret_cast<float2*>(pd + (long)ph * Wp + 2 * tx) = make_float2(m2[0], m2[1]);
      }
    }
  }
  if (amax) mscnn::publish_amax(am, amax, blockIdx.y * gridDim.x + blockIdx.x);
}

// ---- F(4x4,3x3) output transform of layer L fused with the F(4x4,3x3) input transform of layer L + 1 ------------------------------
// For two 3x3 / pad 1 / stride 1 layers of the same resolution (conv2_1 -> conv2_2, conv3_1 -> 3_2 -> 3_3, conv4_1 -> 4_2 -> 4_3) the
// tile grids coincide: output tile (q, j) of L is rows 4q .. 4q + 3, input tile (p, j) of L + 1 rows 4p - 1 .. 4p + 4.  The unfused pair
// moves M (2.25 y) + y | y + V (2.25 y) = 6.5 units of the activation through HBM; this kernel moves M + V = 4.5 (+ the halo columns
// of M) and writes y only when asked to.
// A workgroup owns one channel and a STRIP of <= 62 tile columns, and walks down the tile rows of its row chunk: a wave = one tile row,
// a lane = one tile column of the strip + one halo column either side (64 = 62 + 2).  Per step the waves transform RW tile rows
// of M into a ring of y rows in LDS (RING tile rows x 4 x 256 S floats), then -- one barrier later -- transform the RW input tile
// rows whose six y rows are complete (the row above, its own four, the first of the row below).  Nothing is re-read vertically inside
// a chunk; horizontally a strip re-reads its two halo columns (62 / 60).  The arithmetic is at6 / bt6 in the order of the two kernels
// above, on the same fp32 values, so V (and y) are bit-identical to the unfused pair (tests/test_gpu_ops.py).
// S strips x RW tile rows per step = the waves of a workgroup: the strips of one tile row belong to ONE workgroup, so that a row of V
// (and of M) is touched as one contiguous run by one CU -- with the strips spread over workgroups (which land on different XCDs) the
// 240-byte runs shared their cache lines across L2s and the kernel fell to 3.5 TB/s on the 240-column maps.
template <int S, int RW, int RING, int NTM>
__global__ __launch_bounds__(64 * S * RW) void wino44_outin_kernel(const float* __restrict__ M, const float* __restrict__ bias, float* __restrict__ y,
                                                           float* __restrict__ V, int N, int C, int H, int W, int tiles_h, int tiles_w,
                                                           int T_pad_m, int T_pad_v, int relu, int strip_w, int sgroups,
                                                           int chunks, int chunk_rows, unsigned* __restrict__ amax) {
  static_assert((RING & (RING - 1)) == 0 && RING >= RW + 2, "ring: the tile rows of two consecutive steps");
  __shared__ float4 ring[RING][4][64 * S];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int strip = wave % S, w = wave / S;
  const int c = blockIdx.y;
  int b = blockIdx.x;
  const int sg = b % sgroups; b /= sgroups;                      // (maps of 6 / 8 strips: two workgroups per tile row, S strips each)
  const int chunk = b % chunks;
  const int n = b / chunks;
  const int j0 = (sg * S + strip) * strip_w, sw = max(0, min(strip_w, tiles_w - j0));
  const int r0 = chunk * chunk_rows, r1 = min(r0 + chunk_rows, tiles_h);
  const int jq = j0 - 1 + lane;                                  // phase 1: this lane's tile column (halo included)
  const bool colq = lane < sw + 2 && jq >= 0 && jq < tiles_w;
  const int jp = j0 + lane;                                      // phase 2: this lane's tile column
  const bool colp = lane < sw;
  const long stride_m = (long)C * T_pad_m, stride_v = (long)C * T_pad_v;
  const float bv = bias ? bias[c] : 0.f;
  unsigned am = 0;
  // the 36 plane values of this lane's tile of tile row q (the NEXT step's loads are issued before this step's phase 2, so that M
  // streams while the input transform computes and stores: without it the kernel is latency-bound on maps that exceed the MALL)
  float m[36];
  auto load_m = [&](int q) {
    if (q <= r1 && q >= 0 && q < tiles_h && colq) {
      const float* src = M + (long)c * T_pad_m + ((long)n * tiles_h + q) * tiles_w + jq;
      // (M larger than the Infinity Cache -- conv2_1's 318 MB -- is read nontemporally: 134 -> 119 us; planes that fit -- conv3 / conv4:
      // 160 / 82 MB, still cached from the GEMM that wrote them -- are not: 61 -> 64, 33 -> 36 us.  profiles/r05_ab_nontemporal.txt)
      // (a compile-time choice: with a run-time flag the optimiser merged the two loads into a plain one)
      if constexpr (NTM != 0 || (WINO_NT & 2) != 0) {
#pragma unroll
        for (int e = 0; e < 36; ++e) m[e] = __builtin_nontemporal_load(src + e * stride_m);
      } else {
#pragma unroll
        for (int e = 0; e < 36; ++e) m[e] = src[e * stride_m];
      }
    }
  };
  load_m(r0 - 1 + w);
  for (int k = 0; r0 - 2 + RW * k < r1; ++k) {
    // ---- phase 1: M -> y rows of tile row q (one per wave) into the ring
    const int q = r0 - 1 + RW * k + w;
    if (q <= r1) {
      float out[4][4];
      if (q >= 0 && q < tiles_h && colq) {
        float r[4][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
          float col[6], o[4];
#pragma unroll
          for (int i = 0; i < 6; ++i) col[i] = m[i * 6 + j];
          wino_f4::at6(col, o);
#pragma unroll
          for (int i = 0; i < 4; ++i) r[i][j] = o[i];
        }
        const bool own = q >= r0 && q < r1 && lane >= 1 && lane <= sw;
        float* dst = y ? y + ((long)n * C + c) * H * W + (long)(4 * q) * W + 4 * jq : nullptr;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float o[4];
          wino_f4::at6(r[i], o);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float u = o[j] + bv;
            if (relu) u = u < 0.f ? 0.f : u;
            out[i][j] = u;
          }
          if (own) {
            if (dst) *reinterpret_cast<float4*>(dst + (long)i * W) = make_float4(out[i][0], out[i][1], out[i][2], out[i][3]);
#pragma unroll
            for (int j = 0; j < 4; ++j) { const unsigned a = wino_f4::abs_bits(out[i][j]); am = a > am ? a : am; }
          }
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) out[i][j] = 0.f;            // outside the map: the next layer's zero padding
      }
      const int slot = (q - r0 + 1) & (RING - 1);
#pragma unroll
      for (int i = 0; i < 4; ++i) ring[slot][i][strip * 64 + lane] = make_float4(out[i][0], out[i][1], out[i][2], out[i][3]);
    }
    if (r0 - 2 + RW * (k + 1) < r1) load_m(q + RW);
    __syncthreads();
    // ---- phase 2: y rows 4p - 1 .. 4p + 4 -> the 36 planes of input tile row p (one per wave)
    const int p = r0 - 2 + RW * k + w;
    if (p >= r0 && p < r1 && colp) {
      const float* base = reinterpret_cast<const float*>(&ring[0][0][0]);
      float d[6][6];
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int yr = 4 * (p - r0 + 1) - 1 + i;                  // ring row: (tile row - r0 + 1) * 4 + row in tile
        const float* row = base + ((yr >> 2) & (RING - 1)) * (1024 * S) + (yr & 3) * (256 * S) + 256 * strip + 4 * lane;
        const float4 mid = *reinterpret_cast<const float4*>(row + 4);
        d[i][0] = row[3]; d[i][1] = mid.x; d[i][2] = mid.y; d[i][3] = mid.z; d[i][4] = mid.w; d[i][5] = row[8];
      }
      float* dst = V + (long)c * T_pad_v + ((long)n * tiles_h + p) * tiles_w + jp;
      float r[6][6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const float col[6] = {d[0][j], d[1][j], d[2][j], d[3][j], d[4][j], d[5][j]};
        float o[6];
        wino_f4::bt6(col, o);
#pragma unroll
        for (int i = 0; i < 6; ++i) r[i][j] = o[i];
      }
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        float o[6];
        wino_f4::bt6(r[i], o);
#pragma unroll
        for (int j = 0; j < 6; ++j) st_stream<4>(dst + (i * 6 + j) * stride_v, o[j]);
      }
    }
    __syncthreads();
  }
  if (amax) mscnn::publish_amax(am, amax, blockIdx.y * gridDim.x + blockIdx.x);
}

}  // namespace

namespace mscnn {

int wino_pack_weights(int m, const float* w, float* packed, int Cout, int Cin, int BM, int CK, int MT, int KI, hipStream_t st) {
  const long total = (long)MT * BM * KI * CK;
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (m == 4) wino44_weight_kernel<<<(int)blocks, 256, 0, st>>>(w, packed, Cout, Cin, BM, CK, MT, KI);
  else if (m == 3) wino33_weight_kernel<<<(int)blocks, 256, 0, st>>>(w, packed, Cout, Cin, BM, CK, MT, KI);
  else wino_weight_kernel<<<(int)blocks, 256, 0, st>>>(w, packed, Cout, Cin, BM, CK, MT, KI);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

int wino_input_transform(int m, const float* x, float* V, int N, int Cin, int H, int W, int pad_h, int pad_w, int tiles_h,
                         int tiles_w, int T_pad, hipStream_t st, bool scalar_f4, bool one_tile_per_lane) {
  const int T = N * tiles_h * tiles_w;
  dim3 grid(cdiv(T_pad, 256), Cin);
  if (m == 4 && pad_w == 1 && W % 4 == 0 && tiles_w * 4 == W && reinterpret_cast<uintptr_t>(x) % 16 == 0 && !scalar_f4) {
    // two tiles per lane where that still fills the waves (conv2_x: 240 tiles per row, conv3_x: 120 -- measured in 90 vs 99 us on
    // conv2_2, 51 vs 52 on conv3_2; conv4_x's 60 tiles would use 30 lanes of 64: 33 vs 29 us, so those keep one tile per lane)
    const double fill1 = (double)tiles_w / (64.0 * cdiv(tiles_w, 64)), fill2 = 0.5 * tiles_w / (64.0 * cdiv(tiles_w, 128));
    if (tiles_w % 2 == 0 && T_pad % 2 == 0 && reinterpret_cast<uintptr_t>(V) % 8 == 0 && !one_tile_per_lane && fill2 >= 0.85 * fill1) {
      const int segs = cdiv(tiles_w, 128);
      wino44_input_vec2_kernel<<<dim3(cdiv((long)N * tiles_h * segs, 4), Cin), 256, 0, st>>>(x, V, N, Cin, H, W, pad_h, tiles_h, tiles_w, segs, T_pad);
    } else {
      const int segs = cdiv(tiles_w, 64);
      wino44_input_vec_kernel<<<dim3(cdiv((long)N * tiles_h * segs, 4), Cin), 256, 0, st>>>(x, V, N, Cin, H, W, pad_h, tiles_h, tiles_w, segs, T_pad);
    }
    // (the GEMM's padding columns T .. T_pad of V are NOT written: every column of M depends on its own column of V only, and the
    // output transform never reads a padding column -- whatever the workspace holds there stays in padding columns)
  } else if (m == 4) {
    wino44_input_plane_kernel<<<grid, 256, 0, st>>>(x, V, N, Cin, H, W, pad_h, pad_w, tiles_h, tiles_w, T, T_pad);
  } else if (m == 3 && H * W <= kW33MaxHW) {
    dim3 g3(cdiv(N, kW33Rois), cdiv(Cin, kW33Ch));
    wino33_input_kernel<<<g3, 256, 0, st>>>(x, V, N, Cin, H, W, pad_h, pad_w, tiles_h, tiles_w, T, T_pad);
  } else if (m == 3) {
    wino33_input_plane_kernel<<<grid, 256, 0, st>>>(x, V, N, Cin, H, W, pad_h, pad_w, tiles_h, tiles_w, T, T_pad);
  } else wino_input_kernel<<<grid, 256, 0, st>>>(x, V, N, Cin, H, W, pad_h, pad_w, tiles_h, tiles_w, T, T_pad);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

int wino_output_transform(int m, const float* M, const float* bias, float* y, float* y_pool, int N, int Cout, int Ho, int Wo,
                          int tiles_h, int tiles_w, int T_pad, int relu, hipStream_t st, unsigned* amax, bool scalar_f4) {
  MSCNN_REQUIRE(!amax || m >= 3, "winograd: max |y| is published by the F(3x3,3x3) / F(4x4,3x3) output transforms only");
  // ONE predicate for "the vector F(4x4,3x3) kernel runs": the pool-only mode (y == NULL) exists in that kernel alone, so the check
  // below and the dispatch cannot drift apart (a y_pool that is only 4-byte aligned used to pass the check and fall through to the
  // scalar kernel, which stores through y)
  const bool use_vec44 = m == 4 && Wo % 4 == 0 && tiles_w * 4 == Wo && reinterpret_cast<uintptr_t>(y) % 16 == 0 &&
                         (!y_pool || reinterpret_cast<uintptr_t>(y_pool) % 8 == 0) && !scalar_f4;
  MSCNN_REQUIRE(y || (y_pool && use_vec44), "winograd: only the vector F(4x4,3x3) output transform with fused pooling (8-byte aligned pooled map) runs without y");
  const int T = N * tiles_h * tiles_w;
  dim3 grid(cdiv(T, 256), Cout);
  if (use_vec44) {
    wino44_output_vec_kernel<<<grid, 256, 0, st>>>(M, bias, y, y_pool, N, Cout, Ho, Wo, tiles_h, tiles_w, T, T_pad, relu, amax);
  } else if (m == 4) {
    wino44_output_kernel<<<grid, 256, 0, st>>>(M, bias, y, y_pool, N, Cout, Ho, Wo, tiles_h, tiles_w, T, T_pad, relu, amax);
  } else if (m == 3 && y_pool) {
    MSCNN_REQUIRE(tiles_h % 2 == 0 && tiles_w % 2 == 0, "winograd F(3x3,3x3): fused pooling needs even tile counts");
    dim3 gp(cdiv((long)N * (tiles_h / 2) * (tiles_w / 2), 256), Cout);
    wino33_output_pool_kernel<<<gp, 256, 0, st>>>(M, bias, y, y_pool, N, Cout, Ho, Wo, tiles_h, tiles_w, T_pad, relu, amax);
  } else if (m == 3 && Ho * Wo <= kW33MaxHW && !scalar_f4) {
    // (sm rows are packed [roi][nch][HW]: a float4 read of row rl needs rl * run * 4 bytes 16-aligned -- run % 4 == 0 is checked in the kernel)
    wino33_output_roi_kernel<<<dim3(cdiv(N, kW33ORois), cdiv(Cout, kW33OCh)), 256, 0, st>>>(M, bias, y, N, Cout, Ho, Wo, tiles_h, tiles_w, T_pad, relu, amax);
  } else if (m == 3) {
    wino33_output_kernel<<<grid, 256, 0, st>>>(M, bias, y, N, Cout, Ho, Wo, tiles_h, tiles_w, T, T_pad, relu, amax);
  } else {
    wino_output_kernel<<<grid, 256, 0, st>>>(M, bias, y, y_pool, N, Cout, Ho, Wo, tiles_h, tiles_w, T, T_pad, relu);
  }
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

bool wino44_outin_supported(int H, int W, int tiles_h, int tiles_w) {
  const int ns = cdiv(tiles_w, 62);      // strips of <= 62 tile columns: 1 .. 4 in one workgroup, 6 / 8 in two
  return H % 4 == 0 && W % 4 == 0 && tiles_h * 4 == H && tiles_w * 4 == W && ns <= 8 && ns != 5 && ns != 7;
}

int wino44_output_into_input(const float* M, const float* bias, float* y, float* V, int N, int C, int H, int W, int tiles_h, int tiles_w,
                             int T_pad_m, int T_pad_v, int relu, hipStream_t st, unsigned* amax, int strip_w, int chunk_rows) {
  MSCNN_REQUIRE(wino44_outin_supported(H, W, tiles_h, tiles_w), "winograd F(4x4,3x3) chain: the map must be whole 4x4 tiles");
  MSCNN_REQUIRE(y == nullptr || reinterpret_cast<uintptr_t>(y) % 16 == 0, "winograd F(4x4,3x3) chain: y must be 16-byte aligned");
  // strips of equal width <= 62 tile columns (60 + 60 for 120 columns, 4 x 60 for 240), all strips of a row in one workgroup (six /
  // eight strips -- the 8s-768 net's 160 / 320-column maps take 3 x 54 and 2 x (3 x 54) -- in two)
  const int ns = cdiv(tiles_w, 62);
  const int sgroups = ns > 4 ? 2 : 1, S = ns / sgroups;
  if (strip_w <= 0 || strip_w > 62 || cdiv(tiles_w, strip_w) != ns) strip_w = cdiv(tiles_w, ns);
  if (chunk_rows <= 0) {       // row chunks (each re-reads its two halo tile rows) until the launch has two workgroups per CU
    int chunks = 1;
    while ((long)N * C * chunks * sgroups < 512 && tiles_h / (chunks * 2) >= 9) chunks *= 2;
    chunk_rows = cdiv(tiles_h, chunks);
  }
  const int chunks = cdiv(tiles_h, chunk_rows);
  const dim3 grid(sgroups * chunks * N, C);
  const int nt_m = 36.0 * C * (double)T_pad_m * 4.0 > 200.0e6 ? 1 : 0;      // M does not fit the 256 MB Infinity Cache: stream it past the caches
#define MSCNN_OUTIN_(S_, RW_, RING_, NT_) wino44_outin_kernel<S_, RW_, RING_, NT_><<<grid, 64 * S_ * RW_, 0, st>>>(M, bias, y, V, N, C, H, W, tiles_h, tiles_w, T_pad_m, T_pad_v, relu, strip_w, sgroups, chunks, chunk_rows, amax)
#define MSCNN_OUTIN(S_, RW_, RING_) do { if (nt_m) MSCNN_OUTIN_(S_, RW_, RING_, 1); else MSCNN_OUTIN_(S_, RW_, RING_, 0); } while (0)
  if (S == 1) MSCNN_OUTIN(1, 4, 8);
  else if (S == 2) MSCNN_OUTIN(2, 2, 4);
  else if (S == 3) MSCNN_OUTIN(3, 2, 4);
  else MSCNN_OUTIN(4, 2, 4);
#undef MSCNN_OUTIN
#undef MSCNN_OUTIN_
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

}  // namespace mscnn
