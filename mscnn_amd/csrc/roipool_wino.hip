// ROIPooling x 2 (ROI + context window) fused with the F(3x3,3x3) input transform of the convolution that consumes them (roi_c1)
// for gfx950.  Replaces, for the detection sub-net of the deploy files,
//     ROIPoolingLayer::Forward_gpu x 2  (src/caffe/layers/roi_pooling_layer.cu:19-104)  ->  ConcatLayer::Forward_gpu
//     (concat_layer.cu:28-46)  ->  the im2col of roi_c1 (conv_layer.cu:8-23)
// by ONE pass from the feature map to the transform planes V[25][2C][T] the plane GEMM (wgemm.hip) reads: the R x 2C x 7 x 7
// blob between them (140 MB per 7s-576 frame: written by the pooling kernel, read back by the input transform) is never
// materialised on this path (the Net writes it on demand with roipool.hip's kernel when somebody asks for the blob by name).
//
// Why another pooling kernel.  roipool.hip's row kernel gives a SLOT of lanes one channel and lets the lanes walk the ROI's columns of
// an NCHW plane: a wave load touches 64 / Wp different rows -- 32 .. 128-byte segments of as many cache lines -- and 37 % of the
// lanes idle on ROIs whose width is not a power of two; it moves ~4 GB through the L1s for the 2 x 0.28 GB it needs (281 us,
// profiles/r03_*).  Here LANES OWN CHANNELS of a channel-last copy of the map (featT[n][h][w][c], 35 MB, one transpose launch):
// every load of a wave is one pixel x 64 channels = two full cache lines whatever the ROI's shape, no lane idles, and a wave
// handles one ROI, so its control flow (bin edges, loop bounds) is scalar.  Per bin row the wave takes the column maxima of the
// rows (up to 4 rows x 4 columns = 16 independent loads in flight per lane), parks them in a lane-private LDS column
// (conflict-free: the 64 lanes are the 64 banks' words), and finishes the row's PW bins from there -- each feature value is
// fetched once per bin ROW it belongs to (x 1.14 .. 1.18 of the window on the benchmark frame), not once per bin (x 1.3 .. 1.5).
// max is exact and order independent on post-ReLU maps, the bin edges are the reference's float expressions
// (roi_pooling_layer.cu:33-59): the pooled values are the reference's, bit for bit; V = B^T d B uses the very expressions of
// winograd.hip's ROI input transform (wino33_device.h), so V -- and everything behind it -- is bit-identical to the unfused path.
#include "roipool_wino.h"
#include "wino33_device.h"
#include <cfloat>

namespace {

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kSeg = 32;      // window columns per pass through the wave's LDS column buffer
constexpr int kRB = 4;        // rows of a bin row fetched per batch
constexpr int kCB = 4;        // columns per batch

struct RpwArgs {
  const float* featT;         // [N][H][W][C]
  const float* rois;          // [R][5]
  float* V;                   // [25][2C][T_pad]
  int R, C, H, W, T_pad;
  unsigned feat_bytes, v_bytes;
  float spatial_scale, pad_a, pad_b;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// One wave = one ROI x 64 channels, both windows (pass 0: pad_a -> channels [0, C) of the concatenated blob, pass 1: pad_b -> [C, 2C)).
template <int PH, int PW>
__global__ __launch_bounds__(256) void roipool_wino33_kernel(RpwArgs a) {
  static_assert(PH == 7 && PW == 7, "tiles per ROI: 2 x 2 tiles of 3 x 3 outputs (a 16-byte unit of V per plane)");
  __shared__ float s_col[4][kSeg][64];
  const int lane = threadIdx.x & 63, wave = uni(threadIdx.x >> 6);
  const int r = blockIdx.y * 4 + wave;
  if (r >= a.R) return;                                        // (no workgroup barrier below: waves are independent)
  const int c = blockIdx.x * 64 + lane;
  float(*col)[64] = s_col[wave];
  const __amdgpu_buffer_rsrc_t rF = make_rsrc(a.featT, a.feat_bytes), rV = make_rsrc(a.V, a.v_bytes);
  const unsigned lane_off = (unsigned)c * 4u;

  const float* roi = a.rois + 5 * (size_t)r;
  const int b = (int)roi[0];
  const float x1 = roi[1], y1 = roi[2], x2 = roi[3], y2 = roi[4];
  const int H = a.H, W = a.W, C = a.C;

#pragma unroll 1
  for (int q = 0; q < 2; ++q) {
    // ---- window geometry and bin edges: roi_pooling_layer.cu:33-59, the same float expressions (context padding, no clipping of
    // the ROI itself, edges clipped to the map); every value is wave-uniform and lives in SGPRs
    const float pad_ratio = q == 0 ? a.pad_a : a.pad_b;
    const float pad_w = (x2 - x1 + 1) * pad_ratio;
    const float pad_h = (y2 - y1 + 1) * pad_ratio;
    const int roi_start_w = (int)roundf((x1 - pad_w) * a.spatial_scale);
    const int roi_start_h = (int)roundf((y1 - pad_h) * a.spatial_scale);
    const int roi_end_w = (int)roundf((x2 + pad_w) * a.spatial_scale);
    const int roi_end_h = (int)roundf((y2 + pad_h) * a.spatial_scale);
    const int roi_width = max(roi_end_w - roi_start_w + 1, 1);
    const int roi_height = max(roi_end_h - roi_start_h + 1, 1);
    const float bin_size_h = (float)roi_height / (float)PH;
    const float bin_size_w = (float)roi_width / (float)PW;
    int hs[PH], he[PH], ws[PW], we[PW];
#pragma unroll
    for (int ph = 0; ph < PH; ++ph) {
      hs[ph] = uni(min(max((int)floorf((float)ph * bin_size_h) + roi_start_h, 0), H));
      he[ph] = uni(min(max((int)ceilf((float)(ph + 1) * bin_size_h) + roi_start_h, 0), H));
    }
#pragma unroll
    for (int pw = 0; pw < PW; ++pw) {
      ws[pw] = uni(min(max((int)floorf((float)pw * bin_size_w) + roi_start_w, 0), W));
      we[pw] = uni(min(max((int)ceilf((float)(pw + 1) * bin_size_w) + roi_start_w, 0), W));
    }
    const int x_lo = ws[0], x_hi = we[PW - 1];                 // both edge sequences are non-decreasing in pw

    float pooled[PH * PW];
#pragma unroll
    for (int i = 0; i < PH * PW; ++i) pooled[i] = -FLT_MAX;

    for (int seg_lo = x_lo; seg_lo < x_hi; seg_lo += kSeg) {
      const int seg_hi = min(seg_lo + kSeg, x_hi);
#pragma unroll
      for (int ph = 0; ph < PH; ++ph) {
        const int h0 = hs[ph], h1 = he[ph];
        if (h1 <= h0) continue;
        // column maxima of the bin row's rows over the segment's columns -> col[x - seg_lo][lane]
        for (int x = seg_lo; x < seg_hi; x += kCB) {
          float m[kCB];
#pragma unroll
          for (int j = 0; j < kCB; ++j) m[j] = -FLT_MAX;
          for (int hb = h0; hb < h1; hb += kRB) {
            float v[kRB][kCB];
#pragma unroll
            for (int i = 0; i < kRB; ++i)
#pragma unroll
              for (int j = 0; j < kCB; ++j) {
                v[i][j] = -FLT_MAX;
                if (hb + i < h1 && x + j < seg_hi)             // (wave-uniform: a scalar branch around the load)
                  v[i][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                                          rF, lane_off, (unsigned)(((b * H + hb + i) * W + x + j) * C) * 4u, 0));
              }
#pragma unroll
            for (int i = 0; i < kRB; ++i)
#pragma unroll
              for (int j = 0; j < kCB; ++j)
                if (v[i][j] > m[j]) m[j] = v[i][j];
          }
#pragma unroll
          for (int j = 0; j < kCB; ++j) col[x - seg_lo + j][lane] = m[j];
        }
        // the bin row's PW bins: maxima over their columns inside this segment, folded into the running value
#pragma unroll
        for (int pw = 0; pw < PW; ++pw) {
          const int lo = max(ws[pw], seg_lo), hi = min(we[pw], seg_hi);
          float mm = pooled[ph * PW + pw];
          for (int x = lo; x < hi; x += 4) {
            const int last = hi - 1 - seg_lo;                  // (re-reading the last column never changes a maximum)
            const float u0 = col[x - seg_lo][lane], u1 = col[min(x + 1 - seg_lo, last)][lane];
            const float u2 = col[min(x + 2 - seg_lo, last)][lane], u3 = col[min(x + 3 - seg_lo, last)][lane];
            if (u0 > mm) mm = u0;
            if (u1 > mm) mm = u1;
            if (u2 > mm) mm = u2;
            if (u3 > mm) mm = u3;
          }
          pooled[ph * PW + pw] = mm;
        }
      }
    }
    // empty bins are 0 (roi_pooling_layer.cu:56-58); a bin with cells keeps its maximum
#pragma unroll
    for (int ph = 0; ph < PH; ++ph)
#pragma unroll
      for (int pw = 0; pw < PW; ++pw)
        if (he[ph] <= hs[ph] || we[pw] <= ws[pw]) pooled[ph * PW + pw] = 0.f;

    // ---- V = B^T d B of the ROI's 2 x 2 tiles (patch rows / columns 3 t .. 3 t + 4 of the 7 x 7 map, zero beyond it): the operation
    // order of wino33_input_kernel.  Per plane (i, j) the lane stores its four tiles as ONE 16-byte unit of V[p][q C + c][4 r ..].
    const unsigned row_base = ((unsigned)(q * C) * (unsigned)a.T_pad + (unsigned)r * 4u) * 4u;      // bytes, without plane and lane parts
    const unsigned v_lane = (unsigned)c * (unsigned)a.T_pad * 4u;
    const unsigned plane_bytes = (unsigned)(2 * C) * (unsigned)a.T_pad * 4u;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      float out[5][4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int ty = t >> 1, tx = t & 1;
        float rr[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          float d[5], o[5];
#pragma unroll
          for (int k = 0; k < 5; ++k) {
            const int h = 3 * ty + k, w = 3 * tx + j;
            d[k] = (h < PH && w < PW) ? pooled[h * PW + w] : 0.f;
          }
          mscnn::bt5(d, o);
          rr[j] = o[i];
        }
        float o2[5];
        mscnn::bt5(rr, o2);
#pragma unroll
        for (int j = 0; j < 5; ++j) out[j][t] = o2[j];
      }
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        u32x4 pk;
        pk.x = __builtin_bit_cast(unsigned, out[j][0]); pk.y = __builtin_bit_cast(unsigned, out[j][1]);
        pk.z = __builtin_bit_cast(unsigned, out[j][2]); pk.w = __builtin_bit_cast(unsigned, out[j][3]);
        __builtin_amdgcn_raw_buffer_store_b128(pk, rV, v_lane, row_base + (unsigned)(i * 5 + j) * plane_bytes, 0);
      }
    }
  }
}

// featT[n][hw][c] = feat[n][c][hw]: 64 x 64 tiles through LDS, both sides coalesced
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int HW) {
  __shared__ float t[64][65];
  const int n = blockIdx.z, c0 = blockIdx.y * 64, p0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const float* src = x + (size_t)n * C * HW;
  float* dst = y + (size_t)n * C * HW;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int cc = c0 + ty + 4 * i, pp = p0 + tx;
    t[ty + 4 * i][tx] = (cc < C && pp < HW) ? src[(size_t)cc * HW + pp] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int pp = p0 + ty + 4 * i, cc = c0 + tx;
    if (pp < HW && cc < C) dst[(size_t)pp * C + cc] = t[tx][ty + 4 * i];
  }
}

}  // namespace

namespace mscnn {

bool roipool_wino33_supported(int C, int pooled_h, int pooled_w, int conv_pad_h, int conv_pad_w) {
  return pooled_h == 7 && pooled_w == 7 && conv_pad_h == 0 && conv_pad_w == 0 && C > 0 && C % 64 == 0;
}

size_t roipool_wino33_scratch_bytes(int N, int C, int H, int W) { return (size_t)N * C * H * W * sizeof(float); }

int roipool_wino33_forward(const float* feat, float* featT, const float* rois, float* V, int R, int N, int C, int H, int W, int T_pad,
                           float spatial_scale, float pad_a, float pad_b, hipStream_t st) {
  MSCNN_REQUIRE(feat && featT && rois && V, "roipool+transform: null pointer");
  MSCNN_REQUIRE(R > 0 && N > 0 && C % 64 == 0 && H > 0 && W > 0 && T_pad >= 4 * R && T_pad % 4 == 0, "roipool+transform: bad shape");
  const double fb = (double)N * C * H * W * 4.0, vb = 25.0 * 2.0 * C * (double)T_pad * 4.0;
  MSCNN_REQUIRE(fb < 4.0e9 && vb < 4.0e9, "roipool+transform: feature map or transform planes beyond a 32-bit buffer window");
  MSCNN_REQUIRE(reinterpret_cast<uintptr_t>(V) % 16 == 0, "roipool+transform: V must be 16-byte aligned");
  const int HW = H * W;
  nchw_to_nhwc_kernel<<<dim3(cdiv(HW, 64), cdiv(C, 64), N), 256, 0, st>>>(feat, featT, C, HW);
  MSCNN_POST_LAUNCH();
  RpwArgs a;
  a.featT = featT; a.rois = rois; a.V = V;
  a.R = R; a.C = C; a.H = H; a.W = W; a.T_pad = T_pad;
  a.feat_bytes = (unsigned)fb; a.v_bytes = (unsigned)vb;
  a.spatial_scale = spatial_scale; a.pad_a = pad_a; a.pad_b = pad_b;
  // channel block on grid.x: workgroups go round-robin over the 8 XCDs by linear id, so XCD j only ever touches channel blocks
  // == j (mod 8) of the map -- with C = 512 exactly one 64-channel slice (4.4 MB) per XCD's L2 -- and the four 16-byte pieces
  // that complete a 64-byte run of a V row come from one workgroup, the next four from the next workgroup on the same XCD
  roipool_wino33_kernel<7, 7><<<dim3(C / 64, cdiv(R, 4)), 256, 0, st>>>(a);
  MSCNN_POST_LAUNCH();
  return MSCNN_OK;
}

}  // namespace mscnn
